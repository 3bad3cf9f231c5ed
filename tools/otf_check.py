#!/usr/bin/env python3
"""The 32-disparity form of the strip matcher (SADBM_OTF=1) on the GPU box: parity against the oracle over windows, disparities and
shapes, then its time beside the default choice (quad matcher / 64-disparity strip blocks).   tools/otf_check.py [parity|time|all]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
import oracle

mode = sys.argv[1] if len(sys.argv) > 1 else "all"


def setenv(v, ng=None):
    for k, x in (("SADBM_OTF", v), ("SADBM_OTF_NG", ng)):
        if x is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(x)
    oa.reload_env()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


if mode in ("parity", "all"):
    bad_total = 0
    rng = np.random.default_rng(5)
    ctx = oa.Context(0)
    for win, ngf, D in [(w, n, d) for w in range(4, 26) for (n, ds) in ((None, (1, 5, 16, 17, 30, 32)), (2, (3, 16)), (4, (7, 31, 33, 50, 64)), (8, (40, 70, 100, 128))) for d in ds]:
        setenv(1, ngf)
        if True:
            for (H, W) in ((win + 9, 203), (40, 517), (33, 260)):
                binary = (win + D + W) & 1
                L = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
                R = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
                s = int(rng.integers(0, max(1, D)))
                if s:
                    R[:, :W - s] = L[:, s:]
                if binary:
                    L = np.where(L & 0x80, 255, 0).astype(np.uint8); R = np.where(R & 0x80, 255, 0).astype(np.uint8)
                exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
                got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
                k = ctx.last_kernel()
                if not k.startswith("sad_strip_otf_kernel"):
                    print("NOT OTF", win, D, H, W, k); bad_total += 1; continue
                nb = int((got != exp).sum())
                if nb:
                    bad = np.argwhere(got != exp)
                    print("MISMATCH ng %s win %d D %d %dx%d: %d bad, first %s got %d exp %d" % (ngf, win, D, H, W, nb, bad[0], got[tuple(bad[0])], exp[tuple(bad[0])]), flush=True)
                    bad_total += 1
    setenv(1)
    # batches with frame groups, bands
    for (H, W, n, win, D) in ((61, 517, 5, 7, 30), (40, 1242, 3, 9, 32), (90, 384, 8, 7, 17), (33, 260, 4, 15, 31), (50, 300, 2, 21, 24), (60, 384, 6, 7, 12), (45, 700, 3, 17, 16),
                               (64, 517, 4, 9, 64), (40, 640, 3, 15, 48)):
        Ls, Rs = zip(*[oracle.rds(70 + i, W, H, 2 + i % 3, min(D - 1, 12 + i), i & 1) for i in range(n)])
        out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel().startswith("sad_strip_otf_kernel"), ctx.last_kernel()
        for i in range(n):
            if not np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)):
                print("BATCH MISMATCH", H, W, n, win, D, i); bad_total += 1
        r0, nb = H // 3, H // 3
        i1 = min(r0 + nb + win - 1, H)
        got = oa.compute_band_device(dev(Ls[0][r0:i1]), dev(Rs[0][r0:i1]), H, r0, nb, win, D, 0).cpu().numpy()
        if not np.array_equal(got, oracle.box(Ls[0], Rs[0], win, D)[r0:r0 + nb]):
            print("BAND MISMATCH", H, W, win, D); bad_total += 1
    Lf, Rf = oracle.rds(2, 1920, 1080, 4, 20, 0)
    if not np.array_equal(oa.compute_device(dev(Lf), dev(Rf), 7, 30, 0).cpu().numpy(), oracle.box(Lf, Rf, 7, 30)):
        print("1080p MISMATCH"); bad_total += 1
    print("parity: %d failing cases" % bad_total, flush=True)
    if bad_total:
        sys.exit(1)

if mode in ("time", "all"):
    jobs = [(1920, 1080, 64, 9, 64), (1920, 1080, 64, 7, 48), (3840, 2160, 16, 15, 64), (1920, 1080, 64, 7, 16), (1920, 1080, 64, 7, 8), (1920, 1080, 64, 15, 12), (384, 288, 256, 7, 16),
            (3840, 2160, 32, 7, 32), (1920, 1080, 64, 7, 30), (384, 288, 256, 7, 30), (1920, 1080, 64, 9, 32), (1920, 1080, 64, 15, 32),
            (1242, 375, 128, 7, 16), (3840, 2160, 16, 21, 32), (1920, 1080, 1, 7, 30), (384, 288, 1, 7, 30), (1280, 720, 1, 7, 30), (3840, 2160, 1, 7, 32)]
    for (W, H, pairs, win, D) in jobs:
        L, R = synth.rds(7, W, H, 4, min(D - 1, 20), 0)
        dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
        dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
        res = {}
        outs = {}
        for name, v in (("default", 0), ("otf", 1)):
            setenv(v)
            ctx = oa.Context(0)
            for _ in range(3):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            reps = 10 if pairs > 1 else 50
            ctx.profile_begin(reps)
            for _ in range(reps):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n, m, _p = ctx.profile_end()
            res[name] = (m / n, ctx.last_kernel())
            outs[name] = dO.clone()
        same = bool(torch.equal(outs["default"], outs["otf"]))
        ev = W * H * D * pairs
        print("%4dx%4d x%3d w%2d D%2d: default %.4f ms (%s, %.0f Gd/s)  otf %.4f ms (%.0f Gd/s)  otf/default %.3f  equal %s" % (
            W, H, pairs, win, D, res["default"][0], res["default"][1], ev / res["default"][0] / 1e6, res["otf"][0], ev / res["otf"][0] / 1e6,
            res["otf"][0] / res["default"][0], same), flush=True)
        del dL, dR, dO
    setenv(None)

if mode == "sweep":
    for (W, H, pairs) in ((1920, 1080, 32), (384, 288, 256), (1920, 1080, 1), (384, 288, 1), (640, 480, 8)):
        L, R = synth.rds(7, W, H, 2, 3, 0)
        dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
        dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
        for D in [int(x) for x in os.environ.get("SWEEP_D", "4,8,12,16,17,24,30,32").split(",")]:
            for win in (4, 5, 7, 9, 11, 15, 21, 25):
                res = {}
                for name, v, fl in (("quad", 0, oa.KERNEL_QUAD), ("strip64", 0, oa.KERNEL_STRIP), ("otf", 1, 0)):
                    setenv(v)
                    ctx = oa.Context(0)
                    for _ in range(2):
                        oa.compute_batch_device(dL, dR, win, D, fl, out=dO, ctx=ctx)
                    torch.cuda.synchronize()
                    reps = 6 if pairs > 1 else 30
                    ctx.profile_begin(reps)
                    for _ in range(reps):
                        oa.compute_batch_device(dL, dR, win, D, fl, out=dO, ctx=ctx)
                    torch.cuda.synchronize()
                    n, m, _p = ctx.profile_end()
                    res[name] = m / n
                best = min(res, key=res.get)
                print("%4dx%4d x%3d D%2d w%2d: quad %.4f strip64 %.4f otf %.4f ms  otf/best-other %.3f  best %s" % (
                    W, H, pairs, D, win, res["quad"], res["strip64"], res["otf"], res["otf"] / min(res["quad"], res["strip64"]), best), flush=True)
    setenv(None)
