#!/usr/bin/env python3
"""Colour frames read in place (round 3) against the prep pass (SADBM_DIRECT=0): batches of BGR / BGRA frames, device-resident, ms per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa

def run(W, H, ch, n, win, D):
    g = np.random.default_rng(1)
    L = torch.from_numpy(g.integers(0, 256, (n, H, W, ch), dtype=np.uint8)).cuda()
    R = torch.from_numpy(g.integers(0, 256, (n, H, W, ch), dtype=np.uint8)).cuda()
    out = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
    res = []
    for d in ("-1", "0"):
        os.environ["SADBM_DIRECT"] = d
        oa.reload_env()
        ctx = oa.Context(0)
        for _ in range(3):
            oa.compute_batch_device(L, R, win, D, 0, out=out, ctx=ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            oa.compute_batch_device(L, R, win, D, 0, out=out, ctx=ctx)
        torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) / 20 * 1e3, ctx.last_kernel(), out.cpu().numpy().copy()))
    assert np.array_equal(res[0][2], res[1][2])
    print("%dx%d x%d ch%d win %d D %d: in place %.4f ms (%s)  prep pass %.4f ms (%s)  x%.3f" % (W, H, n, ch, win, D, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] / res[1][0]), flush=True)

for args in ((384, 288, 3, 1, 7, 30), (384, 288, 3, 256, 7, 30), (1920, 1080, 3, 64, 7, 30), (1920, 1080, 4, 64, 7, 30), (1920, 1080, 3, 64, 9, 64), (1920, 1080, 3, 1, 9, 64),
             (1920, 1080, 3, 32, 12, 48), (1920, 1080, 3, 32, 4, 32), (3840, 2160, 3, 16, 8, 32)):
    run(*args)
