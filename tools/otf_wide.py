#!/usr/bin/env python3
"""On-the-fly strip kernels at 128- and 256-disparity blocks (SADBM_OTF=1) beside the staged strip matcher (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
import oracle
jobs = [(3840, 2160, 16, 15, 128), (7680, 4320, 2, 21, 256), (1242, 375, 256, 7, 128), (1920, 1080, 32, 9, 128), (1920, 1080, 32, 7, 100), (3840, 2160, 8, 11, 256), (3840, 2160, 8, 25, 192),
        (3840, 2160, 1, 15, 128), (1920, 1080, 1, 9, 128)]
for (W, H, pairs, win, D) in jobs:
    L, R = synth.rds(7, W, H, 4, min(D - 1, 90), 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    exp = oracle.box(L, R, win, D)
    line = "%4dx%4d x%3d w%2d D%3d:" % (W, H, pairs, win, D)
    for name, env in (("staged", {"SADBM_OTF": "0"}), ("otf", {"SADBM_OTF": "1", "SADBM_DIRECT": "0"}), ("otf+direct", {"SADBM_OTF": "1", "SADBM_DIRECT": "1"})):
        for k in ("SADBM_OTF", "SADBM_DIRECT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        oa.reload_env()
        ctx = oa.Context(0)
        best = 1e9
        for rep in range(2):
            for _ in range(2):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n_ = 6 if pairs > 1 else 30
            ctx.profile_begin(n_)
            for _ in range(n_):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n, m, p = ctx.profile_end()
            best = min(best, (m + p) / n)
        ok = bool(np.array_equal(dO[pairs - 1].cpu().numpy(), exp))
        line += "  %s %.4f ms (%s)%s" % (name, best, ctx.last_kernel().replace("sad_strip_", ""), "" if ok else " MISMATCH")
    print(line, flush=True)
