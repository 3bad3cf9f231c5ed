#!/usr/bin/env python3
"""Times the quad matcher with full-width (256-thread) and half-width (128-thread) blocks over frame widths (GPU box)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

ctx = oa.Context(0)
for (W, H, pairs, win, D) in [(384, 288, 256, 7, 30), (1920, 1080, 8, 7, 30), (3840, 2160, 2, 7, 32), (1920, 1080, 8, 7, 16), (640, 480, 64, 9, 32),
                              (1280, 720, 16, 7, 30), (450, 375, 128, 7, 30), (1920, 1080, 8, 11, 64), (384, 288, 256, 7, 64), (1024, 768, 16, 5, 24),
                              (1920, 1080, 1, 7, 30), (1280, 720, 1, 7, 30)]:
    L, R = synth.rds(7, W, H, 8, min(40, D), 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    res, dig = {}, {}
    for half in ("0", "1", "-1"):
        os.environ["SADBM_QUAD_HALF"] = half
        oa.reload_env()
        for _ in range(2):
            oa.compute_batch_device(dL, dR, win, D, oa.KERNEL_QUAD, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        ctx.profile_begin(8)
        for _ in range(8):
            oa.compute_batch_device(dL, dR, win, D, oa.KERNEL_QUAD, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        n, m, _p = ctx.profile_end()
        res[half] = m / n
        dig[half] = hashlib.sha256(dO.cpu().numpy().tobytes()).hexdigest()[:12]
    ok = dig["0"] == dig["1"] == dig["-1"]
    print("%4dx%4d x%3d w%2d D%3d  full %.4f half %.4f auto %.4f ms  half/full %.3f  same=%s" % (W, H, pairs, win, D, res["0"], res["1"], res["-1"],
                                                                                            res["1"] / res["0"], ok), flush=True)
