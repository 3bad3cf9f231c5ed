#!/usr/bin/env python3
"""Strip vs quad matcher on small D <= 32 jobs (where does the default choice of strip_is_default() cross over?)  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

ctx = oa.Context(0)
for (W, H, win, D) in [(384, 288, 7, 30), (640, 480, 7, 30), (1280, 720, 7, 30), (1920, 1080, 7, 30), (384, 288, 9, 16), (640, 480, 11, 32)]:
    for pairs in (1, 2, 4, 8, 16, 32):
        if W * H * pairs > 9e6:
            continue
        L, R = synth.rds(7, W, H, 8, 20, 0)
        dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
        dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
        ms = {}
        for name, fl in (("quad", oa.KERNEL_QUAD), ("strip", oa.KERNEL_STRIP)):
            for _ in range(3):
                oa.compute_batch_device(dL, dR, win, D, fl, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            ctx.profile_begin(20)
            for _ in range(20):
                oa.compute_batch_device(dL, dR, win, D, fl, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n, m, _p = ctx.profile_end()
            ms[name] = m / n
        print("%4dx%4d x%2d w%2d D%2d  %.2f Mpx  quad %.4f strip %.4f ms  strip/quad %.3f" % (W, H, pairs, win, D, W * H * pairs / 1e6, ms["quad"], ms["strip"],
                                                                                     ms["strip"] / ms["quad"]), flush=True)
