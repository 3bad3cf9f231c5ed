#!/usr/bin/env python3
"""Direct-read frame groups: does the frame stride of the batch matter? (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
W, H, pairs, win, D = 1920, 1080, 64, 7, 30
L, R = synth.rds(7, W, H, 4, 20, 0)
for extra_rows, extra_cols in ((0, 0), (1, 0), (3, 0), (0, 64), (0, 128), (5, 192)):
    bL = torch.zeros((pairs, H + extra_rows, W + extra_cols), dtype=torch.uint8, device="cuda")
    bR = torch.zeros_like(bL)
    bL[:, :H, :W] = torch.from_numpy(L).cuda(); bR[:, :H, :W] = torch.from_numpy(R).cuda()
    dL, dR = bL[:, :H, :W], bR[:, :H, :W]
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    line = "frame stride %d B, pitch %d:" % (dL.stride(0), dL.stride(1))
    for direct, fg in (("0", "8"), ("1", "1"), ("1", "2"), ("1", "8")):
        os.environ.update(SADBM_DIRECT=direct, SADBM_STRIP_FG=fg, SADBM_STRIP_TY="64")
        oa.reload_env()
        ctx = oa.Context(0)
        for _ in range(2):
            oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        ctx.profile_begin(8)
        for _ in range(8):
            oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        n, m, p = ctx.profile_end()
        line += "  direct=%s fg%s %.4f+%.4f" % (direct, fg, m / n, p / n)
    print(line, flush=True)
