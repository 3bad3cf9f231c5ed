#!/usr/bin/env python3
"""Rows-per-block sweep (SADBM_STRIP_TY) of the staged strip matcher on large batches: do whole-frame bands pay? (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
jobs = [(3840, 2160, 32, 15, 128), (3840, 2160, 24, 15, 128), (3840, 2160, 48, 15, 128), (3840, 2160, 64, 15, 128), (7680, 4320, 4, 21, 256), (7680, 4320, 8, 21, 256),
        (1242, 375, 512, 7, 128), (1920, 1080, 64, 9, 128)]
if len(sys.argv) > 1:
    jobs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (W, H, pairs, win, D) in jobs:
    L, R = synth.rds(7, W, H, 4, 90, 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    nv = H - win
    line = "%4dx%4d x%3d w%2d D%3d:" % (W, H, pairs, win, D)
    tys = [0, 64, 128, 256, 512] + sorted({(nv + k - 1) // k for k in (1, 2, 3, 4, 6)}, reverse=True) + [0]
    if pairs <= 2:                                             # single frames: short bands
        tys = [0, 6, 8, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128, 0]
    for ty in tys:
        os.environ["SADBM_STRIP_TY"] = str(ty)
        oa.reload_env()
        ctx = oa.Context(0)
        for _ in range(2):
            oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        reps = 5 if pairs > 2 else 30
        ctx.profile_begin(reps)
        for _ in range(reps):
            oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        n, m, p = ctx.profile_end()
        line += " %s:%.3f" % (ty if ty else "model", m / n)
    print(line, flush=True)
    del dL, dR, dO
