#!/usr/bin/env python3
"""Rows-per-block sweep (SADBM_STRIP_TY) of the on-the-fly strip kernels against the model's choice (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

jobs = [(1920, 1080, 64, 7, 30), (1920, 1080, 64, 9, 64), (384, 288, 256, 7, 30), (3840, 2160, 32, 7, 32), (1242, 375, 256, 7, 16), (3840, 2160, 1, 7, 32), (1920, 1080, 1, 7, 30), (384, 288, 1, 7, 30), (1280, 720, 4, 11, 24), (640, 480, 64, 15, 64), (1920, 1080, 8, 21, 32)]
if len(sys.argv) > 1:                                         # tools/ty_otf.py "W,H,pairs,win,D" ...: these jobs, frames read in place only
    jobs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (W, H, pairs, win, D) in jobs:
    L, R = synth.rds(7, W, H, 4, min(D - 1, 20), 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    for direct in (("1",) if len(sys.argv) > 1 else ("0", "1")):
        line = "%4dx%4d x%3d w%2d D%2d direct=%s:" % (W, H, pairs, win, D, direct)
        for ty in (0, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 0):   # the model's choice first (clocks still ramping up) and last
            os.environ["SADBM_DIRECT"] = direct
            os.environ["SADBM_STRIP_TY"] = str(ty)
            oa.reload_env()
            ctx = oa.Context(0)
            for _ in range(2):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            reps = 8 if pairs > 1 else 40
            ctx.profile_begin(reps)
            for _ in range(reps):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n, m, p = ctx.profile_end()
            line += " %s:%.4f" % (ty if ty else "model", m / n)
        print(line, flush=True)
