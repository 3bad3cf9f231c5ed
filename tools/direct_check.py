#!/usr/bin/env python3
"""Matcher + prep time of the on-the-fly strip kernels with and without direct reads, with and without frame groups (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

jobs = [(1920, 1080, 64, 7, 30), (1920, 1080, 64, 9, 64), (384, 288, 256, 7, 30), (3840, 2160, 32, 7, 32), (1242, 375, 256, 7, 30), (640, 480, 128, 9, 48),
        (1920, 1080, 1, 7, 30), (384, 288, 1, 7, 30)]
for (W, H, pairs, win, D) in jobs:
    L, R = synth.rds(7, W, H, 4, min(D - 1, 20), 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    line = "%4dx%4d x%3d w%2d D%2d:" % (W, H, pairs, win, D)
    for direct in ("0", "1"):
        for fg in ("1", None):
            os.environ["SADBM_DIRECT"] = direct
            if fg is None:
                os.environ.pop("SADBM_STRIP_FG", None)
            else:
                os.environ["SADBM_STRIP_FG"] = fg
            oa.reload_env()
            ctx = oa.Context(0)
            for _ in range(3):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            reps = 10 if pairs > 1 else 50
            ctx.profile_begin(reps)
            for _ in range(reps):
                oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n, m, p = ctx.profile_end()
            line += "  direct=%s fg=%s %.4f+%.4f" % (direct, fg or "auto", m / n, p / n)
    print(line, flush=True)
