#!/usr/bin/env python3
"""Matcher time of a fixed list of jobs with whatever library SADBM_LIB names (A/B of variant builds; GPU box).
   tools/job_times.py "W,H,pairs,win,D" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
for spec in sys.argv[1:]:
    W, H, pairs, win, D = (int(x) for x in spec.split(","))
    L, R = synth.rds(7, W, H, 4, min(D - 1, 20), 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    ctx = oa.Context(0)
    best = 1e9
    for rep in range(3):
        for _ in range(2):
            oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        n_ = 8 if pairs > 1 else 40
        ctx.profile_begin(n_)
        for _ in range(n_):
            oa.compute_batch_device(dL, dR, win, D, 0, out=dO, ctx=ctx)
        torch.cuda.synchronize()
        n, m, p = ctx.profile_end()
        best = min(best, m / n)
    print("%s %.4f %s" % (spec, best, ctx.last_kernel()), flush=True)
