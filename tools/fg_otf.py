#!/usr/bin/env python3
"""Frame-group sweep of the on-the-fly kernels, direct and through the prep pass, at fixed rows per block (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
W, H, pairs, win, D = 1920, 1080, 64, 7, 30
L, R = synth.rds(7, W, H, 4, 20, 0)
dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
for ty in ("64", "0"):
    for direct in ("0", "1"):
        line = "ty=%s direct=%s:" % (ty, direct)
        for fg in ("1", "2", "4", "8"):
            os.environ.update(SADBM_DIRECT=direct, SADBM_STRIP_FG=fg, SADBM_STRIP_TY=ty)
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
            line += "  fg%s %.4f" % (fg, m / n)
        print(line, flush=True)
