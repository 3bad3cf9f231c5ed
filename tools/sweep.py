#!/usr/bin/env python3
"""Times the quad and strip matchers over a (win, D) grid on one frame size (GPU box):  tools/sweep.py [W H pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

W, H, pairs = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 4)
L, R = synth.rds(7, W, H, 8, 40, 0)
dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
ctx = oa.Context(0)
print("W H pairs", W, H, pairs)
for D in [int(x) for x in os.environ.get("SWEEP_D", "32,64,96,128,192,256").split(",")]:
    for win in (5, 7, 9, 11, 13, 15, 17, 21, 25):
        ms = {}
        for name, fl in (("quad", oa.KERNEL_QUAD), ("strip", oa.KERNEL_STRIP)):
            for _ in range(2):
                oa.compute_batch_device(dL, dR, win, D, fl, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            ctx.profile_begin(8)
            for _ in range(8):
                oa.compute_batch_device(dL, dR, win, D, fl, out=dO, ctx=ctx)
            torch.cuda.synchronize()
            n, m, _p = ctx.profile_end()
            ms[name] = m / n
        print("D=%3d win=%2d quad %.4f strip %.4f ms  strip/quad %.3f" % (D, win, ms["quad"], ms["strip"], ms["strip"] / ms["quad"]), flush=True)
