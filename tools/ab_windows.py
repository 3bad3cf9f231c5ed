#!/usr/bin/env python3
"""Matcher time per window and search range for one library build (SADBM_LIB selects it): tools/ab_windows.py [tag]
   prints 'tag win D kernel ms' per case; run once per build and join the outputs (tools/ab_windows_join.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
wins = [int(x) for x in os.environ.get("AB_WINS", ",".join(str(w) for w in range(4, 26))).split(",")]
cases = [(30, 16, 1080, 1920), (64, 16, 1080, 1920), (128, 8, 1080, 1920), (256, 4, 1080, 1920)]
g = torch.Generator().manual_seed(1)
for D, n, H, W in cases:
    L = torch.randint(0, 256, (n, H, W), dtype=torch.uint8, generator=g).cuda()
    R = torch.randint(0, 256, (n, H, W), dtype=torch.uint8, generator=g).cuda()
    for win in wins:
        ctx = oa.Context(0)
        out = oa.compute_batch_device(L, R, win, D, 0, ctx=ctx)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): oa.compute_batch_device(L, R, win, D, 0, ctx=ctx, out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        print(tag, win, D, ctx.last_kernel(), "%.4f" % best, flush=True)
