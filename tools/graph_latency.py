#!/usr/bin/env python3
"""Call-to-sync time of the reference's own configuration (384x288 BGR pair, winSize 7, DSR 30): plain device call,
static entry, and the static entry captured in a HIP graph and replayed (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

for (W, H, ch, win, D) in [(384, 288, 3, 7, 30), (384, 288, 1, 7, 30), (1920, 1080, 1, 9, 64)]:
    L, R = synth.rds(1, W, H, 3, 14, 0)
    if ch == 3:
        L = np.ascontiguousarray(np.stack([L, L, L], -1)); R = np.ascontiguousarray(np.stack([R, R, R], -1))
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ctx = oa.Context(0)

    def timed(fn, n=300):
        for _ in range(20):
            fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e6

    a = timed(lambda: oa.compute_device(dL, dR, win, D, out=out, ctx=ctx))
    b = timed(lambda: oa.compute_device_static(dL, dR, win, D, out=out))
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        oa.compute_device_static(dL, dR, win, D, out=out)
    c = timed(g.replay)
    print("%dx%d ch%d w%d D%d: compute_device %.1f us, static %.1f us, graph replay %.1f us (median call-to-sync, through Python)" % (W, H, ch, win, D, a, b, c), flush=True)
