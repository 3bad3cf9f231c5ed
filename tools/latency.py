#!/usr/bin/env python3
"""Single-call latency of the reference's own configuration (384x288 BGR pair, winSize 7, DSR 30: SAD.cpp:9-11,65-66) and of one
1080p / 4K frame: synchronous host call (pageable numpy in, numpy out) and device-resident call (launch to completion).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

for name, (W, H, ch, win, D) in {"tsukuba-size 384x288 BGR w7 D30": (384, 288, 3, 7, 30), "1080p gray w9 D64": (1920, 1080, 1, 9, 64),
                                 "4K gray w15 D128": (3840, 2160, 1, 15, 128), "1080p BGR w9 D64": (1920, 1080, 3, 9, 64),
                                 "4K BGR w15 D128": (3840, 2160, 3, 15, 128), "4K BGRA w15 D128": (3840, 2160, 4, 15, 128)}.items():
    L, R = synth.rds(5, W, H, 4, 20, 0)
    if ch > 1:
        L = np.ascontiguousarray(np.stack([L] * ch, -1)); R = np.ascontiguousarray(np.stack([R] * ch, -1))
    for _ in range(3):
        oa.get_disparity_host(L, R, win, D)
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        oa.get_disparity_host(L, R, win, D)
    host_us = (time.perf_counter() - t0) / n * 1e6
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    ctx = oa.Context(0)
    for _ in range(3):
        oa.compute_device(dL, dR, win, D, 0, out=out, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        oa.compute_device(dL, dR, win, D, 0, out=out, ctx=ctx)
        torch.cuda.synchronize()
    dev_us = (time.perf_counter() - t0) / n * 1e6
    ctx.profile_begin(n)
    for _ in range(n):
        oa.compute_device(dL, dR, win, D, 0, out=out, ctx=ctx)
    torch.cuda.synchronize()
    nl, m, p = ctx.profile_end()
    print("%-34s host call %8.1f us   device call + sync %7.1f us   matcher kernel %6.1f us  prep %5.1f us   (%s)"
          % (name, host_us, dev_us, m / nl * 1e3, p / nl * 1e3, ctx.last_kernel()), flush=True)
