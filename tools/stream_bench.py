#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) frame rate of the streaming pipeline vs the synchronous host call (GPU box).
   tools/stream_bench.py [frames]   -- rates go to DESIGN.md section 7; never bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import opencv_amd as oa
from opencv_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for name, (W, H, win, D) in {"1080p_w9_d64": (1920, 1080, 9, 64), "1080p_w9_d128": (1920, 1080, 9, 128), "4k_w15_d128": (3840, 2160, 15, 128)}.items():
    L, R = synth.rds(3, W, H, 16, 96, 0)
    exp = oa.get_disparity_host(L, R, win, D)[:64]          # correctness of the pipeline vs the oracle: tests/test_gpu_pipeline.py
    # synchronous host API (pageable buffers, one frame at a time)
    oa.get_disparity_host(L, R, win, D)
    t0 = time.perf_counter()
    for _ in range(10):
        out = oa.get_disparity_host(L, R, win, D)
    sync_fps = 10 / (time.perf_counter() - t0)
    assert np.array_equal(out[:64], exp)
    for depth in (1, 2, 3, 4):
        pipe = oa.Pipeline(H, W, 1, win, D, depth=depth)
        for mode in ("zero-copy", "memcpy"):
            tick = []
            t0 = time.perf_counter()
            for i in range(N):
                if i >= depth:
                    res = pipe.wait(tick[i - depth])
                if mode == "zero-copy":
                    if i < depth:                      # each slot's pinned inputs are filled once (a capture card would DMA into them)
                        li, ri = pipe.inputs(); li[...] = L; ri[...] = R
                    tick.append(pipe.submit())
                else:
                    tick.append(pipe.submit(L, R))
            for i in range(max(0, N - depth), N):
                res = pipe.wait(tick[i])
            dt = time.perf_counter() - t0
            assert np.array_equal(res[:64], exp)
            print("%-14s depth=%d %-9s %8.1f frames/s  %7.1f Gdisp-evals/s   (sync host call: %.1f frames/s)"
                  % (name, depth, mode, N / dt, N * W * H * D / dt / 1e9, sync_fps), flush=True)
        pipe.close()
