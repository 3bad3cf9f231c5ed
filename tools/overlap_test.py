#!/usr/bin/env python3
"""Does splitting a batch over several streams (prep of one part under the matcher of another) pay?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

W, H, pairs, win, D = 3840, 2160, 32, 15, 128
if len(sys.argv) > 5:
    W, H, pairs, win, D = [int(x) for x in sys.argv[1:6]]
L, R = synth.rds(3, W, H, 16, 96, 0)
dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
ref = None
for parts in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    ctxs = [oa.Context(0) for _ in range(parts)]
    per = pairs // parts
    def step():
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                oa.compute_batch_device(dL[i * per:(i + 1) * per], dR[i * per:(i + 1) * per], win, D, 0, out=dO[i * per:(i + 1) * per], ctx=ctxs[i])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    h = int(dO.to(torch.int64).sum().item())
    ref = ref or h
    print("parts %d: %.3f ms/step  %.1f Gd/s  same=%s" % (parts, ms, W * H * D * pairs / ms / 1e6, h == ref), flush=True)
