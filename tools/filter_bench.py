#!/usr/bin/env python3
"""filter2D (35x35 float64 Gaussian of gaus_filter.cpp) on the GPU vs the CPU definition (GPU box).  Numbers go to DESIGN.md section 8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from oracle import filters as F

k = oa.filters.GetgaussianKernel(35, 3)
for name, (H, W) in {"410x361 (cat/dog.bmp size)": (361, 410), "1920x1080": (1080, 1920), "3840x2160": (2160, 3840)}.items():
    img = np.random.default_rng(1).integers(0, 256, (H, W, 3), dtype=np.uint8)
    d = torch.from_numpy(img).cuda()
    out = oa.filters.filter2D(d, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        oa.filters.filter2D(d, k, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    rows = min(H, 120)
    t0 = time.perf_counter(); ref = F.filter2d(img[:rows + 34], k); cpu_s = (time.perf_counter() - t0) * H / (rows + 34)
    ok = np.array_equal(out.cpu().numpy()[:rows], ref[:rows])
    print("%-28s GPU %8.3f ms (%7.1f Mpixel/s, %6.2f TFLOP/s f64)   CPU definition ~%7.1f ms (16 OpenMP threads, extrapolated)   bit-exact=%s"
          % (name, ms, H * W / ms / 1e3, 2 * 3 * 1225 * H * W / ms / 1e9, cpu_s * 1e3, ok), flush=True)
