#!/usr/bin/env python3
"""Prints the launch shape the library picks (SADBM_TRACE=1) for a few jobs (GPU box)."""
import os, sys
os.environ["SADBM_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
for (W, H, pairs, win, D) in [(3840, 2160, 1, 7, 32), (1920, 1080, 8, 21, 32), (1920, 1080, 64, 7, 30), (384, 288, 256, 7, 30)]:
    dL = torch.zeros((pairs, H, W), dtype=torch.uint8, device="cuda"); dR = torch.zeros_like(dL)
    for d in ("0", "1"):
        os.environ["SADBM_DIRECT"] = d
        oa.reload_env()
        oa.compute_batch_device(dL, dR, win, D, 0)
    torch.cuda.synchronize()
