#!/usr/bin/env python3
"""Matcher time of small single frames against the strip matcher's rows per block (SADBM_STRIP_TY) (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth

ctx = oa.Context(0)
for (W, H, win, D) in [(384, 288, 7, 30), (640, 480, 7, 30), (640, 480, 9, 64), (1280, 720, 9, 64), (1920, 1080, 9, 64), (1242, 375, 7, 128), (384, 288, 15, 128)]:
    L, R = synth.rds(1, W, H, 3, 14, 0)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    res = []
    for ty in ("0", "1", "2", "3", "4", "6", "8", "12", "16", "24", "32"):
        if ty == "0":
            os.environ.pop("SADBM_STRIP_TY", None)
        else:
            os.environ["SADBM_STRIP_TY"] = ty
        oa.reload_env()
        for _ in range(5):
            oa.compute_device(dL, dR, win, D, oa.KERNEL_STRIP, out=out, ctx=ctx)
        torch.cuda.synchronize()
        ctx.profile_begin(30)
        for _ in range(30):
            oa.compute_device(dL, dR, win, D, oa.KERNEL_STRIP, out=out, ctx=ctx)
        torch.cuda.synchronize()
        n, m, _p = ctx.profile_end()
        res.append("%s:%.1f" % ("model" if ty == "0" else ty, m / n * 1e3))
    print("%4dx%4d w%2d D%3d  us by TY  %s" % (W, H, win, D, "  ".join(res)), flush=True)
