#!/usr/bin/env python3
"""Parity of the packed-V windows (17..21) on the GPU box: largest costs, ties, k >= D slots, left border, both kernel forms.
   tools/pkv_check.py  -- exits non-zero on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
import oracle

def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
n = 0
for form in ("default", "staged"):
    if form == "staged": os.environ["SADBM_OTF"] = "0"
    else: os.environ.pop("SADBM_OTF", None)
    oa.reload_env()
    for win in (17, 18, 19, 20, 21, 22, 16):
        H, W = win + 9, 620
        big = np.full((H, W), 255, np.uint8); zero = np.zeros((H, W), np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        chk = np.where((xx // 3 + yy) & 1, 255, 0).astype(np.uint8)
        g = np.random.default_rng(win)
        rnd = g.integers(0, 256, (H, W), dtype=np.uint8); rnd2 = np.roll(rnd, -37, axis=1)
        for D in (30, 64, 100, 129, 200, 256):
            for L, R in ((big, zero), (zero, big), (chk, 255 - chk), (big, chk), (rnd, rnd2), (rnd, rnd)):
                exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
                ctx = oa.Context(0)
                got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
                n += 1
                if not np.array_equal(got, exp):
                    bad = np.argwhere(got != exp)
                    print("MISMATCH", form, win, D, ctx.last_kernel(), len(bad), bad[:6], got[tuple(bad[0])], exp[tuple(bad[0])]); sys.exit(1)
    print(form, "ok", n, flush=True)
print("pkv_check passed:", n)
