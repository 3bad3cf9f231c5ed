#!/usr/bin/env python3
"""How long the CPU oracle takes on the box this runs on (sizes the full-frame parity tests)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
for (W, H, win, D) in ((3840, 2160, 7, 32), (1920, 1080, 9, 64), (7680, 4320, 21, 256)):
    t = time.time(); L, R = oracle.rds(3, W, H, 4, 20, 0); t1 = time.time() - t
    t = time.time(); e = oracle.box(L, R, win, D, oracle.RAW_LOC); t2 = time.time() - t
    print("%dx%d w%d D%d: rds %.3f s, box %.3f s, mean loc %.3f, cpus %d" % (W, H, win, D, t1, t2, e.mean(), os.cpu_count()), flush=True)
