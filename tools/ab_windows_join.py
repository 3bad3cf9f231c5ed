#!/usr/bin/env python3
"""tools/ab_windows_join.py base.txt new.txt -- ratio new/base per (win, D)."""
import sys
def load(p):
    d = {}
    for l in open(p):
        f = l.split()
        if len(f) == 5: d[(int(f[1]), int(f[2]))] = (f[3], float(f[4]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
for k in sorted(a):
    if k in b: print("win %2d D %3d  %-36s %8.4f -> %8.4f  x%.3f" % (k[0], k[1], b[k][0], a[k][1], b[k][1], b[k][1] / a[k][1]))
