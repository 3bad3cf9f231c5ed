"""Turns a dump of tools/dbg/abrt_trace.c into candidate frames: every stack word that points into an executable mapping is printed as
library+offset with the nearest preceding dynamic symbol (nm -D / nm) of the same library in THIS container (same image as the GPU box).
    python tools/dbg/resolve_trace.py gpurun_out/abrt.txt"""
import bisect
import os
import subprocess
import sys

lines = open(sys.argv[1]).read().splitlines()
i_maps = lines.index("MAPS")
maps = []
for l in lines[i_maps + 1:]:
    f = l.split()
    if len(f) >= 6 and "x" in f[1]:
        a, b = (int(x, 16) for x in f[0].split("-"))
        maps.append((a, b, int(f[2], 16), f[5]))
syms = {}


def table(path):
    if path not in syms:
        real = path.replace("/tmp/code/Cd297916757__opencv/repo", os.getcwd())
        t = []
        for flag in ("-D", ""):
            try:
                out = subprocess.run(["nm", "-C", "--defined-only"] + ([flag] if flag else []) + [real], capture_output=True, text=True).stdout
            except Exception:
                out = ""
            for s in out.splitlines():
                p = s.split(None, 2)
                if len(p) == 3 and p[1] in "TtWw":
                    t.append((int(p[0], 16), p[2]))
        t.sort()
        syms[path] = t
    return syms[path]


def resolve(v):
    for a, b, off, path in maps:
        if a <= v < b:
            fo = v - a + off
            t = table(path)
            name = ""
            if t:
                k = bisect.bisect_right([x[0] for x in t], fo) - 1
                if k >= 0:
                    name = "%s+0x%x" % (t[k][1], fo - t[k][0])
            return "%s+0x%x  %s" % (os.path.basename(path), fo, name)
    return None


for tag in ("RIP",):
    v = int(lines[lines.index(tag) + 1], 16)
    print(tag, hex(v), resolve(v))
print(lines[0], "ADDR", lines[lines.index("ADDR") + 1])
i_st = lines.index("STACK")
for n, l in enumerate(lines[i_st + 1:i_maps]):
    v = int(l, 16)
    r = resolve(v)
    if r:
        print("sp+%5d  %s" % (8 * n, r))
