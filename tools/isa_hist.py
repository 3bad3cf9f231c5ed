#!/usr/bin/env python3
"""Instruction histogram of one kernel in a device .s file (hipcc --cuda-device-only -S), per basic block.
   tools/isa_hist.py file.s 'sad_strip_kernelILi21ELi32' [min_block_size]"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ("entry", [])
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append(cur); cur = (m.group(1), []); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur[1].append(t.split()[0])
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(ins); tot.update(c)
    if len(ins) >= minb:
        valu = sum(n for k, n in c.items() if k.startswith("v_"))
        print(f"{name}: {len(ins)} instr, {valu} valu :: " + ", ".join(f"{k} {n}" for k, n in c.most_common(14)))
print("TOTAL", sum(tot.values()), "valu", sum(n for k, n in tot.items() if k.startswith("v_")))
