#!/usr/bin/env python3
"""Register / spill table of the strip kernels: compiles strip_kernels.hip with -Rpass-analysis=kernel-resource-usage.
   tools/regs.py [--otf] [extra hipcc flags]"""
import re, subprocess, sys, os
src = os.path.join(os.path.dirname(__file__), "..", "opencv_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-pthread", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", "-o", "/dev/null"]
otf = "--otf" in sys.argv
extra = [a for a in sys.argv[1:] if a != "--otf"]
if otf:
    out = "".join(subprocess.run(cmd + ["-DSADBM_OTF_PART_NG=%d" % n, "strip_otf_kernels.hip"] + extra, cwd=src, capture_output=True, text=True).stderr for n in (1, 2, 4, 8))
else:
    out = "".join(subprocess.run(cmd + ["-DSADBM_STRIP_PART_NG=%d" % n, "strip_kernels.hip"] + extra, cwd=src, capture_output=True, text=True).stderr for n in (4, 8, 16))
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    m = re.search(r"sad_strip_kernelILi(\d+)ELi(\d+)E", name)
    if not m:
        m = re.search(r"sad_strip_otf_kernelILi(\d+)ELi(\d+)ELb(\d)ELi(\d)E", name)
        if m:
            print(f"otf<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)}> vgpr {r.get('VGPRs')} sgpr {r.get('SGPRs')} scratch {r.get('ScratchSize')} sgpr_spill {r.get('SGPRs Spill')} vgpr_spill {r.get('VGPRs Spill')} occ {r.get('Occupancy')}")
        continue
    print(f"strip<{m.group(1)},{m.group(2)}> vgpr {r.get('VGPRs')} sgpr {r.get('SGPRs')} scratch {r.get('ScratchSize')} sgpr_spill {r.get('SGPRs Spill')} vgpr_spill {r.get('VGPRs Spill')} occ {r.get('Occupancy')}")
