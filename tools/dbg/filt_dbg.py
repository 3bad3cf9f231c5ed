import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import opencv_amd as oa
from oracle import filters as F
rng = np.random.default_rng(11)
dog = rng.integers(0, 256, (361, 410, 3), dtype=np.uint8)
cat = rng.integers(0, 256, (361, 410, 3), dtype=np.uint8)
d = lambda a: torch.from_numpy(a).cuda()
k = oa.filters.GetgaussianKernel(31, 10)
dst_low = oa.filters.filter2D(d(dog), k); dst_tmp = oa.filters.filter2D(d(cat), k)
dst_high = oa.filters.highpass(d(cat), dst_tmp, add=12)
hyb = oa.filters.blend(dst_low, dst_high, 0.6)
r_low = F.filter2d(dog, k); r_tmp = F.filter2d(cat, k); r_high = F.highpass(cat, r_tmp, 12); r_h = F.blend(r_low, r_high, 0.6)
for n, a, b in (("low", dst_low, r_low), ("tmp", dst_tmp, r_tmp), ("high", dst_high, r_high), ("hyb", hyb, r_h)):
    a = a.cpu().numpy(); bad = np.argwhere(a != b)
    print(n, len(bad), bad[:3], [(int(a[tuple(x)]), int(b[tuple(x)])) for x in bad[:3]])
