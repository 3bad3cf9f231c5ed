import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import opencv_amd as oa, oracle
from helpers import noise_pair
win, D, H, W = [int(x) for x in sys.argv[1:5]]
L, R = noise_pair(win * 977 + D + W, H, W, shift=5, binary=False)
exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
got = oa.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), win, D, oa.RAW_LOC | oa.KERNEL_STRIP).cpu().numpy()
bad = got != exp
print("bad", bad.sum(), "of", bad.size)
print("bad per row:", bad.sum(1)[:20])
print("bad per col&3:", [int(bad[:, a::4].sum()) for a in range(4)])
print("bad per col/16:", [int(bad[:, s*16:(s+1)*16].sum()) for s in range((W+15)//16)])
for y in (0, 1, 20):
    print("row", y, "got", got[y, :48].tolist())
    print("row", y, "exp", exp[y, :48].tolist())
