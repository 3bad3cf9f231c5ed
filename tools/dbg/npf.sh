# per-window A/B of experiment builds: strip kernel ms at 4K x 4, D=128
for so in opencv_amd/csrc/libsadbm.so opencv_amd/csrc/libsadbm_npf1.so; do
SADBM_LIB=$PWD/$so timeout -k 10 200 python3 - <<'PY' || exit 1
import os, numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
W, H, pairs = 3840, 2160, 4
L, R = synth.rds(7, W, H, 8, 40, 0)
dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
ctx = oa.Context(0)
res = []
for win in (13, 14, 17, 19, 20, 21, 23, 25):
    for _ in range(2): oa.compute_batch_device(dL, dR, win, 128, oa.KERNEL_STRIP, out=dO, ctx=ctx)
    torch.cuda.synchronize(); ctx.profile_begin(10)
    for _ in range(10): oa.compute_batch_device(dL, dR, win, 128, oa.KERNEL_STRIP, out=dO, ctx=ctx)
    torch.cuda.synchronize(); n, m, _p = ctx.profile_end()
    res.append("w%d %.4f" % (win, m / n))
print(os.path.basename(os.environ["SADBM_LIB"]), " ".join(res), flush=True)
PY
done
