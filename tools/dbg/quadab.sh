# quad matcher A/B of experiment builds at D <= 32 (kernel ms, 8 launches)
for so in opencv_amd/csrc/libsadbm.so opencv_amd/csrc/libsadbm_base.so; do
SADBM_LIB=$PWD/$so timeout -k 10 200 python3 - <<'PY' || exit 1
import os, numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
res = []
for (W, H, pairs) in ((1920, 1080, 16), (3840, 2160, 4), (1242, 375, 64)):
    L, R = synth.rds(7, W, H, 8, 20, 0)
    dL = torch.from_numpy(np.stack([L] * pairs)).cuda(); dR = torch.from_numpy(np.stack([R] * pairs)).cuda()
    dO = torch.empty((pairs, H, W), dtype=torch.uint8, device="cuda")
    ctx = oa.Context(0)
    for win, D in ((7, 30), (9, 32), (15, 16)):
        for _ in range(2): oa.compute_batch_device(dL, dR, win, D, oa.KERNEL_QUAD, out=dO, ctx=ctx)
        torch.cuda.synchronize(); ctx.profile_begin(8)
        for _ in range(8): oa.compute_batch_device(dL, dR, win, D, oa.KERNEL_QUAD, out=dO, ctx=ctx)
        torch.cuda.synchronize(); n, m, _p = ctx.profile_end()
        res.append("%dx%dx%d w%d D%d %.4f" % (W, H, pairs, win, D, m / n))
print(os.path.basename(os.environ["SADBM_LIB"]), " | ".join(res), flush=True)
PY
done
