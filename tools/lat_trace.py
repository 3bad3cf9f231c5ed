import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
for (W, H, ch, win, D) in ((384, 288, 3, 7, 30), (384, 288, 1, 7, 30), (640, 480, 1, 9, 64), (1920, 1080, 1, 9, 64)):
    L, R = synth.rds(5, W, H, 4, 20, 0)
    if ch > 1:
        L = np.ascontiguousarray(np.stack([L] * ch, -1)); R = np.ascontiguousarray(np.stack([R] * ch, -1))
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    ctx = oa.Context(0)
    for fw in ("0", "-1"):
        os.environ["SADBM_FASTWARM"] = fw
        oa.reload_env()
        for _ in range(20):
            oa.compute_device(dL, dR, win, D, 0, out=out, ctx=ctx)
        torch.cuda.synchronize()
