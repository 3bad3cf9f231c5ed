"""The static entry point under HIP graph capture (the callers it is for): captured once, replayed on new inputs."""
import numpy as np
import pytest

import opencv_amd as oa
import oracle
from helpers import noise_pair

pytestmark = pytest.mark.gpu


def test_static_entry_is_capturable_and_replays_on_new_inputs():
    import torch
    H, W, win, D = 300, 700, 9, 64
    L, R = noise_pair(1, H, W, shift=4)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    oa.compute_device_static(dL, dR, win, D, out=out)          # first call on the device resolves the scratch address
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    out.zero_()
    with torch.cuda.graph(g, stream=s):
        oa.compute_device_static(dL, dR, win, D, out=out)
    for seed in (2, 3):
        L2, R2 = noise_pair(seed, H, W, shift=3 + seed)
        dL.copy_(torch.from_numpy(L2).cuda()); dR.copy_(torch.from_numpy(R2).cuda())
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle.box(L2, R2, win, D)), seed
