"""WIP: strip matcher parity (run by naming this file)."""
import numpy as np
import pytest
import opencv_amd as oa
import oracle
from helpers import noise_pair

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("win,D", [(15, 128), (7, 30), (9, 64), (7, 128), (21, 256), (4, 64), (5, 17), (6, 70), (8, 100), (12, 64), (13, 200),
                                   (14, 64), (16, 129), (17, 64), (20, 64), (25, 256), (10, 1), (11, 16)])
def test_strip_small(win, D):
    for (H, W) in ((61, 203), (70, 333), (40, 600)):
        for binary in (False, True):
            L, R = noise_pair(win * 977 + D + W, H, W, shift=5, binary=binary)
            exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
            ctx = oa.Context(0)
            got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
            assert ctx.last_kernel() == "sad_strip_kernel<%d>" % win
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (win, D, H, W, binary, len(bad), bad[:8], got[tuple(bad[0])], exp[tuple(bad[0])])
