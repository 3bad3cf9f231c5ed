"""sadbm_compute_device_static: the entry point that allocates nothing (static device scratch).  Called through the C
ABI and compared bit-for-bit with sadbm_compute_device, on two calls in a row, the second with a smaller input."""
import ctypes

import numpy as np
import pytest

import opencv_amd as oa
from helpers import noise_pair

pytestmark = pytest.mark.gpu


def test_static_entry_matches_compute_device_twice_second_smaller():
    import torch
    lib = oa.load_library()
    ctx = oa.Context(0)
    for (H, W, win, D, ch) in ((700, 1300, 15, 128, 1), (181, 407, 7, 30, 3), (150, 300, 60, 100, 1)):   # the last one: sad_wide_kernel
        L, R = noise_pair(H + W, H, W, shift=5, ch=ch)
        dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        ref = oa.compute_device(dL, dR, win, D, ctx=ctx)
        out = torch.full((H, W), 0xAB, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rc = lib.sadbm_compute_device_static(0, dL.data_ptr(), W * ch, dR.data_ptr(), W * ch, H, W, ch, win, D, 0, out.data_ptr(), W,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, (rc, lib.sadbm_last_hip_error())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref.cpu().numpy()), (H, W)


def test_static_entry_rejects_frames_beyond_its_scratch_and_fits_a_4k_pair():
    import torch
    lib = oa.load_library()
    H, W = 2400, 5000                                          # planes need 2 * 5504 * 2399 bytes > 19 MiB
    z = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    rc = lib.sadbm_compute_device_static(0, z.data_ptr(), W, z.data_ptr(), W, H, W, 1, 15, 128, 0, out.data_ptr(), W, None)
    assert rc == -8 and b"static" in lib.sadbm_strerror(rc)
    H, W = 2160, 3840
    import oracle
    L, R = oracle.rds(3, W, H, 16, 96, 0)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    got = oa.compute_device_static(dL, dR, 15, 128)
    assert np.array_equal(got.cpu().numpy(), oa.compute_device(dL, dR, 15, 128).cpu().numpy())
