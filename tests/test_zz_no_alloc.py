"""sadbm_compute_device_static: the entry point that allocates nothing (static device scratch).  Called through the C
ABI and compared bit-for-bit with sadbm_compute_device (and with the oracle), on two calls in a row, the second with a
smaller input."""
import ctypes

import numpy as np
import pytest

import opencv_amd as oa
import oracle
from helpers import noise_pair

pytestmark = pytest.mark.gpu


def test_static_entry_matches_compute_device_twice_second_smaller():
    import torch
    lib = oa.load_library()
    ctx = oa.Context(0)
    # grey frames and BGR frames at the reference's parameters are read in place (no scratch at all); the static scratch planes are
    # what BGR frames under the staged kernel (third case), BGRA frames and the wide-window matcher (last case) go through
    for (H, W, win, D, ch) in ((700, 1300, 15, 128, 1), (181, 407, 7, 30, 3), (420, 900, 15, 128, 3), (300, 640, 9, 64, 4), (150, 300, 60, 100, 1)):
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
        p0 = (lambda a: np.ascontiguousarray(a[..., 0])) if ch > 1 else (lambda a: a)
        assert np.array_equal(out.cpu().numpy(), oracle.box(p0(L), p0(R), win, D)), (H, W, "oracle")


def test_static_entry_rejects_frames_beyond_its_scratch_and_fits_a_4k_pair():
    import torch
    lib = oa.load_library()
    H, W = 2400, 5000                                          # planes need 2 * 5504 * 2399 bytes > 19 MiB
    z = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")   # BGR: needs the plane-extraction pass, hence the scratch
    out = torch.full((H, W), 0xAB, dtype=torch.uint8, device="cuda")
    rc = lib.sadbm_compute_device_static(0, z.data_ptr(), 3 * W, z.data_ptr(), 3 * W, H, W, 3, 15, 128, 0, out.data_ptr(), W, None)
    assert rc == -8 and b"static" in lib.sadbm_strerror(rc)
    torch.cuda.synchronize()
    assert int((out != 0xAB).sum().item()) == 0                # refused before anything was launched: the output is untouched
    # a single-channel frame of the same size that the matcher reads in place needs no scratch: not refused
    g = np.random.default_rng(5)
    L1 = g.integers(0, 256, (300, W), dtype=np.uint8)
    R1 = np.roll(L1, -9, axis=1)
    L1b, R1b = np.tile(L1, (8, 1)), np.tile(R1, (8, 1))         # 2400 x 5000
    got = oa.compute_device_static(torch.from_numpy(L1b).cuda(), torch.from_numpy(R1b).cuda(), 7, 30)
    assert np.array_equal(got.cpu().numpy(), oracle.box(L1b, R1b, 7, 30))
    H, W = 2160, 3840
    L, R = oracle.rds(3, W, H, 16, 96, 0)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    got = oa.compute_device_static(dL, dR, 15, 128)
    assert np.array_equal(got.cpu().numpy(), oa.compute_device(dL, dR, 15, 128).cpu().numpy())
