"""Whole-output parity at the BASELINE configurations' full sizes (configs 3, 4, 5): every byte of the GPU result
is compared with the sliding-sum oracle (oracle.box) -- no bands, no crops.  Calls go through the C ABI."""
import numpy as np
import pytest

import opencv_amd as oa
import oracle

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def assert_same(got, exp, what):
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError("%s: %d bytes differ, first at %s (got %d, expected %d)"
                             % (what, len(bad), tuple(bad[0]), got[tuple(bad[0])], exp[tuple(bad[0])]))


def test_config3_whole_4k_frame():
    """3840x2160, 15x15, D=128: rds(3, ...) of SURVEY 8(d), raw loc and the encoded output, whole frame."""
    W, H, win, D = 3840, 2160, 15, 128
    L, R = oracle.rds(3, W, H, 16, 96, 0)
    exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
    ctx = oa.Context(0)
    got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel() == "sad_strip_kernel<15>+direct"      # single-channel device frames are read in place
    assert_same(got, exp, "cfg3 raw")
    enc = oa.compute_device(dev(L), dev(R), win, D, ctx=ctx).cpu().numpy()
    assert_same(enc, ((exp.astype(np.int32) * 16) & 255).astype(np.uint8), "cfg3 encoded")    # SAD.cpp:57


def test_config4_whole_8k_frame_as_8_row_bands():
    """7680x4320, 21x21, D=256 computed as the 8 row bands of BASELINE config 4 (540 rows + 20-row downward halo each)
    through scatter_band / compute_band_device on one GPU, concatenated, against the oracle on the whole frame."""
    W, H, win, D, world = 7680, 4320, 21, 256, 8
    L, R = oracle.rds(4, W, H, 32, 200, 0)
    exp = oracle.box(L, R, win, D)
    parts = []
    for rank in range(world):
        (r0, r1, per), lb, rb = oa.scatter_band(L, R, rank, world, win)
        assert per == 540 and lb.shape[0] == (560 if rank < world - 1 else 540)
        parts.append(oa.compute_band_device(dev(lb), dev(rb), H, r0, r1 - r0, win, D).cpu().numpy())
    assert_same(np.concatenate(parts, 0), exp, "cfg4 bands")


def test_config4_whole_8k_frame_single_call():
    W, H, win, D = 7680, 4320, 21, 256
    L, R = oracle.rds(4, W, H, 32, 200, 0)
    got = oa.compute_device(dev(L), dev(R), win, D).cpu().numpy()
    assert_same(got, oracle.box(L, R, win, D), "cfg4 single call")


@pytest.mark.parametrize("win", [7, 9])
def test_config5_kitti_batch_of_64(win):
    """1242x375, D=128, a batch of 64 distinct pairs (seeds and planted disparities of SURVEY 8(d)), w = 7 and 9."""
    W, H, D, n = 1242, 375, 128, 64
    pairs = [oracle.rds(1000 + i, W, H, 5 + i % 16, 60 + i % 64, 0) for i in range(n)]
    out = oa.compute_batch_device(dev(np.stack([p[0] for p in pairs])), dev(np.stack([p[1] for p in pairs])), win, D).cpu().numpy()
    for i, (L, R) in enumerate(pairs):
        assert_same(out[i], oracle.box(L, R, win, D), "cfg5 w%d pair %d" % (win, i))


def test_config2_binary_dots_whole_frame():
    """1920x1080, 9x9, D=64 on binary dots (heavy cost ties: the first-minimum rule decides most pixels)."""
    L, R = oracle.rds(2, 1920, 1080, 8, 40, 1)
    got = oa.compute_device(dev(L), dev(R), 9, 64, oa.RAW_LOC).cpu().numpy()
    assert_same(got, oracle.box(L, R, 9, 64, oracle.RAW_LOC), "cfg2 binary")


def test_reference_parameters_whole_1080p_and_4k_frames():
    """winSize 7 / DSR 30 (SAD.cpp:9,11) and D = 32 on whole 1080p and 4K frames: the on-the-fly strip kernels, through the
    prep pass and reading the frames directly, grey dots and binary dots."""
    for (W, H, D, seed, binary) in ((1920, 1080, 30, 2, 0), (3840, 2160, 32, 3, 0), (1920, 1080, 30, 12, 1)):
        L, R = oracle.rds(seed, W, H, 4, 20, binary)
        exp = oracle.box(L, R, 7, D, oracle.RAW_LOC)
        ctx = oa.Context(0)
        got = oa.compute_device(dev(L), dev(R), 7, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == "sad_strip_otf_kernel<7,2>+direct"
        assert_same(got, exp, "w7 D%d %dx%d" % (D, W, H))
        L3 = np.stack([L, R, L], -1); R3 = np.stack([R, L, R], -1)       # BGR: channel 0 gathered by the matcher itself (round 3) ...
        got = oa.compute_device(dev(L3), dev(R3), 7, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == "sad_strip_otf_kernel<7,2>+direct"
        assert_same(got, exp, "w7 D%d %dx%d BGR" % (D, W, H))
        L4 = np.stack([L, R, L, R], -1); R4 = np.stack([R, L, R, L], -1)  # ... BGRA: through the prep pass
        got = oa.compute_device(dev(L4), dev(R4), 7, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == "sad_strip_otf_kernel<7,2>"
        assert_same(got, exp, "w7 D%d %dx%d BGRA" % (D, W, H))


def test_config1_batch_of_256_and_config2_batch_whole_frames():
    """BASELINE config 1's frame size (384x288, 7x7, D=30) as a batch of 256 distinct pairs (frame groups), and config 2
    (1920x1080, 9x9, D=64) as a batch of 8: every byte of every frame."""
    n = 256
    pairs = [oracle.rds(100 + i, 384, 288, 1 + i % 7, 8 + i % 20, i & 1) for i in range(n)]
    out = oa.compute_batch_device(dev(np.stack([p[0] for p in pairs])), dev(np.stack([p[1] for p in pairs])), 7, 30).cpu().numpy()
    for i, (L, R) in enumerate(pairs):
        assert_same(out[i], oracle.box(L, R, 7, 30), "cfg1 pair %d" % i)
    pairs = [oracle.rds(20 + i, 1920, 1080, 8 + i, 40 + i, 0) for i in range(8)]
    ctx = oa.Context(0)
    out = oa.compute_batch_device(dev(np.stack([p[0] for p in pairs])), dev(np.stack([p[1] for p in pairs])), 9, 64, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel() == "sad_strip_otf_kernel<9,4>+direct"  # read in place like every single-channel device job (late round 3)
    for i, (L, R) in enumerate(pairs):
        assert_same(out[i], oracle.box(L, R, 9, 64), "cfg2 pair %d" % i)
