"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Every call goes through the C ABI of
include/sadbm.h (ctypes -> libsadbm.so -> HIP kernels) and is compared bit-for-bit with the CPU
oracle and the committed golden fixtures.  Bar: bit-exact (integer/byte path)."""
import numpy as np
import pytest

import opencv_amd as oa
import oracle
from helpers import GOLDEN, noise_pair, read_pnm, sha, tsukuba_bgr

pytestmark = pytest.mark.gpu


def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(a):
    return torch_cuda().from_numpy(np.ascontiguousarray(a)).cuda()


def test_library_is_the_hip_build():
    assert oa.load_library().sadbm_version() == 200


def test_config1_tsukuba_golden_host_api():
    """BASELINE config 1 through sadbm_get_disparity (host buffers, BGR, channel 0 on device)."""
    L, R = tsukuba_bgr()
    out = oa.getDisparity(L, R)                                   # defaults winSize=7, DSR=30 (SAD.cpp:9,11)
    assert np.array_equal(out, read_pnm(GOLDEN + "/tsukuba_disp_w7_d30.pgm"))
    assert sha(out) == "7da39f56141fbb2fc045c375b5be9fe84e8620beabec3009c99a5194075ea166"
    raw = oa.get_disparity_host(L, R, 7, 30, oa.RAW_LOC)
    assert np.array_equal(raw, read_pnm(GOLDEN + "/tsukuba_loc_w7_d30.pgm"))
    gen = oa.get_disparity_host(L, R, 7, 30, oa.FORCE_GENERIC)
    assert np.array_equal(gen, out)


@pytest.mark.parametrize("tag,binary", [("u8", 0), ("bin", 1)])
def test_rds_golden_device_api(tag, binary):
    L, R = oracle.rds(1, 320, 240, 4, 20, binary)
    out = oa.compute_device(dev(L), dev(R), 9, 64).cpu().numpy()
    assert np.array_equal(out, read_pnm(GOLDEN + "/rds_seed1_320x240_w9_d64_%s_disp.pgm" % tag))


@pytest.mark.parametrize("win", [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 21, 25, 27])
@pytest.mark.parametrize("D", [1, 16, 30, 64, 128, 256])
def test_sweep_matches_oracle(win, D):
    H, W = 61, 203                       # odd sizes, not multiples of any tile
    for binary in (False, True):
        L, R = noise_pair(win * 977 + D, H, W, shift=5, binary=binary)
        exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
        got = oa.get_disparity_host(L, R, win, D, oa.RAW_LOC)
        assert np.array_equal(got, exp), (win, D, binary, np.argwhere(got != exp)[:5])
        got = oa.get_disparity_host(L, R, win, D, oa.RAW_LOC | oa.KERNEL_GLOBAL)     # L1-operand variant
        assert np.array_equal(got, exp), ("global", win, D, binary, np.argwhere(got != exp)[:5])


@pytest.mark.parametrize("kc", ["8", "16"])
@pytest.mark.parametrize("win,D", [(7, 30), (9, 64), (15, 128), (21, 256), (16, 200), (3, 17), (1, 1)])
def test_both_disparity_chunkings(kc, win, D, monkeypatch):
    """The matcher is instantiated for 8 and 16 disparities per thread (chosen by D); force each."""
    monkeypatch.setenv("SADBM_KC", kc)
    oa.reload_env()                                  # the library reads its knobs once
    L, R = noise_pair(win + D, 70, 333, shift=9)
    exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
    ctx = oa.Context(0)
    got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_QUAD, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel() == "sad_quad_lds_kernel<%d,%s>" % (win, kc)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("half", ["0", "1", None])
def test_quad_half_width_blocks(half, monkeypatch):
    """The quad matcher runs 256-thread blocks or, for frames whose column quads fill them better (384 = 3 x 32 quads),
    half-width 128-thread blocks with their own prefetch roles; force each (and the automatic choice) over widths on
    both sides of the rule, single frames, batches, row bands, both chunkings and the border-heavy W < D case."""
    if half is None:
        monkeypatch.delenv("SADBM_QUAD_HALF", raising=False)
    else:
        monkeypatch.setenv("SADBM_QUAD_HALF", half)
    oa.reload_env()
    for (H, W) in ((45, 384), (33, 129), (40, 640), (30, 1920), (37, 333), (35, 100)):
        L, R = noise_pair(7 * W + H, H, W, shift=6)
        for win, D in ((7, 30), (9, 32), (5, 16), (3, 8), (11, 64), (15, 128), (21, 256), (25, 200), (7, 1)):
            exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
            got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_QUAD).cpu().numpy()
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (half, H, W, win, D, len(bad), bad[:6])
    # a batch of reference-sized frames (the case the half-width blocks are for) and a row band of a taller frame
    Ls, Rs = zip(*[noise_pair(900 + i, 60, 384, shift=2 + i) for i in range(5)])
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), 7, 30, oa.KERNEL_QUAD).cpu().numpy()
    for i in range(5):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], 7, 30)), i
    L, R = noise_pair(77, 150, 384, shift=4)
    exp = oracle.box(L, R, 7, 30)
    ctx = oa.Context(0)
    band = oa.compute_band_device(dev(L[50:116]), dev(R[50:116]), 150, 50, 60, 7, 30, oa.KERNEL_QUAD, ctx=ctx).cpu().numpy()   # 60 rows + win-1 halo
    assert np.array_equal(band, exp[50:110])


@pytest.mark.parametrize("shape", [(9, 300), (300, 9), (40, 17), (8, 8), (7, 50), (50, 7), (1, 1), (33, 129)])
def test_small_and_degenerate_shapes(shape):
    H, W = shape
    L, R = noise_pair(H * 31 + W, H, W)
    for win, D in ((7, 30), (3, 64), (9, 256)):
        exp = oracle.literal(L, R, win, D)
        got = oa.get_disparity_host(L, R, win, D)
        assert np.array_equal(got, exp), (shape, win, D)


def test_w_less_than_d_and_constant_and_identical():
    L, R = noise_pair(4, 30, 40)
    assert np.array_equal(oa.get_disparity_host(L, R, 5, 128), oracle.literal(L, R, 5, 128))
    c1 = np.full((30, 90), 10, np.uint8)
    c2 = np.full((30, 90), 200, np.uint8)
    assert np.array_equal(oa.get_disparity_host(c1, c2, 5, 16, oa.RAW_LOC), oracle.literal(c1, c2, 5, 16, oracle.RAW_LOC))
    assert not oa.get_disparity_host(L, L, 7, 30, oa.RAW_LOC).any()           # K1


@pytest.mark.parametrize("ch", [3, 4])
def test_multichannel_uses_plane0(ch):
    L, R = noise_pair(17 + ch, 50, 77, ch=ch)
    exp = oracle.literal(np.ascontiguousarray(L[:, :, 0]), np.ascontiguousarray(R[:, :, 0]), 7, 30)
    assert np.array_equal(oa.get_disparity_host(L, R, 7, 30), exp)
    assert np.array_equal(oa.compute_device(dev(L), dev(R), 7, 30).cpu().numpy(), exp)


def test_roi_views_with_step_and_larger_right():
    bl, br = noise_pair(23, 80, 160)
    L = bl[10:70, 16:140]
    R = br[10:70, 16:140]
    exp = oracle.literal(np.ascontiguousarray(L), np.ascontiguousarray(R), 7, 30)
    assert np.array_equal(oa.get_disparity_host(L, R, 7, 30), exp)
    # a larger right image is legal; its extra area is ignored (SAD.cpp:26-27 reads only left's size)
    Rbig = br[10:75, 16:150]
    assert np.array_equal(oa.get_disparity_host(L, Rbig, 7, 30), exp)
    # device tensors with a row pitch and an unaligned output view
    torch = torch_cuda()
    dl = dev(bl)[10:70, 16:140]
    dr = dev(br)[10:70, 16:140]
    obuf = torch.zeros((60, 131), dtype=torch.uint8, device="cuda")
    out = obuf[:, 3:127]
    oa.compute_device(dl, dr, 7, 30, out=out)
    assert np.array_equal(out.cpu().numpy(), exp)
    assert not obuf[:, :3].any() and not obuf[:, 127:].any()


def test_errors_match_reference_failure_modes():
    L, R = noise_pair(2, 20, 20)
    with pytest.raises(oa.SadbmError) as e:
        oa.get_disparity_host(L, R[:19], 7, 30)
    assert e.value.code == -3
    with pytest.raises(oa.SadbmError) as e:
        oa.get_disparity_host(np.stack([L] * 3, -1), R, 7, 30)
    assert e.value.code == -2
    with pytest.raises(oa.SadbmError) as e:
        oa.get_disparity_host(L, R, 0, 30)
    assert e.value.code == -4
    with pytest.raises(oa.SadbmError) as e:                      # raw loc output is a byte; the reference's encoding has no such limit
        oa.get_disparity_host(L, R, 7, 257, oa.RAW_LOC)
    assert e.value.code == -4


def test_batch_matches_per_pair():
    torch = torch_cuda()
    n, H, W = 5, 47, 131
    Ls, Rs = zip(*[oracle.rds(1000 + i, W, H, 5 + i, 20 + i, 0) for i in range(n)])
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), 7, 64).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], 7, 64)), i


@pytest.mark.parametrize("nb", [1, 2, 3, 8])
def test_row_bands_concatenate_to_full_frame(nb):
    H, W, win, D = 83, 140, 9, 40
    L, R = noise_pair(77, H, W, shift=4)
    exp = oracle.box(L, R, win, D)
    per, ranges = oa.band_ranges(H, nb)
    parts = []
    for (r0, r1) in ranges:
        if r1 == r0:
            continue
        i0, i1 = oa.band_input_rows(r0, r1, H, win)
        parts.append(oa.compute_band_device(dev(L[i0:i1]), dev(R[i0:i1]), H, r0, r1 - r0, win, D).cpu().numpy())
    assert np.array_equal(np.concatenate(parts, 0), exp)


def test_repeatability_is_bit_exact():
    L, R = oracle.rds(9, 640, 360, 8, 40, 0)
    a = oa.compute_device(dev(L), dev(R), 9, 64).cpu().numpy()
    for _ in range(3):
        assert np.array_equal(a, oa.compute_device(dev(L), dev(R), 9, 64).cpu().numpy())


def test_full_size_1080p_config2_against_box_oracle():
    """BASELINE config 2 at full size: rds(2,1920,1080,8,40), 9x9, D=64 vs the sliding-sum oracle."""
    L, R = oracle.rds(2, 1920, 1080, 8, 40, 0)
    got = oa.compute_device(dev(L), dev(R), 9, 64, oa.RAW_LOC).cpu().numpy()
    exp = oracle.box(L, R, 9, 64, oracle.RAW_LOC)
    assert np.array_equal(got, exp)
    # size-independent property: interior of background/foreground recovers the planted disparities
    assert (got[100, 100:300] == 8).all() and (got[500, 700:1100] == 40).all()


def test_disparity_row_bands_world1_uses_hip_band_path():
    """The band + all-gather host function on one GPU (world=1): same code as the N-GPU path minus the
    collective; the 2-rank collective itself is covered on CPU with gloo (tests/test_multi_cpu.py)."""
    H, W, win, D = 97, 150, 15, 48
    L, R = noise_pair(5, H, W, shift=6)
    (r0, r1, per), lb, rb = oa.scatter_band(dev(L), dev(R), 0, 1, win)
    full = oa.disparity_row_bands(lb, rb, H, win, D, 0, 1)
    assert np.array_equal(full.cpu().numpy(), oracle.box(L, R, win, D))


def test_cfg4_band_geometry_matches_oracle():
    """BASELINE config 4 geometry: 4320 rows, 21x21, D=256, 8 bands of 540 rows + 20-row downward halo
    (column crop 1024 to keep the checker fast).  Band 3 (interior) and band 7 (holds the zero frame)."""
    H, W, win, D = 4320, 1024, 21, 256
    L, R = oracle.rds(4, W, H, 32, 200, 0)
    per, ranges = oa.band_ranges(H, 8)
    assert per == 540
    for b in (3, 7):
        r0, r1 = ranges[b]
        i0, i1 = oa.band_input_rows(r0, r1, H, win)
        assert i1 - i0 == (560 if b == 3 else 540)
        got = oa.compute_band_device(dev(L[i0:i1]), dev(R[i0:i1]), H, r0, r1 - r0, win, D).cpu().numpy()
        nvalid = min(r1 - r0, H - win - r0)
        # sliding-sum oracle on the band's rows; one spare row appended so its own i < rows-win bound
        # (SAD.cpp:34) admits every row the halo supports
        pad = np.zeros((1, W), np.uint8)
        Lb = np.concatenate([L[i0:i1], pad]) if b == 3 else L[i0:i1]
        Rb = np.concatenate([R[i0:i1], pad]) if b == 3 else R[i0:i1]
        exp = oracle.box(Lb, Rb, win, D)[: r1 - r0]
        assert np.array_equal(got, exp), b
        if b == 7:
            assert nvalid == 540 - 21 and not got[nvalid:].any()


def test_full_size_4k_config3_properties_and_band_of_oracle():
    """BASELINE config 3 at full size (3840x2160, 15x15, D=128).  Whole-frame checks are size-independent
    properties; a 200-row band is compared bit-for-bit with the sliding-sum oracle."""
    W, H, win, D = 3840, 2160, 15, 128
    L, R = oracle.rds(3, W, H, 16, 96, 0)
    got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC).cpu().numpy()
    # zero frame (SAD.cpp:31,34,36)
    assert not got[H - win:, :].any() and not got[:, W - win:].any()
    # left border ramp (SAD.cpp:40,44): loc = j+1 for j < d_bg, then the planted disparity is found exactly
    assert (got[10, :16] == np.arange(1, 17)).all()
    assert (got[100, 200:900] == 16).all() and (got[1000, 1200:2700] == 96).all()
    # the planted disparity dominates: background rows are d_bg, foreground box is d_fg
    assert (got[: H // 4 - win, 16: W - win - 16] == 16).mean() > 0.999
    # idempotence / determinism
    again = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC).cpu().numpy()
    assert np.array_equal(got, again)
    # band 500..700 against the oracle (needs input rows 500..715)
    r0, n = 500, 200
    exp = oracle.box(L[r0: r0 + n + win], R[r0: r0 + n + win], win, D, oracle.RAW_LOC)[:n]
    assert np.array_equal(got[r0: r0 + n], exp)
    # encode relation (SAD.cpp:57): default output == (loc*16)&255
    enc = oa.compute_device(dev(L), dev(R), win, D).cpu().numpy()
    assert np.array_equal(enc, (got.astype(np.int32) * 16 & 255).astype(np.uint8))


def test_config5_kitti_batch_against_oracle():
    """BASELINE config 5 geometry: 1242x375 pairs, D=128, 7x7; a batch of 6 distinct pairs."""
    W, H, win, D = 1242, 375, 7, 128
    pairs = [oracle.rds(1000 + n, W, H, 5 + n % 16, 60 + n % 64, 0) for n in range(6)]
    out = oa.compute_batch_device(dev(np.stack([p[0] for p in pairs])), dev(np.stack([p[1] for p in pairs])), win, D).cpu().numpy()
    for n, (L, R) in enumerate(pairs):
        assert np.array_equal(out[n], oracle.box(L, R, win, D)), n


@pytest.mark.parametrize("win", [26, 27, 29, 31, 32, 33, 40, 48, 49, 50])
def test_windows_beyond_the_strip_matchers(win):
    """Windows wider than 25 columns have no strip instantiation: the quad matcher serves them up to 49 (32-bit costs), the
    column-sum matcher (sad_wide_kernel) beyond -- frames around the block sizes, batches, binary dots, BGR, D not a multiple of 8."""
    import torch
    from helpers import kname
    for (H, W, n, D, binary) in ((win + 5, 300, 1, 30, False), (win + 9, 517, 3, 64, True), (win + 2, win + 70, 2, 200, False), (70, 260, 1, 13, False)):
        Ls, Rs = zip(*[noise_pair(win * 31 + D + i, H, W, shift=min(7 + i, D - 1), binary=binary) for i in range(n)])
        ctx = oa.Context(0)
        out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
        assert kname(ctx) == ("sad_quad_lds_kernel<%d,%d>" % (win, 8 if D <= 64 else 16) if win <= 49 else "sad_wide_kernel")
        for i in range(n):
            exp = oracle.box(Ls[i], Rs[i], win, D, oracle.RAW_LOC)
            assert np.array_equal(out[i], exp), (win, H, W, D, i, np.argwhere(out[i] != exp)[:4])
    L, R = noise_pair(win, win + 20, 200, shift=4, ch=3)
    assert np.array_equal(oa.get_disparity_host(L, R, win, 30), oracle.literal(np.ascontiguousarray(L[:, :, 0]), np.ascontiguousarray(R[:, :, 0]), win, 30))


@pytest.mark.parametrize("win", [51, 64, 97, 128, 129, 200, 254, 255])
def test_wide_windows_column_sum_matcher(win):
    """sad_wide_kernel, windows of 50..255 columns (the API's maximum): frames one column wider than the window, frames that
    span several column blocks (257 - win output columns each) and row bands (32 rows), batches, every D class (below 8, not a
    multiple of 8, 256), saturated costs (binary dots: cost up to win^2 * 255 < 2^24), against the oracle and the brute force."""
    from helpers import kname
    for (H, W, n, D, binary) in ((win + 1, win + 1, 1, 5, False), (win + 40, win + 300, 2, 30, True), (win + 33, 2 * (257 - win) + win + 3, 1, 256, False),
                                  (win + 7, win + 130, 3, 67, True)):
        Ls, Rs = zip(*[noise_pair(win * 13 + D + i, H, W, shift=min(5 + i, D - 1, W - 1), binary=binary) for i in range(n)])
        ctx = oa.Context(0)
        dL, dR = dev(np.stack(Ls)), dev(np.stack(Rs))
        out = oa.compute_batch_device(dL, dR, win, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
        assert kname(ctx) == "sad_wide_kernel"
        gen = oa.compute_batch_device(dL, dR, win, D, oa.RAW_LOC | oa.FORCE_GENERIC, ctx=ctx).cpu().numpy()
        assert kname(ctx) == "sad_generic_kernel"
        for i in range(n):
            exp = oracle.box(Ls[i], Rs[i], win, D, oracle.RAW_LOC)
            assert np.array_equal(out[i], exp), (win, H, W, D, i, np.argwhere(out[i] != exp)[:4])
            assert np.array_equal(gen[i], exp), (win, H, W, D, i)
    # a row band of a larger frame (what one GPU of sadbm_compute_multi computes), the bottom band included: zero rows below rows - win
    H, W, D = win + 50, win + 90, 40
    L, R = noise_pair(win, H, W, shift=6)
    full = oracle.box(L, R, win, D)
    for r0, nb in ((0, 20), (20, 17), (37, H - 37)):
        i1 = min(r0 + nb + win - 1, H)
        got = oa.compute_band_device(dev(L[r0:i1]), dev(R[r0:i1]), H, r0, nb, win, D).cpu().numpy()
        assert np.array_equal(got, full[r0:r0 + nb]), (win, r0, nb)


@pytest.mark.parametrize("D", [257, 300, 512, 1000])
def test_search_ranges_beyond_256_disparities(D):
    """SAD.cpp:11 `int DSR` has no upper limit and SAD.cpp:57 stores (loc*16) mod 256: DSR > 256 is a legal input (SURVEY b2 limits only
    the raw-loc output).  The column-sum matcher serves it for every window up to 255 columns: 64-disparity passes merged by cost,
    keys carry k mod 256.  Against oracle.literal (and oracle.box on the larger frames): frames wider and narrower than DSR, binary dots
    (cost ties across the 256-disparity boundary: the earlier disparity must win), batches, a row band, BGR, the host API."""
    from helpers import kname
    for (H, W, win, binary, shift) in ((23, 700, 7, False, 290), (30, 1100, 9, True, 300), (19, 420, 15, True, 263), (40, 600, 21, False, 5), (12, 280, 4, True, 270),
                                       (70, 640, 51, True, 259)):
        L, R = noise_pair(D + win, H, W, shift=min(shift, W - 1), binary=binary)
        exp = oracle.literal(L, R, win, D) if H * W <= 30000 else oracle.box(L, R, win, D)
        ctx = oa.Context(0)
        got = oa.compute_device(dev(L), dev(R), win, D, ctx=ctx).cpu().numpy()
        assert kname(ctx) == "sad_wide_kernel"
        assert np.array_equal(got, exp), (D, H, W, win, np.argwhere(got != exp)[:4])
        assert np.array_equal(oa.get_disparity_host(L, R, win, D), exp), ("host", D, H, W, win)
    # constant images: every valid cost equal, loc = j + 1 up to DSR - 1 and 0 beyond (K2), encoded mod 256
    c1, c2 = np.full((12, D + 40), 7, np.uint8), np.full((12, D + 40), 9, np.uint8)
    got = oa.compute_device(dev(c1), dev(c2), 5, D).cpu().numpy()
    exp = np.zeros_like(c1)
    for j in range(D + 40 - 5):
        exp[:12 - 5, j] = (((j + 1) if j + 1 < D else 0) * 16) & 255
    assert np.array_equal(got, exp)
    # batch + band + BGR
    Ls, Rs = zip(*[noise_pair(900 + i, 33, 530, shift=280 + i, binary=True) for i in range(3)])
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), 9, D).cpu().numpy()
    for i in range(3):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], 9, D)), i
    full = oracle.box(Ls[0], Rs[0], 9, D)
    got = oa.compute_band_device(dev(Ls[0][10:33]), dev(Rs[0][10:33]), 33, 10, 23, 9, D).cpu().numpy()
    assert np.array_equal(got, full[10:33])
    L3, R3 = noise_pair(77, 20, 400, shift=300, ch=3)
    assert np.array_equal(oa.get_disparity_host(L3, R3, 7, D), oracle.literal(np.ascontiguousarray(L3[:, :, 0]), np.ascontiguousarray(R3[:, :, 0]), 7, D))


@pytest.mark.parametrize("win", [256, 257, 260, 300])
def test_windows_beyond_255_columns(win):
    """SAD.cpp:9 `int winSize` has no upper limit either.  Windows beyond 255 columns run the brute-force matcher with 64-bit keys; from
    257 x 257 on the reference's cost passes through `float sub_Sum` inexactly (SAD.cpp:10,20-21: 257^2 * 255 > 2^24) -- binary dots
    drive costs up there -- and oracle.literal, which keeps the float, is the oracle."""
    from helpers import kname
    for (H, W, D, binary) in ((win + 3, win + 9, 6, True), (win + 2, win + 5, 300, False)):
        L, R = noise_pair(win + D, H, W, shift=2, binary=binary)
        if binary:                                              # make the windows differ almost everywhere: costs near win^2 * 255
            R = np.where(L > 0, 0, 255).astype(np.uint8)
            R[::7, ::5] = L[::7, ::5]
        exp = oracle.literal(L, R, win, D)
        ctx = oa.Context(0)
        got = oa.compute_device(dev(L), dev(R), win, D, ctx=ctx).cpu().numpy()
        assert kname(ctx) == "sad_generic_kernel"
        assert np.array_equal(got, exp), (win, H, W, D, np.argwhere(got != exp)[:4])
    # a window as large as the frame: nothing to compute (SAD.cpp:34,36), whatever its size
    L, R = noise_pair(1, 40, 50)
    assert not oa.get_disparity_host(L, R, 5000, 30).any()
