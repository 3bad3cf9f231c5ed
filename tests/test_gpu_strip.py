"""GPU parity of the strip matcher (forced with KERNEL_STRIP) against the oracle, and of the default matcher choice
against both forced kernels.  Bit-exact bar; calls go through the C ABI.  The family has two members: sad_strip_kernel
(operands pre-packed in LDS; D > 64) and sad_strip_otf_kernel (operands packed on the fly; D <= 64); tests that take
the `form` fixture run once with the default choice and once with SADBM_OTF=0, i.e. sad_strip_kernel at every D."""
import numpy as np
import pytest
import opencv_amd as oa
import oracle
from helpers import kname, noise_pair, strip_kernel_name

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "staged"])
def form(request, monkeypatch):
    if request.param == "staged":
        monkeypatch.setenv("SADBM_OTF", "0")
    else:
        monkeypatch.delenv("SADBM_OTF", raising=False)
    oa.reload_env()
    return request.param == "staged"


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("win,D", [(15, 128), (7, 30), (9, 64), (7, 128), (21, 256), (4, 64), (5, 17), (6, 70), (8, 100), (12, 64), (13, 200),
                                   (14, 64), (16, 129), (17, 64), (20, 64), (25, 256), (10, 1), (11, 16)])
def test_strip_small(win, D, form):
    for (H, W) in ((61, 203), (70, 333), (40, 600)):
        for binary in (False, True):
            L, R = noise_pair(win * 977 + D + W, H, W, shift=5, binary=binary)
            exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
            ctx = oa.Context(0)
            got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
            assert kname(ctx) == strip_kernel_name(win, D, form, cols=W)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (win, D, H, W, binary, len(bad), bad[:8], got[tuple(bad[0])], exp[tuple(bad[0])])


def test_default_choice_matches_both_kernels():
    """Whatever run_pipeline picks by default equals the forced quad and strip kernels (and the oracle)."""
    L, R = oracle.rds(11, 900, 300, 6, 50, 0)
    for win, D in ((9, 128), (15, 128), (21, 96), (7, 65), (13, 256)):
        exp = oracle.box(L, R, win, D)
        for fl in (0, oa.KERNEL_QUAD, oa.KERNEL_STRIP):
            assert np.array_equal(oa.compute_device(dev(L), dev(R), win, D, fl).cpu().numpy(), exp), (win, D, fl)


def test_strip_batch_bands_and_multichannel():
    """Strip matcher through the batch entry point (pair stride), a row band, BGR input and an unaligned output view."""
    import torch
    n, H, W, win, D = 3, 90, 517, 15, 128
    Ls, Rs = zip(*[oracle.rds(50 + i, W, H, 4 + i, 70 + i, 0) for i in range(n)])
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, oa.KERNEL_STRIP).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), i
    # band of a taller frame: rows 30..60 of pair 0 (needs input rows 30..74)
    got = oa.compute_band_device(dev(Ls[0][30:74]), dev(Rs[0][30:74]), H, 30, 30, win, D, oa.KERNEL_STRIP).cpu().numpy()
    assert np.array_equal(got, oracle.box(Ls[0], Rs[0], win, D)[30:60])
    # 3-channel input (plane 0) and an output view that is not 4-byte aligned
    L3 = np.stack([Ls[1], Ls[2], Ls[0]], -1); R3 = np.stack([Rs[1], Rs[2], Rs[0]], -1)
    obuf = torch.zeros((H, W + 7), dtype=torch.uint8, device="cuda")
    oa.compute_device(dev(L3), dev(R3), win, D, oa.KERNEL_STRIP, out=obuf[:, 3:3 + W])
    assert np.array_equal(obuf[:, 3:3 + W].cpu().numpy(), oracle.box(Ls[1], Rs[1], win, D))
    assert not obuf[:, :3].any() and not obuf[:, 3 + W:].any()


def test_strip_full_size_4k_band():
    """BASELINE config 3 at full size through the strip matcher: zero frame, planted disparities, a 150-row band vs the oracle."""
    W, H, win, D = 3840, 2160, 15, 128
    L, R = oracle.rds(3, W, H, 16, 96, 0)
    ctx = oa.Context(0)
    got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
    assert kname(ctx) == "sad_strip_kernel<15>"
    assert not got[H - win:, :].any() and not got[:, W - win:].any()
    assert (got[10, :16] == np.arange(1, 17)).all()
    assert (got[100, 200:900] == 16).all() and (got[1000, 1200:2700] == 96).all()
    r0, n = 1000, 150
    exp = oracle.box(L[r0: r0 + n + win], R[r0: r0 + n + win], win, D, oracle.RAW_LOC)[:n]
    assert np.array_equal(got[r0: r0 + n], exp)


def test_strip_short_searches_batches_bands_and_borders(form):
    """Searches of at most 64 disparities (the reference's own DSR = 30; 64-, 128- and 256-thread blocks of the on-the-fly form):
    batches with frame groups, a row band, BGR, D not a multiple of 16, binary dots (ties), the reference's own pair."""
    import torch
    from helpers import tsukuba_bgr
    for (H, W, n, win, D) in ((61, 517, 5, 7, 30), (40, 1242, 3, 9, 32), (90, 384, 8, 7, 17), (33, 260, 4, 15, 31), (50, 300, 2, 21, 24), (60, 384, 6, 7, 12),
                             (45, 700, 3, 17, 16), (64, 517, 4, 9, 64), (40, 640, 3, 15, 48), (30, 530, 5, 25, 5)):
        Ls, Rs = zip(*[oracle.rds(70 + i, W, H, 2 + i % 3, min(D - 1, 12 + i), i & 1) for i in range(n)])
        ctx = oa.Context(0)
        out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
        assert kname(ctx) == strip_kernel_name(win, D, form, cols=W)
        for i in range(n):
            assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), (H, W, win, D, i)
        r0, nb = H // 3, H // 3
        i1 = min(r0 + nb + win - 1, H)
        got = oa.compute_band_device(dev(Ls[0][r0:i1]), dev(Rs[0][r0:i1]), H, r0, nb, win, D, oa.KERNEL_STRIP).cpu().numpy()
        assert np.array_equal(got, oracle.box(Ls[0], Rs[0], win, D)[r0:r0 + nb]), (H, W, win, D, "band")
    L, R = tsukuba_bgr()
    exp = oracle.literal(np.ascontiguousarray(L[:, :, 0]), np.ascontiguousarray(R[:, :, 0]), 7, 30)
    assert np.array_equal(oa.compute_device(dev(L), dev(R), 7, 30, oa.KERNEL_STRIP).cpu().numpy(), exp)
    Lf, Rf = oracle.rds(2, 1920, 1080, 4, 20, 0)                 # a full 1080p frame at the reference's parameters
    assert np.array_equal(oa.compute_device(dev(Lf), dev(Rf), 7, 30, oa.KERNEL_STRIP).cpu().numpy(), oracle.box(Lf, Rf, 7, 30))


@pytest.mark.parametrize("shape", [(9, 300), (300, 9), (40, 17), (8, 8), (16, 50), (50, 16), (1, 1), (33, 129), (20, 255), (20, 257), (20, 513)])
def test_strip_small_and_degenerate_shapes(shape, form):
    """Shapes around the strip (16), block (256) and window sizes, W < D, W <= win, H <= win -- forced strip matcher."""
    H, W = shape
    L, R = noise_pair(H * 31 + W, H, W)
    for win, D in ((7, 30), (4, 64), (9, 256), (15, 100), (21, 65), (13, 32), (5, 3)):
        exp = oracle.literal(L, R, win, D)
        got = oa.get_disparity_host(L, R, win, D, oa.KERNEL_STRIP)
        assert np.array_equal(got, exp), (shape, win, D)


def test_strip_constant_identical_and_binary_ties():
    c1 = np.full((40, 300), 10, np.uint8)
    c2 = np.full((40, 300), 200, np.uint8)
    assert np.array_equal(oa.get_disparity_host(c1, c2, 15, 128, oa.RAW_LOC | oa.KERNEL_STRIP), oracle.literal(c1, c2, 15, 128, oracle.RAW_LOC))   # K2
    L, R = noise_pair(3, 40, 300)
    assert not oa.get_disparity_host(L, L, 9, 128, oa.RAW_LOC | oa.KERNEL_STRIP).any()                                                          # K1
    z = np.zeros((40, 300), np.uint8)                          # pixel value 0 everywhere: every cost is 0, first minimum = 0
    assert not oa.get_disparity_host(z, z, 13, 200, oa.RAW_LOC | oa.KERNEL_STRIP).any()
    Lb, Rb = oracle.rds(1, 320, 240, 4, 20, 1)                 # binary dots: heavy ties (K9)
    assert np.array_equal(oa.get_disparity_host(Lb, Rb, 9, 128, oa.KERNEL_STRIP), oracle.box(Lb, Rb, 9, 128))
    big = np.full((30, 400), 255, np.uint8)                    # largest costs: 16-bit halves must not overflow (16x16x255 = 65280)
    zero = np.zeros((30, 400), np.uint8)
    assert np.array_equal(oa.get_disparity_host(big, zero, 16, 128, oa.RAW_LOC | oa.KERNEL_STRIP), oracle.literal(big, zero, 16, 128, oracle.RAW_LOC))


@pytest.mark.parametrize("win", list(range(4, 26)))
def test_strip_every_window_instantiation(win, form):
    """Every window the strip matchers are instantiated for (their geometry is constexpr per WIN), at every block size:
    16 / 32 / 64 disparities (on-the-fly form, or 64-disparity blocks of the staged form) and 128 / 256."""
    W = 283
    L, R = noise_pair(900 + win, 57, W, shift=7)
    for D in ((9, 16, 30, 32, 50, 64) if not form else (9, 30, 50)) + (97, 256):
        ctx = oa.Context(0)
        got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
        assert kname(ctx) == strip_kernel_name(win, D, form, cols=W)
        assert np.array_equal(got, oracle.box(L, R, win, D, oracle.RAW_LOC)), (win, D)


@pytest.mark.parametrize("win", list(range(4, 26)))
def test_on_the_fly_form_512_thread_blocks_every_window(win, monkeypatch):
    """SADBM_OTF=1: the on-the-fly form wherever it is instantiated, i.e. up to 128 disparities (512-thread blocks; by default only
    for narrow windows at 65-128 disparities that are not a multiple of 16), plane-reading and direct variants."""
    monkeypatch.setenv("SADBM_OTF", "1")
    L, R = noise_pair(1700 + win, 49, 531, shift=11)
    for D in (65, 100, 128):
        exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
        for direct in ("0", "1"):
            monkeypatch.setenv("SADBM_DIRECT", direct)
            oa.reload_env()
            ctx = oa.Context(0)
            got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC, ctx=ctx).cpu().numpy()
            assert ctx.last_kernel() == "sad_strip_otf_kernel<%d,8>%s" % (win, "+direct" if direct == "1" else "")
            assert np.array_equal(got, exp), (win, D, direct)


@pytest.mark.parametrize("shape,pairs", [((40, 203), 5), ((33, 640), 3), ((25, 100), 7), ((20, 16), 9), ((30, 250), 2), ((22, 1242), 4),
                                         ((12, 37), 33)])
def test_strip_frame_groups(shape, pairs, monkeypatch, form):
    """Batches whose frame width is not a multiple of 256: the strip matcher lays several frames side by side in one
    plane (run_pipeline / strip_frame_groups) -- every frame still equals the oracle, for widths that are not multiples
    of 16, short last groups, BGR input, an unaligned output, a border-heavy D and one group per frame forced."""
    import torch
    H, W = shape
    Ls, Rs = zip(*[noise_pair(31 * W + i, H, W, shift=3 + i % 5) for i in range(pairs)])
    for win, D in ((7, 128), (9, 64), (15, 130), (21, 256), (7, 30), (13, 16)) if not form else ((9, 64), (7, 30)):
        exp = [oracle.box(Ls[i], Rs[i], win, D) for i in range(pairs)]
        for fg in (None, "1", "3"):
            if fg is None:
                monkeypatch.delenv("SADBM_STRIP_FG", raising=False)
            else:
                monkeypatch.setenv("SADBM_STRIP_FG", fg)
            oa.reload_env()
            ctx = oa.Context(0)
            out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, oa.KERNEL_STRIP, ctx=ctx).cpu().numpy()
            assert kname(ctx) == (strip_kernel_name(win, D, form, cols=W) if H > win and W > win else "none")
            for i in range(pairs):
                bad = np.argwhere(out[i] != exp[i])
                assert bad.size == 0, (shape, pairs, win, D, fg, i, len(bad), bad[:6])
    monkeypatch.delenv("SADBM_STRIP_FG", raising=False)
    oa.reload_env()
    # BGR frames (plane 0) into an output batch whose rows are not 4-byte aligned
    win, D = 9, 128
    L3 = np.stack([np.stack([Ls[i], Rs[i], Ls[i]], -1) for i in range(pairs)])
    R3 = np.stack([np.stack([Rs[i], Ls[i], Rs[i]], -1) for i in range(pairs)])
    obuf = torch.full((pairs, H, W + 5), 7, dtype=torch.uint8, device="cuda")
    oa.compute_batch_device(dev(L3), dev(R3), win, D, oa.KERNEL_STRIP, out=obuf[:, :, 1:1 + W])
    got = obuf.cpu().numpy()
    for i in range(pairs):
        assert np.array_equal(got[i, :, 1:1 + W], oracle.box(Ls[i], Rs[i], win, D)), i
    assert (got[:, :, 0] == 7).all() and (got[:, :, 1 + W:] == 7).all()


@pytest.mark.parametrize("win,D", [(7, 30), (9, 64), (15, 16), (21, 48), (4, 9), (25, 32)])
def test_direct_reads_of_caller_images(win, D, monkeypatch):
    """The on-the-fly form reads single-channel inputs itself (no prep pass, "+direct"): views at the very start and end of
    their allocation (the byte-wise edge path), row-strided ROI views with different pitches for left and right, a larger
    right image, pitch == cols (units run into the next rows), frames narrower than a 16-byte unit, batches, a row band --
    equal to the oracle and to the prep-pass path (SADBM_DIRECT=0)."""
    import torch
    if D <= 16 and win >= 13:                                  # 64-thread blocks with wide windows keep the prep pass by default (spills)
        monkeypatch.setenv("SADBM_DIRECT", "1")
        oa.reload_env()
    rng = np.random.default_rng(win * 100 + D)
    for (H, W) in ((win + 3, 300), (40, 517), (win + 2, 9 + win), (win + 8, 256), (win + 1, 1000)):
        L, R = noise_pair(H * 7 + W + D, H, W, shift=min(5, W - 1))
        exp = oracle.box(L, R, win, D)
        ctx = oa.Context(0)
        # exact-size allocations: nothing readable before the first or after the last byte
        got = oa.compute_device(dev(L), dev(R), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == strip_kernel_name(win, D, cols=W) + "+direct", ctx.last_kernel()
        assert np.array_equal(got, exp), (H, W, "exact")
        # ROI views: left inside a wider buffer at a column offset, right inside a taller and wider one (its own pitch)
        bl = torch.from_numpy(rng.integers(0, 256, size=(H, W + 37), dtype=np.uint8)).cuda()
        br = torch.from_numpy(rng.integers(0, 256, size=(H + 3, W + 80), dtype=np.uint8)).cuda()
        bl[:, 5:5 + W] = dev(L); br[2:2 + H, 41:41 + W] = dev(R)
        got = oa.compute_device(bl[:, 5:5 + W], br[2:2 + H, 41:41 + W], win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel().endswith("+direct")
        assert np.array_equal(got, exp), (H, W, "roi")
        # a larger right image (SAD.cpp:47 takes its ROI at the left image's coordinates)
        Rb = rng.integers(0, 256, size=(H + 2, W + 9), dtype=np.uint8); Rb[:H, :W] = R
        assert np.array_equal(oa.compute_device(dev(L), dev(Rb), win, D, 0, ctx=ctx).cpu().numpy(), exp), (H, W, "larger right")
        monkeypatch.setenv("SADBM_DIRECT", "0")
        oa.reload_env()
        got0 = oa.compute_device(dev(L), dev(R), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == strip_kernel_name(win, D, cols=W)
        assert np.array_equal(got0, exp)
        if D <= 16 and win >= 13:
            monkeypatch.setenv("SADBM_DIRECT", "1")
        else:
            monkeypatch.delenv("SADBM_DIRECT")
        oa.reload_env()
    # batches (frame width a multiple of 256: no frame groups to lose) and a row band of a taller frame
    n, H, W = 5, 45, 512
    Ls, Rs = zip(*[oracle.rds(30 + i, W, H, 1 + i % 3, min(D - 1, 9 + i), i & 1) for i in range(n)])
    ctx = oa.Context(0)
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().endswith("+direct")
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), i
    if H > win + 12:
        r0, nb = 7, 12
        i1 = min(r0 + nb + win - 1, H)
        got = oa.compute_band_device(dev(Ls[0][r0:i1]), dev(Rs[0][r0:i1]), H, r0, nb, win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel().endswith("+direct")
        assert np.array_equal(got, oracle.box(Ls[0], Rs[0], win, D)[r0:r0 + nb])


@pytest.mark.parametrize("W", [384, 203, 16, 640])
def test_direct_reads_with_frame_groups(W, monkeypatch):
    """Direct reads where frames share a group: frames lie a multiple of 16 columns apart in the group's column space; a staging
    unit that starts in the previous frame (the c bytes before column 0 of R's copy c) is read from the row above instead --
    no competing slot needs either.  Widths that are and are not multiples of 16 / 256, cropped batch views (row pitch and
    frame stride passed through), SADBM_DIRECT=1 (also 64-thread blocks with wide windows)."""
    import torch
    n, H = 9, 38
    for win, D in ((7, 30), (9, 64)):
        Ls, Rs = zip(*[noise_pair(W + 13 * i + win, H, W, shift=min(2 + i, W - 1)) for i in range(n)])
        bl = torch.full((n, H + 2, W + 21), 99, dtype=torch.uint8, device="cuda"); br = torch.full((n, H + 2, W + 21), 77, dtype=torch.uint8, device="cuda")
        bl[:, 1:1 + H, 5:5 + W] = dev(np.stack(Ls)); br[:, 1:1 + H, 5:5 + W] = dev(np.stack(Rs))
        ctx = oa.Context(0)
        out = oa.compute_batch_device(bl[:, 1:1 + H, 5:5 + W], br[:, 1:1 + H, 5:5 + W], win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == strip_kernel_name(win, D, cols=W) + "+direct"
        for i in range(n):
            assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), ("view", W, win, D, i)
    monkeypatch.setenv("SADBM_DIRECT", "1")
    oa.reload_env()
    for win, D in ((7, 30), (9, 64), (13, 12)):
        Ls, Rs = zip(*[noise_pair(W + 11 * i + win, H, W, shift=min(3 + i, W - 1), binary=bool(i & 1)) for i in range(n)])
        ctx = oa.Context(0)
        out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == strip_kernel_name(win, D, cols=W) + "+direct"
        for i in range(n):
            assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), (W, win, D, i)


def test_direct_is_not_used_for_host_mapped_inputs_but_for_colour():
    from helpers import tsukuba_bgr
    L, R = tsukuba_bgr()
    ctx = oa.Context(0)
    exp = oracle.literal(np.ascontiguousarray(L[:, :, 0]), np.ascontiguousarray(R[:, :, 0]), 7, 30)
    assert np.array_equal(oa.compute_device(dev(L), dev(R), 7, 30, ctx=ctx).cpu().numpy(), exp)
    assert ctx.last_kernel() == "sad_strip_otf_kernel<7,2>+direct"        # round 3: the reference's own call (a BGR pair, SAD.cpp:65-66,76) is one launch
    assert np.array_equal(oa.compute_device(dev(L), dev(R), 15, 30, ctx=ctx).cpu().numpy(),
                          oracle.box(np.ascontiguousarray(L[:, :, 0]), np.ascontiguousarray(R[:, :, 0]), 15, 30))
    assert ctx.last_kernel() == "sad_strip_otf_kernel<15,2>"              # windows beyond 8: channel 0 goes through the prep pass
    assert np.array_equal(oa.get_disparity_host(L, R, 7, 30, ctx=ctx), exp)
    assert ctx.last_kernel() == "sad_strip_otf_kernel<7,2>"               # small host frames are read over PCIe: prep pass, once
    big = np.ascontiguousarray(np.tile(L[:, :, 0], (3, 3))); bigr = np.ascontiguousarray(np.tile(R[:, :, 0], (3, 3)))
    assert np.array_equal(oa.get_disparity_host(big, bigr, 7, 30, ctx=ctx), oracle.box(big, bigr, 7, 30))
    assert ctx.last_kernel() == "sad_strip_otf_kernel<7,2>+direct"        # staged on the device first: read directly


@pytest.mark.parametrize("tail", ["0", "1", "5", "100000"])
def test_half_height_blocks_at_the_end_of_the_launch(tail, monkeypatch):
    """SADBM_TAIL: the last units of every XCD's share of an on-the-fly launch run as two half-height blocks (by default only on
    jobs of three rounds of block slots and more) -- forced here on small batches: odd band heights, one-row bands, a band
    shorter than the others, frame groups, the zero frame below the computed rows."""
    monkeypatch.setenv("SADBM_TAIL", tail)
    for ty in ("0", "1", "7"):
        monkeypatch.setenv("SADBM_STRIP_TY", ty)
        oa.reload_env()
        for (H, W, n, win, D) in ((47, 530, 5, 7, 30), (30, 384, 9, 9, 64), (26, 270, 3, 15, 12)):
            Ls, Rs = zip(*[noise_pair(H + 3 * i + win, H, W, shift=min(2 + i, D - 1)) for i in range(n)])
            out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, 0).cpu().numpy()
            for i in range(n):
                assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), (tail, ty, H, W, win, D, i)


@pytest.mark.parametrize("win", [5, 6, 7, 9, 11, 8, 13, 16, 17, 18, 19, 20, 21, 22, 25])
def test_largest_costs_with_poisoned_sums(win):
    """All-255 against all-0 (every absolute difference 255: the largest cost a window can have) and the reverse, checkerboards of
    the two, at D = 30 / 64 / 100: the poisoned sums of never-competing slots (k >= D, k > column) must stay above every real
    cost without overflowing their 16-bit fields -- at the left border, where both kinds of slot meet."""
    H, W = win + 6, 300
    big = np.full((H, W), 255, np.uint8)
    zero = np.zeros((H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    chk = np.where((xx // 3 + yy) & 1, 255, 0).astype(np.uint8)
    for D in (30, 64, 100, 200, 256):
        for L, R in ((big, zero), (zero, big), (chk, 255 - chk), (big, chk)):
            exp = oracle.literal(L, R, win, D, oracle.RAW_LOC)
            got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC).cpu().numpy()
            assert np.array_equal(got, exp), (win, D, np.argwhere(got != exp)[:4])


@pytest.mark.parametrize("win", [17, 18, 19, 20, 21])
def test_packed_word_sums_of_windows_17_to_21(win, form):
    """Windows of 17-21 columns keep two disparities' whole-word sums in one register as 16-bit halves (SG::PKV): a word sum
    reaches 4*255*win and a 3-word partial sum 64260 of 65535 -- all-255 against all-0 drives every half to its maximum, in
    both kernel forms, with k >= D slots (poison through the key's k term) and the left border in the same frame."""
    H, W = win + 7, 560
    big = np.full((H, W), 255, np.uint8)
    zero = np.zeros((H, W), np.uint8)
    g = np.random.default_rng(win)
    rnd = g.integers(0, 256, (H, W), dtype=np.uint8)
    for D in (20, 50, 64, 90, 128, 250):
        for L, R in ((big, zero), (zero, big), (rnd, np.roll(rnd, -33, axis=1)), (rnd, 255 - rnd)):
            exp = oracle.box(L, R, win, D, oracle.RAW_LOC)
            got = oa.compute_device(dev(L), dev(R), win, D, oa.RAW_LOC | oa.KERNEL_STRIP).cpu().numpy()
            assert np.array_equal(got, exp), (win, D, np.argwhere(got != exp)[:4])


@pytest.mark.parametrize("win,D", [(7, 128), (15, 128), (21, 256), (9, 80), (13, 200), (17, 96), (25, 130), (5, 256)])
def test_staged_kernel_reads_caller_images(win, D, monkeypatch):
    """Round 3: sad_strip_kernel reads single-channel device images in place too ("+direct": no prep pass, no scratch planes).  Its edge rows
    -- the first row of a batch (look-back left of column 0) and the last one (16-byte windows past the end) -- are assembled byte-wise
    into spare row images before the march.  Exact-size allocations (nothing readable before the first or after the last byte), ROI views
    inside larger buffers, pitch == cols, frame groups of a batch, several bands, a row band, binary dots; equal to the oracle and to the
    prep-pass path (SADBM_DIRECT=0).  Left and right with different pitches, and rows shorter than 384 bytes, keep the planes."""
    import torch
    rng = np.random.default_rng(win * 1000 + D)
    for (H, W, binary) in ((win + 3, 384, False), (win + 40, 517, True), (win + 2, 1000, False), (130, 400, False), (win + 70, 2048, True)):
        L, R = noise_pair(H * 7 + W + D, H, W, shift=min(5, W - 1), binary=binary)
        exp = oracle.box(L, R, win, D)
        ctx = oa.Context(0)
        got = oa.compute_device(dev(L), dev(R), win, D, 0, ctx=ctx).cpu().numpy()                 # exact-size allocations
        assert ctx.last_kernel() == "sad_strip_kernel<%d>+direct" % win, ctx.last_kernel()
        assert np.array_equal(got, exp), (H, W, "exact", np.argwhere(got != exp)[:4])
        # ROI views with a common pitch: left at a column offset of one buffer, right at another offset of an equally wide one
        bl = torch.from_numpy(rng.integers(0, 256, size=(H + 1, W + 80), dtype=np.uint8)).cuda()
        br = torch.from_numpy(rng.integers(0, 256, size=(H + 3, W + 80), dtype=np.uint8)).cuda()
        bl[1:1 + H, 5:5 + W] = dev(L); br[2:2 + H, 41:41 + W] = dev(R)
        got = oa.compute_device(bl[1:1 + H, 5:5 + W], br[2:2 + H, 41:41 + W], win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel().endswith("+direct")
        assert np.array_equal(got, exp), (H, W, "roi")
        # the same image on both sides (one batch is the other): loc 0 everywhere (K1)
        same = oa.compute_device(dev(L), dev(L), win, D, oa.RAW_LOC, ctx=ctx)
        assert ctx.last_kernel().endswith("+direct") and not same.any()
        # different pitches: the planes (the direct form addresses both images from one base with one pitch)
        Rb = rng.integers(0, 256, size=(H + 2, W + 9), dtype=np.uint8); Rb[:H, :W] = R
        got = oa.compute_device(dev(L), dev(Rb), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == "sad_strip_kernel<%d>" % win
        assert np.array_equal(got, exp), (H, W, "larger right")
        monkeypatch.setenv("SADBM_DIRECT", "0")
        oa.reload_env()
        got0 = oa.compute_device(dev(L), dev(R), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == "sad_strip_kernel<%d>" % win
        assert np.array_equal(got0, exp)
        monkeypatch.delenv("SADBM_DIRECT")
        oa.reload_env()
    # batches: frame groups (400-column frames, Wp = 400: 2-3 frames per 256-column block row), cropped batch views, a band
    n, H, W = 7, win + 21, 400
    Ls, Rs = zip(*[noise_pair(W + 13 * i + win, H, W, shift=min(2 + i, D - 1), binary=bool(i & 1)) for i in range(n)])
    ctx = oa.Context(0)
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().endswith("+direct")
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), ("batch", i)
    bl = torch.full((n, H + 2, W + 21), 99, dtype=torch.uint8, device="cuda"); br = torch.full((n, H + 2, W + 21), 77, dtype=torch.uint8, device="cuda")
    bl[:, 1:1 + H, 5:5 + W] = dev(np.stack(Ls)); br[:, 1:1 + H, 5:5 + W] = dev(np.stack(Rs))
    out = oa.compute_batch_device(bl[:, 1:1 + H, 5:5 + W], br[:, 1:1 + H, 5:5 + W], win, D, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().endswith("+direct")
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], win, D)), ("view", i)
    r0, nb = 3, 11
    i1 = min(r0 + nb + win - 1, H)
    got = oa.compute_band_device(dev(Ls[0][r0:i1]), dev(Rs[0][r0:i1]), H, r0, nb, win, D, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().endswith("+direct")
    assert np.array_equal(got, oracle.box(Ls[0], Rs[0], win, D)[r0:r0 + nb])


@pytest.mark.parametrize("ch", [3])
@pytest.mark.parametrize("win,D", [(7, 30), (7, 17), (8, 32), (5, 17), (6, 24)])
def test_colour_frames_read_in_place(win, D, ch, monkeypatch):
    """Round 3: interleaved BGR device frames (cv::imread's layout, SAD.cpp:65-66; the cost is channel 0's, SAD.cpp:19-20) are staged by
    the on-the-fly kernel itself -- 48 bytes per unit, channel 0 gathered with v_perm_b32 -- at 17..32 disparities and windows up to 8
    (the reference's winSize 7 / DSR 30): exact-size allocations (byte-wise edge path), planes that differ in every channel, ROI views
    with different pitches, frame groups of a batch, a band, binary dots; equal to the oracle on plane 0 and to the prep-pass path.
    Frames narrower than 17 columns search fewer than 17 disparities (64-thread blocks): prep pass."""
    import torch
    rng = np.random.default_rng(win * 100 + D + ch)
    p0 = lambda a: np.ascontiguousarray(a[..., 0])
    for (H, W, binary) in ((win + 3, 300, False), (40, 517, True), (win + 2, 9 + win, False), (win + 8, 256, False), (win + 1, 1000, True), (288, 384, False)):
        L, R = noise_pair(H * 7 + W + D, H, W, shift=min(5, W - 1), binary=binary, ch=ch)
        exp = oracle.box(p0(L), p0(R), win, D)
        ctx = oa.Context(0)
        got = oa.compute_device(dev(L), dev(R), win, D, 0, ctx=ctx).cpu().numpy()                 # exact-size allocations
        in_place = min(D, W) > 16
        assert ctx.last_kernel() == strip_kernel_name(win, D, cols=W) + ("+direct" if in_place else ""), ctx.last_kernel()
        assert np.array_equal(got, exp), (H, W, "exact", np.argwhere(got != exp)[:4])
        bl = torch.from_numpy(rng.integers(0, 256, size=(H, W + 37, ch), dtype=np.uint8)).cuda()
        br = torch.from_numpy(rng.integers(0, 256, size=(H + 3, W + 80, ch), dtype=np.uint8)).cuda()
        bl[:, 5:5 + W] = dev(L); br[2:2 + H, 41:41 + W] = dev(R)
        got = oa.compute_device(bl[:, 5:5 + W], br[2:2 + H, 41:41 + W], win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel().endswith("+direct") == in_place
        assert np.array_equal(got, exp), (H, W, "roi")
        monkeypatch.setenv("SADBM_DIRECT", "0")
        oa.reload_env()
        got0 = oa.compute_device(dev(L), dev(R), win, D, 0, ctx=ctx).cpu().numpy()
        assert ctx.last_kernel() == strip_kernel_name(win, D, cols=W)
        assert np.array_equal(got0, exp)
        monkeypatch.delenv("SADBM_DIRECT")
        oa.reload_env()
    n, H, W = 7, 45, 203
    Ls, Rs = zip(*[noise_pair(30 + i + win, H, W, shift=min(D - 1, 3 + i), binary=bool(i & 1), ch=ch) for i in range(n)])
    ctx = oa.Context(0)
    out = oa.compute_batch_device(dev(np.stack(Ls)), dev(np.stack(Rs)), win, D, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().endswith("+direct")
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(p0(Ls[i]), p0(Rs[i]), win, D)), i
    r0, nb = 7, 12
    i1 = min(r0 + nb + win - 1, H)
    got = oa.compute_band_device(dev(Ls[0][r0:i1]), dev(Rs[0][r0:i1]), H, r0, nb, win, D, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().endswith("+direct")
    assert np.array_equal(got, oracle.box(p0(Ls[0]), p0(Rs[0]), win, D)[r0:r0 + nb])


def test_profile_events_report_no_prep_time_when_no_prep_pass_runs():
    """sadbm_profile_begin / _end (HIP events round the kernels): launches that read the frames in place report 0 for the prep pass (two events
    recorded back to back would read as ~5 us), launches through the prep pass a positive time; more launches than asked for are not counted."""
    import torch
    ctx = oa.Context(0)
    L, R = noise_pair(3, 300, 640, shift=7)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    L4, R4 = noise_pair(4, 300, 640, shift=7, ch=4)
    dL4, dR4 = torch.from_numpy(L4).cuda(), torch.from_numpy(R4).cuda()
    ctx.profile_begin(4)
    for _ in range(6):
        oa.compute_device(dL, dR, 7, 30, ctx=ctx)
    torch.cuda.synchronize()
    n, m, p = ctx.profile_end()
    assert ctx.last_kernel().endswith("+direct") and n == 4 and m > 0 and p == 0
    ctx.profile_begin(4)
    for _ in range(3):
        oa.compute_device(dL4, dR4, 7, 30, ctx=ctx)
    oa.compute_device(dL, dR, 7, 30, ctx=ctx)
    torch.cuda.synchronize()
    n, m, p = ctx.profile_end()
    assert n == 4 and m > 0 and p > 0
    ctx.profile_begin(2)                                       # nothing launched: zeros
    assert ctx.profile_end() == (0, 0.0, 0.0)
