"""Seeded random configurations of the batch entry point against the oracle: shapes around the strip (16), block (256)
and plane-group boundaries, every matcher, raw / encoded output, 1/3/4 channels, pitched outputs.  Bit-exact bar."""
import numpy as np
import pytest
import opencv_amd as oa
import oracle
from helpers import noise_pair

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        win = int(rng.integers(1, 27))
        D = int(rng.choice([1, 2, 15, 16, 17, 30, 31, 32, 33, 47, 48, 64, 65, 96, 100, 127, 128, 129, 200, 255, 256]))
        H = int(rng.integers(1, 70))
        W = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 300)), int(rng.choice([255, 256, 257, 271, 272, 273, 511, 512, 513])),
                            int(rng.integers(300, 700))]))
        pairs = int(rng.choice([1, 1, 2, 3, 5, 8, 17]))
        ch = int(rng.choice([1, 1, 3, 4]))
        flags = int(rng.choice([0, oa.RAW_LOC])) | int(rng.choice([0, 0, oa.KERNEL_STRIP, oa.KERNEL_QUAD]))
        pad = int(rng.choice([0, 0, 3, 16]))
        out.append((win, D, H, W, pairs, ch, flags, pad))
    return out


@pytest.mark.parametrize("chunk", range(10))
def test_random_batches_match_oracle(chunk):
    import torch
    for case, (win, D, H, W, pairs, ch, flags, pad) in enumerate(_cases(20, 1000 + chunk)):
        Ls, Rs = zip(*[noise_pair(7 * case + i + 100 * chunk, H, W, shift=(1 + (i + case) % 9 if W > 12 else None), binary=(case % 4 == 3), ch=ch) for i in range(pairs)])
        L, R = np.stack(Ls), np.stack(Rs)
        obuf = torch.full((pairs, H, W + pad), 0xAB, dtype=torch.uint8, device="cuda")
        out = obuf[:, :, pad // 2: pad // 2 + W] if pad else obuf
        oa.compute_batch_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), win, D, flags, out=out)
        got = obuf.cpu().numpy()
        oflags = oracle.RAW_LOC if (flags & oa.RAW_LOC) else 0
        for i in range(pairs):
            exp = oracle.box(Ls[i], Rs[i], win, D, oflags)
            g = got[i, :, pad // 2: pad // 2 + W]
            bad = np.argwhere(g != exp)
            assert bad.size == 0, ((win, D, H, W, pairs, ch, flags, pad), i, len(bad), bad[:5])
        if pad:
            assert (got[:, :, :pad // 2] == 0xAB).all() and (got[:, :, pad // 2 + W:] == 0xAB).all()


@pytest.mark.parametrize("chunk", range(2))
def test_random_wide_windows_match_oracle(chunk):
    """The same bar for windows of 50..255 columns (sad_wide_kernel): frames narrower, as wide as and a few column blocks wider than
    the window, 1/3/4 channels, pitched outputs whose padding must stay untouched."""
    import torch
    rng = np.random.default_rng(77 + chunk)
    for case in range(14):
        win = int(rng.choice([50, 51, 63, 64, 65, 100, 128, 177, 254, 255, int(rng.integers(50, 256))]))
        D = int(rng.choice([1, 7, 8, 9, 30, 63, 64, 65, 100, 128, 129, 200, 256]))
        H = int(rng.choice([win - 3, win, win + 1, win + int(rng.integers(2, 70))]))
        W = int(rng.choice([win, win + 1, win + int(rng.integers(2, 40)), win + int(rng.integers(40, 700))]))
        pairs, ch, pad = int(rng.choice([1, 2, 5])), int(rng.choice([1, 1, 3, 4])), int(rng.choice([0, 3, 16]))
        flags = int(rng.choice([0, oa.RAW_LOC]))
        Ls, Rs = zip(*[noise_pair(31 * case + i + 500 * chunk, H, W, shift=1 + (i + case) % 9, binary=(case % 3 == 2), ch=ch) for i in range(pairs)])
        obuf = torch.full((pairs, H, W + pad), 0xAB, dtype=torch.uint8, device="cuda")
        out = obuf[:, :, pad // 2: pad // 2 + W] if pad else obuf
        oa.compute_batch_device(torch.from_numpy(np.stack(Ls)).cuda(), torch.from_numpy(np.stack(Rs)).cuda(), win, D, flags, out=out)
        got = obuf.cpu().numpy()
        for i in range(pairs):
            Li, Ri = (Ls[i], Rs[i]) if ch == 1 else (np.ascontiguousarray(Ls[i][:, :, 0]), np.ascontiguousarray(Rs[i][:, :, 0]))
            exp = oracle.box(Li, Ri, win, D, oracle.RAW_LOC if (flags & oa.RAW_LOC) else 0)
            bad = np.argwhere(got[i, :, pad // 2: pad // 2 + W] != exp)
            assert bad.size == 0, ((win, D, H, W, pairs, ch, flags, pad), i, len(bad), bad[:5])
        if pad:
            assert (got[:, :, :pad // 2] == 0xAB).all() and (got[:, :, pad // 2 + W:] == 0xAB).all()


def test_very_wide_frame_and_many_tiny_frames():
    """Frames wider than the strip matcher's 16-bit column fields fall back to the quad matcher; hundreds of tiny frames
    go into wide plane groups."""
    import torch
    L, R = noise_pair(5, 12, 66000, shift=7)
    ctx = oa.Context(0)
    got = oa.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), 5, 40, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel().startswith("sad_quad")
    assert np.array_equal(got, oracle.box(L, R, 5, 40))
    n = 700
    Ls, Rs = zip(*[noise_pair(900 + i, 11, 23, shift=2) for i in range(n)])
    out = oa.compute_batch_device(torch.from_numpy(np.stack(Ls)).cuda(), torch.from_numpy(np.stack(Rs)).cuda(), 4, 48, 0, ctx=ctx).cpu().numpy()
    assert ctx.last_kernel() == "sad_strip_kernel<4>"            # 4-wide windows keep the staged form from 17 disparities up
    for i in range(n):
        assert np.array_equal(out[i], oracle.box(Ls[i], Rs[i], 4, 48)), i


@pytest.mark.parametrize("chunk", range(12))
def test_random_views_read_in_place_match_oracle(chunk):
    """Round 3: the matchers read device frames in place (grey frames under both strip kernels, BGR under the on-the-fly kernel).  Random
    views of larger buffers -- the frame at the very start or end of its allocation or somewhere inside, any row pitch, frames of a batch
    any distance apart, bands of a taller frame -- must give the oracle's bits and leave everything round the output untouched: the
    byte-wise edge rows, the frame-group arithmetic and the 32-bit offsets of the staged kernel's direct form are all in play."""
    import torch
    rng = np.random.default_rng(4242 + chunk)
    for case in range(16):
        win = int(rng.integers(4, 26))
        D = int(rng.choice([17, 30, 32, 48, 64, 65, 96, 128, 130, 200, 256]))
        W = int(rng.choice([384, 385, 400, 511, 512, 640, int(rng.integers(384, 1400))]))
        H = int(rng.integers(win + 1, win + 60))
        pairs = int(rng.choice([1, 1, 2, 3, 6]))
        ch = 3 if (D <= 32 and win <= 8 and rng.random() < 0.5) else 1
        top, left = int(rng.choice([0, 0, 1, 5])), int(rng.choice([0, 0, 3, 16, 41]))
        bottom, right = int(rng.choice([0, 0, 2])), int(rng.choice([0, 0, 7, 64]))
        gap = int(rng.choice([0, 0, 1, 3]))                      # extra rows between the frames of the batch
        Ls, Rs = zip(*[noise_pair(9 * case + i + 1000 * chunk, H, W, shift=1 + (i + case) % 9, binary=(case % 5 == 4), ch=ch) for i in range(pairs)])
        shp = (pairs, top + H + bottom + gap, left + W + right) + ((ch,) if ch > 1 else ())
        bl = torch.from_numpy(rng.integers(0, 256, size=shp, dtype=np.uint8)).cuda()
        br = torch.from_numpy(rng.integers(0, 256, size=shp, dtype=np.uint8)).cuda()
        vl, vr = bl[:, top:top + H, left:left + W], br[:, top:top + H, left:left + W]
        vl.copy_(torch.from_numpy(np.stack(Ls)).cuda()); vr.copy_(torch.from_numpy(np.stack(Rs)).cuda())
        p0 = (lambda a: np.ascontiguousarray(a[..., 0])) if ch > 1 else (lambda a: a)
        exps = [oracle.box(p0(Ls[i]), p0(Rs[i]), win, D) for i in range(pairs)]
        ctx = oa.Context(0)
        obuf = torch.full((pairs, H + 2, W + 6), 0xAB, dtype=torch.uint8, device="cuda")
        out = obuf[:, 1:1 + H, 3:3 + W]
        oa.compute_batch_device(vl, vr, win, D, 0, out=out, ctx=ctx)
        got = obuf.cpu().numpy()
        what = (chunk, case, win, D, H, W, pairs, ch, top, left, bottom, right, gap, ctx.last_kernel())
        assert ctx.last_kernel().endswith("+direct") or (win == 4 and D <= 64), what      # 4-wide windows: quad matcher / 256-thread staged blocks keep the planes
        for i in range(pairs):
            bad = np.argwhere(got[i, 1:1 + H, 3:3 + W] != exps[i])
            assert bad.size == 0, (what, i, len(bad), bad[:5])
        assert (got[:, 0] == 0xAB).all() and (got[:, 1 + H:] == 0xAB).all() and (got[:, :, :3] == 0xAB).all() and (got[:, :, 3 + W:] == 0xAB).all(), what
        # a band of the first frame (what one GPU of a row-band split computes), read in place as well
        if H > win + 6:
            r0 = int(rng.integers(0, H - win - 3)); nb = int(rng.integers(1, H - r0 + 1))
            i1 = min(r0 + nb + win - 1, H)
            gb = oa.compute_band_device(vl[0, r0:i1], vr[0, r0:i1], H, r0, nb, win, D, 0, ctx=ctx).cpu().numpy()
            assert np.array_equal(gb, exps[0][r0:r0 + nb]), (what, "band", r0, nb)
