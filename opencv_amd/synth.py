"""Synthetic random-dot stereograms for benchmarks and demos: the generator of SURVEY.md section 8(d) d2
(splitmix64; left image = the first W*H draws >> 56, an independent noise image = the next W*H draws; the right
image is the left one shifted by d_fg inside the centred half-size box and by d_bg elsewhere, noise where the
shift leaves the frame).  Product-side twin of the checker's generator -- tests verify the two agree."""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_stream(seed, n, start=0):
    """Outputs start+1 .. start+n of splitmix64(seed) as uint64 (state after i draws = seed + i*gamma, so it vectorises)."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def rds(seed, W, H, d_bg, d_fg, binary=0):
    """(left, right) uint8 H x W random-dot stereogram with planted disparities d_bg (background) and d_fg (box)."""
    n = W * H
    L = (splitmix64_stream(seed, n) >> np.uint64(56)).astype(np.uint8).reshape(H, W)
    N = (splitmix64_stream(seed, n, start=n) >> np.uint64(56)).astype(np.uint8).reshape(H, W)
    if binary:
        L = np.where(L & 0x80, 255, 0).astype(np.uint8)
        N = np.where(N & 0x80, 255, 0).astype(np.uint8)
    y = np.arange(H)[:, None]
    x = np.arange(W)[None, :]
    d = np.where((y >= H // 4) & (y < 3 * H // 4) & (x >= W // 4) & (x < 3 * W // 4), d_fg, d_bg)
    xs = x + d
    R = np.where(xs < W, np.take_along_axis(L, np.minimum(xs, W - 1), axis=1), N)
    return np.ascontiguousarray(L), np.ascontiguousarray(R.astype(np.uint8))
