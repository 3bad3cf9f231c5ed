"""Shared test helpers (fixtures are data; the oracle is the checker)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def read_pnm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    magic, dims, maxv, body = parts
    w, h = (int(v) for v in dims.split())
    assert maxv == b"255"
    if magic == b"P5":
        return np.frombuffer(body, dtype=np.uint8).reshape(h, w).copy()
    assert magic == b"P6"
    return np.frombuffer(body, dtype=np.uint8).reshape(h, w, 3).copy()


def golden_meta():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


def tsukuba_bgr():
    """left/right as cv::imread(path) would return them: 8-bit 3-channel BGR (SAD.cpp:65-66)."""
    l = read_pnm(os.path.join(GOLDEN, "tsukuba_left.ppm"))[:, :, ::-1]
    r = read_pnm(os.path.join(GOLDEN, "tsukuba_right.ppm"))[:, :, ::-1]
    return np.ascontiguousarray(l), np.ascontiguousarray(r)


def noise_pair(seed, H, W, shift=None, binary=False, ch=1):
    rng = np.random.default_rng(seed)
    shape = (H, W) if ch == 1 else (H, W, ch)
    L = rng.integers(0, 256, size=shape, dtype=np.uint8)
    R = rng.integers(0, 256, size=shape, dtype=np.uint8)
    if shift is not None:
        R[:, : W - shift] = L[:, shift:]
    if binary:
        L = np.where(L & 0x80, 255, 0).astype(np.uint8)
        R = np.where(R & 0x80, 255, 0).astype(np.uint8)
    return L, R


def strip_kernel_name(win, D, staged=False, cols=None):
    """(cols: the frame width -- a search wider than the frame is run as the search over [0, cols), SAD.cpp:44.)  Which member of the strip family run_pipeline launches under KERNEL_STRIP (sadbm.hip, otf_is_default): the
    on-the-fly form for searches of at most 64 disparities (and narrow windows at 65-128 disparities that are not a multiple
    of 16), unless SADBM_OTF=0 (`staged`)."""
    if cols is not None:
        D = min(D, cols)
    if not staged and ((D <= 64 and (win >= 5 or D <= 16)) or (64 < D <= 128 and D % 16 and 5 <= win <= 11 and win != 10)):
        return "sad_strip_otf_kernel<%d,%d>" % (win, 1 if D <= 16 else 2 if D <= 32 else 4 if D <= 64 else 8)
    return "sad_strip_kernel<%d>" % win


def kname(ctx):
    """Last kernel of a context without the "+direct" suffix (the on-the-fly form reading the caller's images itself)."""
    return ctx.last_kernel().split("+")[0]
