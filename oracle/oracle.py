"""ctypes loader for oracle/libsad_oracle.so plus a tiny pure-numpy restatement for small cases.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/SAD.cpp:13-61 (see sad_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

RAW_LOC = 1
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsad_oracle.so")
_lib = None

_u8p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_u32 = ctypes.c_uint32


def build(force=False):
    """Compile the C restatement with gcc (no GPU, no reference sources involved)."""
    src = os.path.join(_HERE, "sad_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsad_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        common = [_u8p, _i64, _u8p, _i64, _int, _int, _int, _int, _int, _u32, _u8p, _i64]
        L.sad_oracle_literal.argtypes = common
        L.sad_oracle_literal_rows.argtypes = common + [_int, _int]
        L.sad_oracle_literal_mt.argtypes = common + [_int]
        L.sad_oracle_box.argtypes = common + [_int]
        for f in (L.sad_oracle_literal, L.sad_oracle_literal_rows, L.sad_oracle_literal_mt, L.sad_oracle_box):
            f.restype = _int
        L.sad_oracle_band.argtypes = [_u8p, _i64, _u8p, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _u32, _u8p, _i64]
        L.sad_oracle_band.restype = _int
        L.sad_oracle_rds.argtypes = [ctypes.c_uint64, _int, _int, _int, _int, _int, _u8p, _u8p]
        L.sad_oracle_rds.restype = _int
        L.sad_oracle_splitmix64.argtypes = [ctypes.c_uint64, _int]
        L.sad_oracle_splitmix64.restype = ctypes.c_uint64
        _lib = L
    return _lib


def _views(left, right):
    left = np.asarray(left)
    right = np.asarray(right)
    assert left.dtype == np.uint8 and right.dtype == np.uint8
    if left.ndim == 2:
        left = left[:, :, None]
    if right.ndim == 2:
        right = right[:, :, None]
    for a in (left, right):
        ch = a.shape[2]
        assert (ch == 1 or a.strides[2] == 1) and (a.shape[1] <= 1 or a.strides[1] == ch), "pixel-contiguous rows required"
    return left, right


def _call(fn, left, right, win, dsr, flags, extra=(), out_rows=None):
    left, right = _views(left, right)
    rows, cols, ch = left.shape
    out = np.empty((rows if out_rows is None else out_rows, cols), dtype=np.uint8)
    rc = fn(left.ctypes.data, left.strides[0], right.ctypes.data, right.strides[0], rows, cols, ch,
            win, dsr, flags, out.ctypes.data, out.strides[0] if out.size else max(cols, 1), *extra)
    if rc != 0:
        raise ValueError("sad_oracle error %d" % rc)
    return out


def literal(left, right, win=7, dsr=30, flags=0):
    return _call(lib().sad_oracle_literal, left, right, win, dsr, flags)


def literal_rows(left, right, win, dsr, row0, row1, flags=0):
    return _call(lib().sad_oracle_literal_rows, left, right, win, dsr, flags, (row0, row1), out_rows=row1 - row0)


def band(left_band, right_band, frame_rows, row0, rows_out, win, dsr, flags=0):
    """Checker twin of sadbm_compute_band_device (input rows [row0, row0+len(left_band)))."""
    lb, rb = _views(left_band, right_band)
    rows_in, cols, ch = lb.shape
    out = np.empty((rows_out, cols), dtype=np.uint8)
    rc = lib().sad_oracle_band(lb.ctypes.data, lb.strides[0], rb.ctypes.data, rb.strides[0], frame_rows, row0, rows_in,
                               rows_out, cols, ch, win, dsr, flags, out.ctypes.data, max(cols, 1))
    if rc != 0:
        raise ValueError("sad_oracle error %d" % rc)
    return out


def literal_mt(left, right, win=7, dsr=30, flags=0, threads=None):
    return _call(lib().sad_oracle_literal_mt, left, right, win, dsr, flags, (threads or os.cpu_count() or 1,))


def box(left, right, win=7, dsr=30, flags=0, threads=None):
    return _call(lib().sad_oracle_box, left, right, win, dsr, flags, (threads or min(os.cpu_count() or 1, 16),))


def rds(seed, W, H, d_bg, d_fg, binary=0):
    Lo = np.empty((H, W), dtype=np.uint8)
    Ro = np.empty((H, W), dtype=np.uint8)
    rc = lib().sad_oracle_rds(seed, W, H, d_bg, d_fg, int(binary), Lo.ctypes.data, Ro.ctypes.data)
    assert rc == 0
    return Lo, Ro


def splitmix64(seed, n):
    return int(lib().sad_oracle_splitmix64(seed, n))


def numpy_restatement(left, right, win=7, dsr=30, raw=False):
    """Third, independent formulation (numpy, plane 0 only) for small frames: the closed form of
    SURVEY.md section 8(a): cost(k)=window SAD for k<=j, 0 for k>j; first minimum; (loc*16)&255."""
    left, right = _views(left, right)
    PL = left[:, :, 0].astype(np.int64)
    PR = right[:, :, 0].astype(np.int64)
    H, W = PL.shape
    out = np.zeros((H, W), dtype=np.uint8)
    oh, ow = H - win, W - win
    if oh <= 0 or ow <= 0:
        return out
    cost = np.zeros((dsr, oh, ow), dtype=np.int64)
    for k in range(dsr):
        if k >= W:
            break
        ad = np.zeros((H, W), dtype=np.int64)
        ad[:, k:] = np.abs(PL[:, k:] - PR[:, : W - k])
        ii = np.zeros((H + 1, W + 1), dtype=np.int64)
        ii[1:, 1:] = ad.cumsum(0).cumsum(1)
        c = ii[win : win + oh, win : win + ow] - ii[:oh, win : win + ow] - ii[win : win + oh, :ow] + ii[:oh, :ow]
        c = c.copy()
        c[:, : min(k, ow)] = 0          # j < k: invalid slot keeps the zero of SAD.cpp:40
        cost[k] = c
    loc = cost.argmin(axis=0)           # numpy argmin returns the first minimum
    out[:oh, :ow] = loc.astype(np.uint8) if raw else ((loc * 16) & 255).astype(np.uint8)
    return out
