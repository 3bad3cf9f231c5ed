"""ctypes loader for oracle/libfilter_oracle.so (Gaussian low/high-pass programs of the reference)
plus a numpy restatement used to cross-check it.  TEST INFRASTRUCTURE ONLY.  See filter_oracle.c for the
reference lines each function follows and for the 'parity unpinned' statement."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfilter_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "filter_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libfilter_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _c3(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a if a.ndim == 3 else a[:, :, None]


def gaussian_kernel(size, sigma):
    """GetgaussianKernel (gaus_filter.cpp:15-40): integer-divided exponent, normalised, float64 size x size."""
    out = np.empty((size, size), np.float64)
    assert lib().filter_oracle_gaussian_kernel(int(size), int(sigma), out.ctypes.data_as(ctypes.c_void_p)) == 0
    return out


def gaussian_kernel_np(size, sigma):
    c = size // 2
    d = np.arange(size) - c
    r2 = d[:, None] ** 2 + d[None, :] ** 2
    q = -(r2 // (2 * sigma * sigma))            # r2 >= 0: C's truncating division of -r2 equals -(r2 // den)
    k = np.exp(q.astype(np.float64))
    s = 0.0
    for v in k.ravel():                         # same summation order as the C loop
        s += v
    return k / s


def filter2d(src, kern):
    s = _c3(src)
    k = np.ascontiguousarray(kern, dtype=np.float64)
    out = np.empty_like(s)
    rc = lib().filter_oracle_filter2d(s.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(s.strides[0]), s.shape[0], s.shape[1], s.shape[2],
                                      k.ctypes.data_as(ctypes.c_void_p), k.shape[0], k.shape[1],
                                      out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(out.strides[0]))
    assert rc == 0
    return out.reshape(np.asarray(src).shape)


def filter2d_np(src, kern):
    """Same definition with numpy: taps accumulated in row-major order in float64 (bit-identical sums)."""
    s = _c3(src).astype(np.float64)
    kr, kc = kern.shape
    ay, ax = kr // 2, kc // 2
    p = np.pad(s, ((ay, kr - 1 - ay), (ax, kc - 1 - ax), (0, 0)), mode="reflect") if min(s.shape[:2]) > max(kr, kc) else None
    assert p is not None, "numpy cross-check needs an image larger than the kernel"
    H, W = s.shape[:2]
    acc = np.zeros_like(s)
    for i in range(kr):
        for j in range(kc):
            acc = acc + kern[i, j] * p[i:i + H, j:j + W]
    out = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out.reshape(np.asarray(src).shape)


def highpass(src, low, add=0):
    a, b = _c3(src), _c3(low)
    out = np.empty_like(a)
    rc = lib().filter_oracle_highpass(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(a.strides[0]), b.ctypes.data_as(ctypes.c_void_p),
                                      ctypes.c_int64(b.strides[0]), a.shape[0], a.shape[1] * a.shape[2], int(add),
                                      out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(out.strides[0]))
    assert rc == 0
    return out.reshape(np.asarray(src).shape)


def blend(m1, m2, alpha):
    a, b = _c3(m1), _c3(m2)
    out = np.empty_like(a)
    rc = lib().filter_oracle_blend(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(a.strides[0]), b.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_int64(b.strides[0]), a.shape[0], a.shape[1], a.shape[2], ctypes.c_double(alpha),
                                   out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(out.strides[0]))
    assert rc == 0
    return out
