"""Gaussian low/high-pass path of the reference's filter programs (gaus_filter.cpp, hybrid_image.cpp) over the C ABI
of include/sadbm_filters.h.  Names follow the reference: GetgaussianKernel, filter2D, addWeighted(src,1,low,-1,0)
(here `highpass`), add(mat1, mat2, alpha) (here `blend`).  Inputs are torch CUDA uint8 tensors (H, W) or (H, W, C);
there is no CPU fallback."""
import ctypes

import numpy as np

from .stereo import SadbmError, _Image, _check, _ctx_for, load_library

_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        L.sadbm_gaussian_kernel.argtypes = [i32, i32, vp]
        L.sadbm_filter2d_device.argtypes = [vp, vp, i64, i32, i32, i32, vp, i32, i32, vp, i64, vp]
        L.sadbm_highpass_device.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, vp, i64, vp]
        L.sadbm_blend_device.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, dbl, vp, i64, vp]
        L.sadbm_ctx_device.argtypes = [vp]
        u8p, img = ctypes.c_void_p, ctypes.POINTER(_Image)
        L.sadbm_gaus_filter_host.argtypes = [img, i32, i32, u8p, i64, u8p, i64]
        L.sadbm_hybrid_image_host.argtypes = [img, img, i32, i32, i32, dbl, u8p, i64, u8p, i64, u8p, i64]
        for n in ("sadbm_gaussian_kernel", "sadbm_filter2d_device", "sadbm_highpass_device", "sadbm_blend_device", "sadbm_ctx_device",
                  "sadbm_gaus_filter_host", "sadbm_hybrid_image_host"):
            getattr(L, n).restype = i32
        _bound = True
    return L


def GetgaussianKernel(size, sigma):
    """size x size float64 kernel of gaus_filter.cpp:15-40 (integer-divided exponent, normalised)."""
    out = np.empty((int(size), int(size)), np.float64)
    _check(_lib().sadbm_gaussian_kernel(int(size), int(sigma), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def _view(t, is_out=False):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() in (2, 3)):
        raise TypeError("expected a CUDA uint8 tensor (H, W) or (H, W, C)")
    ch = 1 if t.dim() == 2 else t.shape[2]
    if t.stride(-1) != 1 or (t.dim() == 3 and t.stride(1) != ch):
        if is_out:          # a copy would receive the result instead of the caller's tensor
            raise SadbmError(-3, "out must have contiguous pixels (unit channel stride, pixel stride = channels)")
        t = t.contiguous()
    return t, t.data_ptr(), t.stride(0), t.shape[0], t.shape[1], ch


def _out_view(out, like, rows, cols, ch):
    import torch
    if out is None:
        out = torch.empty_like(like)
    o, op, opitch, orows, ocols, och = _view(out, is_out=True)
    if (orows, ocols, och) != (rows, cols, ch) or out.device != like.device:
        raise SadbmError(-3, "out must have the source's shape and device")
    return out, op, opitch


def _launch(t):
    """(context, stream pointer, device guard) for the device `t` lives on -- not the process-current device."""
    import torch
    dev = t.device.index or 0
    st = torch.cuda.current_stream(dev).cuda_stream
    return _ctx_for(dev, st), ctypes.c_void_p(st), torch.cuda.device(dev)


def filter2D(src, kernel, out=None):
    """cv::filter2D(src, dst, src.depth(), kernel) as defined in include/sadbm_filters.h."""
    s, sp, spitch, rows, cols, ch = _view(src)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    if k.ndim != 2:
        raise SadbmError(-4, "kernel must be 2-D")
    out, op, opitch = _out_view(out, s, rows, cols, ch)
    ctx, st, guard = _launch(s)
    with guard:
        _check(_lib().sadbm_filter2d_device(ctx.handle, sp, spitch, rows, cols, ch, k.ctypes.data_as(ctypes.c_void_p), k.shape[0], k.shape[1],
                                            op, opitch, st))
    return out


def highpass(src, low, add=0, out=None):
    """addWeighted(src, 1, low, -1, 0): saturate(saturate(src + add) - low)."""
    s, sp, spitch, rows, cols, ch = _view(src)
    l, lp, lpitch, lrows, lcols, lch = _view(low)
    if (lrows, lcols, lch) != (rows, cols, ch) or l.device != s.device:
        raise SadbmError(-3, "shapes or devices differ")
    out, op, opitch = _out_view(out, s, rows, cols, ch)
    ctx, st, guard = _launch(s)
    with guard:
        _check(_lib().sadbm_highpass_device(ctx.handle, sp, spitch, lp, lpitch, rows, cols * ch, int(add), op, opitch, st))
    return out


def blend(mat1, mat2, alpha, out=None):
    """add(mat1, mat2, alpha) of hybrid_image.cpp:78-100."""
    a, ap, apitch, rows, cols, ch = _view(mat1)
    b, bp, bpitch, brows, bcols, bch = _view(mat2)
    if (brows, bcols, bch) != (rows, cols, ch) or b.device != a.device:
        raise SadbmError(-3, "shapes or devices differ")
    out, op, opitch = _out_view(out, a, rows, cols, ch)
    ctx, st, guard = _launch(a)
    with guard:
        _check(_lib().sadbm_blend_device(ctx.handle, ap, apitch, bp, bpitch, rows, cols, ch, ctypes.c_double(alpha), op, opitch, st))
    return out


def _host_image(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3:
        raise SadbmError(-3, "images must be HxW or HxWxC")
    return a, _Image(a.ctypes.data, a.shape[0], a.shape[1], a.shape[2], max(a.shape[1] * a.shape[2], 1))


def gaus_filter_host(src, size=35, sigma=3):
    """main() of gaus_filter.cpp:42-70 on a host image (numpy): (low_pass, high_pass).  Defaults = the program's #defines."""
    a, ia = _host_image(src)
    low, high = np.empty_like(a), np.empty_like(a)
    step = max(a.shape[1] * a.shape[2], 1)
    _check(_lib().sadbm_gaus_filter_host(ctypes.byref(ia), int(size), int(sigma), low.ctypes.data, step, high.ctypes.data, step))
    return low.reshape(np.asarray(src).shape), high.reshape(np.asarray(src).shape)


def hybrid_image_host(img1, img2, size=31, sigma=10, add=12, alpha=0.6):
    """main() of hybrid_image.cpp:102-143 on host images: (low_pass of img1, high_pass of img2, hybrid)."""
    a, ia = _host_image(img1)
    b, ib = _host_image(img2)
    low, high, mix = np.empty_like(a), np.empty_like(a), np.empty_like(a)
    step = max(a.shape[1] * a.shape[2], 1)
    _check(_lib().sadbm_hybrid_image_host(ctypes.byref(ia), ctypes.byref(ib), int(size), int(sigma), int(add), ctypes.c_double(alpha),
                                          low.ctypes.data, step, high.ctypes.data, step, mix.ctypes.data, step))
    return low, high, mix
