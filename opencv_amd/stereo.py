"""Python host layer over libsadbm.so (ctypes).  Mirrors /root/reference/SAD.cpp's call surface:

    Mat getDisparity(Mat &left, Mat &right)        SAD.cpp:24   with globals winSize=7 (:9), DSR=30 (:11)

`getDisparity(left, right, winSize=7, DSR=30)` takes HxW or HxWxC uint8 arrays (numpy, host) or
torch CUDA tensors (device-resident path) and returns the HxW uint8 disparity image.  torch is used
only for device memory and streams; all arithmetic happens in the HIP kernels.
"""
import collections
import ctypes
import os
import subprocess
import threading

import numpy as np

RAW_LOC = 1
FORCE_GENERIC = 2
KERNEL_GLOBAL = 4
KERNEL_STRIP = 8
KERNEL_QUAD = 16
DEFAULT_WIN_SIZE = 7     # SAD.cpp:9
DEFAULT_DSR = 30         # SAD.cpp:11

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("SADBM_LIB") or os.path.join(_CSRC, "libsadbm.so")   # SADBM_LIB: experiment builds
_lib = None


class SadbmError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("sadbm error %d: %s" % (code, text))
        self.code = code


class _Image(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("channels", ctypes.c_int32), ("step", ctypes.c_int64)]


class _Plane(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("step", ctypes.c_int64)]


def library_path():
    return _SO


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU); make decides what is stale."""
    subprocess.check_call(["make", "-C", _CSRC, "-j8"] + (["-B"] if force else []) + ["libsadbm.so"])
    return _SO


def load_library():
    """dlopen libsadbm.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise SadbmError(-100, "libsadbm.so not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "there is no CPU fallback")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same soname as
    # /opt/rocm's).  Import torch first so libsadbm.so binds to the runtime torch uses -- device
    # pointers and streams are then shared between the two; two runtimes would each see no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(_SO)
    vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32
    L.sadbm_version.restype = i32
    L.sadbm_strerror.restype = ctypes.c_char_p
    L.sadbm_strerror.argtypes = [i32]
    L.sadbm_last_hip_error.restype = ctypes.c_char_p
    L.sadbm_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.sadbm_destroy.argtypes = [vp]
    L.sadbm_destroy.restype = None
    L.sadbm_get_disparity.argtypes = [ctypes.POINTER(_Image), ctypes.POINTER(_Image), i32, i32, u32, ctypes.POINTER(_Plane)]
    L.sadbm_get_disparity_ctx.argtypes = [vp] + L.sadbm_get_disparity.argtypes
    L.sadbm_compute_device.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, u32, vp, i64, vp]
    L.sadbm_compute_band_device.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, u32, vp, i64, vp]
    L.sadbm_compute_batch_device.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, u32, vp, i64, i64, vp]
    L.sadbm_last_kernel_name.argtypes = [vp]
    L.sadbm_last_kernel_name.restype = ctypes.c_char_p
    L.sadbm_profile_begin.argtypes = [vp, i32]
    L.sadbm_profile_end.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.sadbm_pipeline_create.argtypes = [i32, i32, i32, i32, i32, i32, i32, u32, ctypes.POINTER(vp)]
    L.sadbm_pipeline_destroy.argtypes = [vp]
    L.sadbm_pipeline_destroy.restype = None
    L.sadbm_pipeline_inputs.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64)]
    L.sadbm_pipeline_submit.argtypes = [vp, ctypes.POINTER(_Image), ctypes.POINTER(_Image), ctypes.POINTER(i64)]
    L.sadbm_pipeline_wait.argtypes = [vp, i64, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(_Plane)]
    for name in ("sadbm_pipeline_create", "sadbm_pipeline_inputs", "sadbm_pipeline_submit", "sadbm_pipeline_wait"):
        getattr(L, name).restype = i32
    L.sadbm_compute_device_static.argtypes = [i32, vp, i64, vp, i64, i32, i32, i32, i32, i32, u32, vp, i64, vp]
    L.sadbm_multi_create.argtypes = [i32, ctypes.POINTER(i32), u32, ctypes.POINTER(vp)]
    L.sadbm_multi_destroy.argtypes = [vp]
    L.sadbm_multi_destroy.restype = None
    L.sadbm_compute_multi.argtypes = [vp, ctypes.POINTER(_Image), ctypes.POINTER(_Image), i32, i32, u32, i32, ctypes.POINTER(_Plane)]
    L.sadbm_multi_device_map.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    L.sadbm_compute_multi_device.argtypes = [vp, i32, vp, i64, vp, i64, i32, i32, i32, i32, i32, u32, i32, vp, i64, vp]
    L.sadbm_multi_device_id.argtypes = [vp, i32]
    L.sadbm_compute_multi_batch.argtypes = [vp, ctypes.POINTER(_Image), ctypes.POINTER(_Image), i32, i32, i32, u32, ctypes.POINTER(_Plane)]
    L.sadbm_band_plan.argtypes = [i32, i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.sadbm_reload_env.restype = None
    L.sadbm_plan_describe.argtypes = [i32, i32, i32, i32, i32, i32, u32, i32, ctypes.c_char_p, i32]
    L.sadbm_plan_describe.restype = i32
    L.sadbm_thread_release.restype = None
    for name in ("sadbm_compute_device_static", "sadbm_multi_create", "sadbm_compute_multi", "sadbm_multi_device_map",
                 "sadbm_compute_multi_batch", "sadbm_band_plan", "sadbm_compute_multi_device", "sadbm_multi_device_id"):
        getattr(L, name).restype = i32
    for name in ("sadbm_profile_begin", "sadbm_profile_end", "sadbm_create", "sadbm_get_disparity", "sadbm_get_disparity_ctx", "sadbm_compute_device",
                 "sadbm_compute_band_device", "sadbm_compute_batch_device"):
        getattr(L, name).restype = i32
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        L = load_library()
        text = L.sadbm_strerror(rc).decode()
        if rc in (-5, -9):
            text += " (" + L.sadbm_last_hip_error().decode() + ")"
        raise SadbmError(rc, text)


class Context:
    """One sadbm_ctx: device scratch + a stream.  One per (thread, device, stream); the module-level functions keep
    such a table themselves (_ctx_for) when no ctx is passed."""

    def __init__(self, device=0):
        L = load_library()
        h = ctypes.c_void_p()
        _check(L.sadbm_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            load_library().sadbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def profile_begin(self, max_launches=1024):
        _check(load_library().sadbm_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        """(n_launches, matcher_ms_total, prep_ms_total) -- HIP events around the kernels."""
        n = ctypes.c_int(0)
        m = ctypes.c_float(0)
        p = ctypes.c_float(0)
        _check(load_library().sadbm_profile_end(self._h, ctypes.byref(n), ctypes.byref(m), ctypes.byref(p)))
        return n.value, m.value, p.value

    def last_kernel(self):
        return load_library().sadbm_last_kernel_name(self._h).decode()


_default_ctx = threading.local()
_DEFAULT_CTX_MAX = 8


def _ctx_for(device, stream=None):
    """The default context of the calling thread for (device, stream): a sadbm_ctx owns scratch planes that every
    call rewrites, so threads and streams must not share one.  At most _DEFAULT_CTX_MAX per thread are kept (least
    recently used first out; closing a context waits for the device)."""
    if stream is None:
        import torch
        stream = torch.cuda.current_stream(device).cuda_stream
    table = getattr(_default_ctx, "table", None)
    if table is None:
        table = _default_ctx.table = collections.OrderedDict()
    key = (int(device), int(stream))
    c = table.get(key)
    if c is None:
        c = table[key] = Context(device)
        while len(table) > _DEFAULT_CTX_MAX:
            table.popitem(last=False)[1].close()
    else:
        table.move_to_end(key)
    return c


def plan(rows, cols, channels=1, n_pairs=1, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, cus=0):
    """sadbm_plan_describe as a dict: what a call on n_pairs whole device-resident frames would launch (needs no GPU)."""
    buf = ctypes.create_string_buffer(512)
    _check(load_library().sadbm_plan_describe(int(rows), int(cols), int(channels), int(n_pairs), int(winSize), int(DSR), int(flags),
                                              int(cus), buf, len(buf)))
    out = {}
    for kv in buf.value.decode().split():
        k, v = kv.split("=", 1)
        out[k] = int(v) if v.lstrip("-").isdigit() else v
    return out


def reload_env():
    """Re-read the SADBM_* tuning knobs (the library reads the environment once, at its first call)."""
    load_library().sadbm_reload_env()


def thread_release():
    """Release the calling thread's implicit context of the host API now (it also goes when the thread exits)."""
    load_library().sadbm_thread_release()


def _host_views(left, right):
    left = np.asarray(left)
    right = np.asarray(right)
    if left.dtype != np.uint8 or right.dtype != np.uint8:
        raise SadbmError(-2, "images must be uint8 (CV_8U)")
    if left.ndim == 2:
        left = left[:, :, None]
    if right.ndim == 2:
        right = right[:, :, None]
    if left.ndim != 3 or right.ndim != 3:
        raise SadbmError(-3, "images must be HxW or HxWxC")
    out = []
    for a in (left, right):
        # ROI views with a row stride are passed through (cv::Mat step); pixels must be contiguous
        ch = a.shape[2]
        ok = (ch == 1 or a.strides[2] == 1) and (a.shape[1] <= 1 or a.strides[1] == ch) and \
             (a.shape[0] <= 1 or a.strides[0] >= a.shape[1] * ch)
        out.append(a if ok else np.ascontiguousarray(a))
    return out


def get_disparity_host(left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, ctx=None):
    """Host arrays in, host array out (sadbm_get_disparity / _ctx)."""
    L = load_library()
    left, right = _host_views(left, right)
    rows, cols, ch = left.shape
    out = np.empty((rows, cols), dtype=np.uint8)
    li = _Image(left.ctypes.data, rows, cols, ch, left.strides[0] if rows > 1 else max(cols * ch, 1))
    ri = _Image(right.ctypes.data, right.shape[0], right.shape[1], right.shape[2],
                right.strides[0] if right.shape[0] > 1 else max(right.shape[1] * right.shape[2], 1))
    op = _Plane(out.ctypes.data, rows, cols, max(cols, 1))
    if ctx is None:
        _check(L.sadbm_get_disparity(ctypes.byref(li), ctypes.byref(ri), int(winSize), int(DSR), int(flags), ctypes.byref(op)))
    else:
        _check(L.sadbm_get_disparity_ctx(ctx.handle, ctypes.byref(li), ctypes.byref(ri), int(winSize), int(DSR), int(flags), ctypes.byref(op)))
    return out


def _tensor_view(t):
    """(ptr, pitch, rows, cols, channels) of a uint8 CUDA tensor HxW or HxWxC with contiguous pixels."""
    import torch
    if t.dtype != torch.uint8 or not t.is_cuda:
        raise SadbmError(-2, "device path needs uint8 CUDA tensors")
    if t.dim() == 2:
        t = t.unsqueeze(-1)
    if t.dim() != 3:
        raise SadbmError(-3, "images must be HxW or HxWxC")
    rows, cols, ch = t.shape
    if cols > 0 and rows > 0 and (t.stride(2) != 1 or (cols > 1 and t.stride(1) != ch)):
        t = t.contiguous()
    pitch = t.stride(0) if rows > 1 else max(cols * ch, 1)
    return t, t.data_ptr(), pitch, rows, cols, ch


def _storage_range(t):
    """[first, last) byte addresses a strided tensor can touch."""
    if t.numel() == 0:
        return (0, 0)
    last = sum((n - 1) * st for n, st in zip(t.shape, t.stride()))
    return (t.data_ptr(), t.data_ptr() + (last + 1) * t.element_size())


def _check_out(out, rows, cols, *inputs):
    """A caller-supplied output plane goes straight to the kernels: rows x cols uint8 on the inputs' device, unit column stride,
    and not overlapping an input (the direct-read kernels read the inputs while they write the output)."""
    import torch
    if (out.dtype != torch.uint8 or tuple(out.shape) != (rows, cols) or out.device != inputs[0].device
            or (cols > 1 and out.stride(1) != 1) or (rows > 1 and out.stride(0) < cols)):
        raise SadbmError(-3, "out must be a rows x cols uint8 tensor on the inputs' device with unit column stride")
    o0, o1 = _storage_range(out)
    for t in inputs:
        i0, i1 = _storage_range(t)
        if o0 < i1 and i0 < o1:
            raise SadbmError(-3, "out overlaps an input image")


def _stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def compute_device(left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, out=None, ctx=None):
    """Device-resident getDisparity on torch CUDA tensors (sadbm_compute_device), async on the
    current torch stream."""
    import torch
    L = load_library()
    lt, lp, lpitch, rows, cols, ch = _tensor_view(left)
    rt, rp, rpitch, rrows, rcols, rch = _tensor_view(right)
    if rch != ch:
        raise SadbmError(-2, L.sadbm_strerror(-2).decode())
    if rrows < rows or rcols < cols:
        raise SadbmError(-3, L.sadbm_strerror(-3).decode())
    dev = left.device.index or 0
    ctx = ctx or _ctx_for(dev)
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.uint8, device=left.device)
    else:
        _check_out(out, rows, cols, lt, rt)
    opitch = out.stride(0) if rows > 1 else max(cols, 1)
    with torch.cuda.device(dev):
        _check(L.sadbm_compute_device(ctx.handle, lp, lpitch, rp, rpitch, rows, cols, ch, int(winSize), int(DSR), int(flags),
                                      out.data_ptr(), opitch, _stream_ptr(dev)))
    return out


def compute_device_static(left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, out=None):
    """sadbm_compute_device_static: the device path without a context and without any allocation inside the library
    (static device scratch; frames up to one 4K pair).  `out` must be given if the caller may not allocate either."""
    import torch
    L = load_library()
    lt, lp, lpitch, rows, cols, ch = _tensor_view(left)
    rt, rp, rpitch, rrows, rcols, rch = _tensor_view(right)
    if rch != ch:
        raise SadbmError(-2, L.sadbm_strerror(-2).decode())
    if rrows < rows or rcols < cols:
        raise SadbmError(-3, L.sadbm_strerror(-3).decode())
    dev = left.device.index or 0
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.uint8, device=left.device)
    else:
        _check_out(out, rows, cols, lt, rt)
    opitch = out.stride(0) if rows > 1 else max(cols, 1)
    with torch.cuda.device(dev):
        _check(L.sadbm_compute_device_static(dev, lp, lpitch, rp, rpitch, rows, cols, ch, int(winSize), int(DSR), int(flags),
                                             out.data_ptr(), opitch, _stream_ptr(dev)))
    return out


def compute_band_device(left_band, right_band, frame_rows, row0, rows_out, winSize, DSR, flags=0, out=None, ctx=None):
    """Row band [row0, row0+rows_out) of a frame of `frame_rows` rows; left_band/right_band hold input
    rows [row0, row0+left_band.shape[0]) (sadbm_compute_band_device)."""
    import torch
    L = load_library()
    lt, lp, lpitch, rows_in, cols, ch = _tensor_view(left_band)
    rt, rp, rpitch, rrows, rcols, rch = _tensor_view(right_band)
    if rch != ch:
        raise SadbmError(-2, L.sadbm_strerror(-2).decode())
    if rrows < rows_in or rcols < cols:
        raise SadbmError(-3, L.sadbm_strerror(-3).decode())
    dev = left_band.device.index or 0
    ctx = ctx or _ctx_for(dev)
    if out is None:
        out = torch.empty((rows_out, cols), dtype=torch.uint8, device=left_band.device)
    else:
        _check_out(out, int(rows_out), cols, lt, rt)
    opitch = out.stride(0) if rows_out > 1 else max(cols, 1)
    with torch.cuda.device(dev):
        _check(L.sadbm_compute_band_device(ctx.handle, lp, lpitch, rp, rpitch, int(frame_rows), int(row0), rows_in,
                                           int(rows_out), cols, ch, int(winSize), int(DSR), int(flags),
                                           out.data_ptr(), opitch, _stream_ptr(dev)))
    return out


def compute_batch_device(left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, out=None, ctx=None):
    """N pairs at once: left/right are NxHxW (or NxHxWxC) uint8 CUDA tensors with contiguous pixels; views with a row pitch
    and a frame stride (crops of a larger batch) are passed through when left and right share them, anything else is copied."""
    import torch
    L = load_library()
    if left.dim() == 3:
        left = left.unsqueeze(-1)
        right = right.unsqueeze(-1)
    if left.shape != right.shape:
        raise SadbmError(-3, "batched left/right must have equal shapes")
    n, rows, cols, ch = left.shape

    def _pitched(t):
        return (t.stride(3) == 1 or ch == 1) and (cols <= 1 or t.stride(2) == ch) and (rows <= 1 or t.stride(1) >= cols * ch) and \
               (n <= 1 or t.stride(0) >= rows * (t.stride(1) if rows > 1 else cols * ch))
    if not (_pitched(left) and _pitched(right) and (rows <= 1 or left.stride(1) == right.stride(1)) and (n <= 1 or left.stride(0) == right.stride(0))):
        left = left.contiguous()
        right = right.contiguous()
    pitch_in = left.stride(1) if rows > 1 else cols * ch
    stride_in = left.stride(0) if n > 1 else rows * pitch_in
    dev = left.device.index or 0
    ctx = ctx or _ctx_for(dev)
    if out is None:
        out = torch.empty((n, rows, cols), dtype=torch.uint8, device=left.device)
    elif (out.dtype != torch.uint8 or tuple(out.shape) != (n, rows, cols) or out.device != left.device
          or (cols > 1 and out.stride(2) != 1)):
        raise SadbmError(-3, "out must be an NxHxW uint8 tensor on the inputs' device with unit column stride")
    with torch.cuda.device(dev):
        _check(L.sadbm_compute_batch_device(ctx.handle, left.data_ptr(), right.data_ptr(), pitch_in, stride_in, n,
                                            rows, cols, ch, int(winSize), int(DSR), int(flags), out.data_ptr(),
                                            out.stride(1), out.stride(0), _stream_ptr(dev)))   # pitched views allowed
    return out


def getDisparity(left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0):
    """Drop-in for SAD.cpp:24 `Mat getDisparity(Mat &left, Mat &right)`.

    numpy arrays -> host path (H2D, kernels, D2H, synchronous); torch CUDA tensors -> device path
    (asynchronous on the current stream).  Returns an HxW uint8 image of (loc*16)&255 (SAD.cpp:57).
    """
    try:
        import torch
        if isinstance(left, torch.Tensor):
            if left.is_cuda:
                return compute_device(left, right, winSize, DSR, flags)
            return torch.from_numpy(get_disparity_host(left.numpy(), right.numpy(), winSize, DSR, flags))
    except ImportError:
        pass
    return get_disparity_host(left, right, winSize, DSR, flags)


class Pipeline:
    """Streaming getDisparity over host frames (sadbm_pipeline_*): `depth` slots with pinned buffers and their own
    streams, so the H2D copy, the matcher and the D2H copy of consecutive frames overlap.

        pipe = Pipeline(rows, cols, depth=3)
        t = pipe.submit(left, right)          # numpy u8 frames; returns a ticket at once
        disp = pipe.wait(t)                   # numpy view of the pinned result (valid until the slot is reused)
    """

    def __init__(self, rows, cols, channels=1, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, depth=3, device=0):
        L = load_library()
        h = ctypes.c_void_p()
        _check(L.sadbm_pipeline_create(int(device), int(depth), int(rows), int(cols), int(channels), int(winSize), int(DSR),
                                       int(flags), ctypes.byref(h)))
        self._h = h
        self.rows, self.cols, self.channels, self.depth = int(rows), int(cols), int(channels), int(depth)

    def close(self):
        if getattr(self, "_h", None):
            load_library().sadbm_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inputs(self):
        """(left, right) numpy views of the next slot's pinned input buffers: fill them, then submit()."""
        import numpy as np
        lp, rp, step = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        _check(load_library().sadbm_pipeline_inputs(self._h, ctypes.byref(lp), ctypes.byref(rp), ctypes.byref(step)))
        rowb = self.cols * self.channels

        def view(ptr):
            buf = (ctypes.c_uint8 * (step.value * self.rows)).from_address(ptr.value)
            a = np.frombuffer(buf, dtype=np.uint8).reshape(self.rows, step.value)[:, :rowb]
            return a if self.channels == 1 else a.reshape(self.rows, self.cols, self.channels)

        return view(lp), view(rp)

    def submit(self, left=None, right=None):
        """Enqueue one frame; left/right None = the pinned inputs() were filled in place."""
        t = ctypes.c_int64()
        if left is None:
            _check(load_library().sadbm_pipeline_submit(self._h, None, None, ctypes.byref(t)))
            return t.value
        left, right = _host_views(left, right)                 # (rows, cols, ch) u8 views; kept alive until the call returns
        li = _Image(left.ctypes.data, left.shape[0], left.shape[1], left.shape[2], left.strides[0])
        ri = _Image(right.ctypes.data, right.shape[0], right.shape[1], right.shape[2], right.strides[0])
        _check(load_library().sadbm_pipeline_submit(self._h, ctypes.byref(li), ctypes.byref(ri), ctypes.byref(t)))
        return t.value

    def wait(self, ticket, copy=False):
        import numpy as np
        ptr, step = ctypes.c_void_p(), ctypes.c_int64()
        _check(load_library().sadbm_pipeline_wait(self._h, int(ticket), ctypes.byref(ptr), ctypes.byref(step), None))
        buf = (ctypes.c_uint8 * (step.value * self.rows)).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=np.uint8).reshape(self.rows, step.value)[:, :self.cols]
        return a.copy() if copy else a


MULTI_COPY_GATHER = 1
MULTI_FORCE_PEER_COPY = 2


def band_plan(rows, parts, win, index):
    """(row0, row1, in1) of sadbm_band_plan: output rows [row0, row1), input rows [row0, in1)."""
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _check(load_library().sadbm_band_plan(int(rows), int(parts), int(win), int(index), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def _img(a):
    return _Image(a.ctypes.data, a.shape[0], a.shape[1], a.shape[2], a.strides[0] if a.shape[0] > 1 else max(a.shape[1] * a.shape[2], 1))


class Multi:
    """sadbm_multi: single-process multi-GPU (one frame as row bands + one RCCL all-gather, or a batch of pairs dealt
    round-robin).  devices=None means 0..ngpu-1; multi_flags=MULTI_COPY_GATHER replaces RCCL by peer copies."""

    def __init__(self, ngpu, devices=None, multi_flags=0):
        L = load_library()
        h = ctypes.c_void_p()
        arr = (ctypes.c_int * ngpu)(*devices) if devices is not None else None
        _check(L.sadbm_multi_create(int(ngpu), arr, int(multi_flags), ctypes.byref(h)))
        self._h = h
        self.ngpu = int(ngpu)

    def close(self):
        if getattr(self, "_h", None):
            load_library().sadbm_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute(self, left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, sub_bands=1):
        left, right = _host_views(left, right)
        rows, cols, _ = left.shape
        out = np.empty((rows, cols), dtype=np.uint8)
        op = _Plane(out.ctypes.data, rows, cols, max(cols, 1))
        li, ri = _img(left), _img(right)
        _check(load_library().sadbm_compute_multi(self._h, ctypes.byref(li), ctypes.byref(ri), int(winSize), int(DSR), int(flags),
                                                  int(sub_bands), ctypes.byref(op)))
        return out

    def device_id(self, index):
        """The HIP device behind entry `index` of the handle's device list."""
        return load_library().sadbm_multi_device_id(self._h, int(index))

    def device_map(self, index, rows, cols):
        """Copy (torch uint8 tensor on that GPU) of GPU `index`'s full map of the last frame."""
        import torch
        ptr, pitch = ctypes.c_void_p(), ctypes.c_int64()
        _check(load_library().sadbm_multi_device_map(self._h, int(index), ctypes.byref(ptr), ctypes.byref(pitch)))
        dev = self.device_id(index)                             # the pointer lives on THAT GPU, whatever the current device is

        class _Raw:                                             # raw device pointer -> torch via the CUDA array interface
            __cuda_array_interface__ = {"shape": (rows, cols), "typestr": "|u1", "data": (ptr.value, False), "version": 2,
                                        "strides": (pitch.value, 1)}
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)                         # the async device form may still be filling the map
            return torch.as_tensor(_Raw(), device=torch.device("cuda", dev)).clone()

    def compute_device(self, left, right, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0, sub_bands=1, src_index=None, out=None):
        """sadbm_compute_multi_device: left/right are uint8 CUDA tensors (HxW or HxWxC) on one of the handle's GPUs; returns
        the map as a tensor on that GPU (asynchronous on its current stream); every GPU keeps a copy (device_map)."""
        import torch
        L = load_library()
        lt, lp, lpitch, rows, cols, ch = _tensor_view(left)
        rt, rp, rpitch, rrows, rcols, rch = _tensor_view(right)
        if rch != ch:
            raise SadbmError(-2, L.sadbm_strerror(-2).decode())
        if rrows < rows or rcols < cols:
            raise SadbmError(-3, L.sadbm_strerror(-3).decode())
        dev = left.device.index or 0
        if src_index is None:
            src_index = next((i for i in range(self.ngpu) if self.device_id(i) == dev), -1)
        if src_index < 0 or self.device_id(src_index) != dev:
            raise SadbmError(-4, "the inputs are not on one of the handle's GPUs")
        if out is None:
            out = torch.empty((rows, cols), dtype=torch.uint8, device=left.device)
        _check_out(out, rows, cols, left, right)
        opitch = out.stride(0) if rows > 1 else max(cols, 1)
        with torch.cuda.device(dev):
            _check(L.sadbm_compute_multi_device(self._h, int(src_index), lp, lpitch, rp, rpitch, rows, cols, ch, int(winSize), int(DSR), int(flags),
                                                int(sub_bands), out.data_ptr(), opitch, _stream_ptr(dev)))
        return out

    def compute_batch(self, lefts, rights, winSize=DEFAULT_WIN_SIZE, DSR=DEFAULT_DSR, flags=0):
        n = len(lefts)
        views = [_host_views(l, r) for l, r in zip(lefts, rights)]
        rows, cols, _ = views[0][0].shape
        outs = [np.empty((rows, cols), dtype=np.uint8) for _ in range(n)]
        li = (_Image * n)(*[_img(v[0]) for v in views])
        ri = (_Image * n)(*[_img(v[1]) for v in views])
        po = (_Plane * n)(*[_Plane(o.ctypes.data, rows, cols, max(cols, 1)) for o in outs])
        _check(load_library().sadbm_compute_multi_batch(self._h, li, ri, n, int(winSize), int(DSR), int(flags), po))
        return outs
