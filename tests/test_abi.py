"""CPU tests of the boundary: libsadbm.so loads, exports every symbol include/sadbm.h declares,
argument validation mirrors the reference's failure modes, and there is no CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest

import opencv_amd as oa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for h in ("sadbm.h", "sadbm_filters.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(sadbm_[a-z_0-9]+)\s*\(", text))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    lib = oa.load_library()
    syms = declared_symbols()
    assert {"sadbm_get_disparity", "sadbm_compute_device", "sadbm_compute_band_device",
            "sadbm_compute_batch_device", "sadbm_create", "sadbm_destroy"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.sadbm_version() == 200


def test_strerror_covers_codes():
    lib = oa.load_library()
    texts = {lib.sadbm_strerror(c) for c in range(0, -10, -1)}
    assert len(texts) == 10


def test_multi_and_static_entry_points_validate_before_touching_a_device():
    lib = oa.load_library()
    assert lib.sadbm_compute_device_static(0, 1, 8, 1, 8, 8, 8, 2, 3, 8, 0, 1, 8, None) == -2     # channels
    assert lib.sadbm_compute_device_static(0, 1, 4, 1, 8, 8, 8, 1, 3, 8, 0, 1, 8, None) == -3     # pitch < row bytes
    assert lib.sadbm_compute_device_static(0, None, 8, 1, 8, 8, 8, 1, 3, 8, 0, 1, 8, None) == -1
    assert lib.sadbm_compute_device_static(0, 1, 8, 1, 8, 0, 8, 1, 3, 8, 0, 1, 8, None) == 0      # empty image: nothing to do
    assert lib.sadbm_compute_multi(None, None, None, 7, 30, 0, 1, None) == -1
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.sadbm_band_plan(10, 0, 7, 0, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == -4
    h = ctypes.c_void_p()
    assert lib.sadbm_multi_create(0, None, 0, ctypes.byref(h)) == -4


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = np.zeros((16, 16), np.uint8)
    with pytest.raises(oa.SadbmError) as e:
        oa.getDisparity(L, L)
    assert e.value.code == -5          # SADBM_E_HIP: refuses, never computes on the CPU


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "opencv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace("the oracle", ""), f


def test_argument_validation_happens_before_device_use():
    """NULL / type / size / parameter errors are detected on the host (reference: cv::Exception from
    absdiff on channel mismatch, from Mat::operator()(Rect) when right is smaller)."""
    lib = oa.load_library()
    from opencv_amd.stereo import _Image, _Plane
    buf = np.zeros((8, 8, 3), np.uint8)
    out = np.zeros((8, 8), np.uint8)

    def img(rows, cols, ch):
        return _Image(buf.ctypes.data, rows, cols, ch, 24)

    pl = _Plane(out.ctypes.data, 8, 8, 8)
    fake = ctypes.c_void_p(1)          # non-null context handle: validation must not dereference it
    f = lib.sadbm_get_disparity_ctx
    assert f(None, ctypes.byref(img(8, 8, 1)), ctypes.byref(img(8, 8, 1)), 3, 8, 0, ctypes.byref(pl)) == -1
    assert f(fake, ctypes.byref(img(8, 8, 2)), ctypes.byref(img(8, 8, 2)), 3, 8, 0, ctypes.byref(pl)) == -2
    assert f(fake, ctypes.byref(img(8, 8, 3)), ctypes.byref(img(8, 8, 1)), 3, 8, 0, ctypes.byref(pl)) == -2
    assert f(fake, ctypes.byref(img(8, 8, 1)), ctypes.byref(img(7, 8, 1)), 3, 8, 0, ctypes.byref(pl)) == -3
    assert f(fake, ctypes.byref(img(8, 8, 1)), ctypes.byref(img(8, 8, 1)), 0, 8, 0, ctypes.byref(pl)) == -4
    assert f(fake, ctypes.byref(img(8, 8, 1)), ctypes.byref(img(8, 8, 1)), 3, 0, 0, ctypes.byref(pl)) == -4
    assert f(fake, ctypes.byref(img(8, 8, 1)), ctypes.byref(img(8, 8, 1)), 3, 257, 1, ctypes.byref(pl)) == -4   # raw loc is a byte: DSR <= 256
    g = lib.sadbm_compute_device
    assert g(None, 1, 8, 1, 8, 8, 8, 1, 3, 8, 0, 1, 8, None) == -1
    assert g(fake, 1, 4, 1, 8, 8, 8, 1, 3, 8, 0, 1, 8, None) == -3        # pitch < row bytes
    assert g(fake, None, 8, 1, 8, 8, 8, 1, 3, 8, 0, 1, 8, None) == -1
    b = lib.sadbm_compute_band_device
    assert b(fake, 1, 8, 1, 8, 16, 4, 5, 4, 8, 1, 3, 8, 0, 1, 8, None) == -3   # rows_in < rows_out + win - 1
    assert b(fake, 1, 8, 1, 8, 16, 14, 4, 2, 8, 1, 3, 8, 0, 1, 8, None) == -3  # band beyond the frame


def test_cv_mat_adapter_compiles_and_the_paste_in_is_unambiguous(tmp_path):
    """include/sadbm.hpp's cv::Mat adapter + the INTEGRATION.md paste-in (SAD.cpp:13-61 replaced, main() kept) compile
    against tests/stubs/opencv2/core.hpp -- a stub, OpenCV is absent -- so a typo or an ambiguous overload is caught."""
    import subprocess
    src = tmp_path / "paste_in.cpp"
    src.write_text("""
#include <opencv2/core.hpp>
using namespace cv;
int winSize = 7;            // SAD.cpp:9
int DSR = 30;               // SAD.cpp:11
#include "sadbm.hpp"
Mat getDisparity(Mat &left, Mat &right) { return sadbm::getDisparity(left, right, winSize, DSR); }
int main() { Mat left, right; Mat Disparity = getDisparity(left, right); return Disparity.rows; }   // SAD.cpp:76
""")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"),
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
