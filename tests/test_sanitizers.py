"""CPU-side code under AddressSanitizer + UBSan: the hand-written image-file parsers of
imageio.hpp (PNG / DEFLATE / BMP / PNM read untrusted files) on valid, truncated and corrupted inputs, and the CPU
restatements of oracle/ on ordinary and degenerate shapes.  `make asan` builds both; a sanitizer report aborts the
program (-fno-sanitize-recover), so "exit status 0 or 1" is the whole check."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONVERT = os.path.join(ROOT, "opencv_amd", "csrc", "sadbm_convert_asan")
SELFTEST = os.path.join(ROOT, "oracle", "oracle_asan_selftest")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1")
ENV.pop("LD_PRELOAD", None)          # ASan must come first in the library list; the tools under test link nothing else


@pytest.fixture(scope="module")
def asan_tools():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "opencv_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)


def convert(src, dst):
    r = subprocess.run([CONVERT, src, dst], capture_output=True, text=True, timeout=60, env=ENV)
    assert r.returncode in (0, 1), (src, r.returncode, r.stderr[-2000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (src, r.stderr[-2000:])
    return r.returncode


def test_oracles_under_asan_ubsan(asan_tools):
    r = subprocess.run([SELFTEST], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "oracle selftest: ok" in r.stdout


def test_host_row_copies_under_asan_ubsan(asan_tools):
    """hostcopy.hpp (the SSSE3 / AVX2 channel-0 gather and the copy pool of the host API) on exact-size heap buffers: odd
    widths, 3 / 4 channels, tails, pitched rows, pool-split copies, four calling threads."""
    exe = os.path.join(ROOT, "opencv_amd", "csrc", "hostcopy_asan_selftest")
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99")     # the pool's detached workers never exit
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "hostcopy selftest: ok" in r.stdout


def test_valid_files_convert_under_asan(asan_tools, tmp_path):
    for name in ("tsukuba_left.png", "tsukuba_left.ppm", "tsukuba_disp_w7_d30.pgm", "cat.bmp", "gaus_cat_low.png"):
        for ext in ("png", "bmp", "ppm"):
            assert convert(os.path.join(GOLDEN, name), str(tmp_path / ("o." + ext))) == 0, (name, ext)


@pytest.mark.parametrize("name", ["tsukuba_left.png", "cat.bmp", "tsukuba_left.ppm", "hybrid_dog_cat.png"])
def test_truncated_and_corrupted_files_are_rejected_cleanly(asan_tools, name, tmp_path):
    data = open(os.path.join(GOLDEN, name), "rb").read()
    rng = np.random.default_rng(len(data))
    out = str(tmp_path / "o.ppm")
    cuts = [0, 1, 2, 7, 8, 13, 14, 20, 33, 40, 53, 54, 55, 57, 100, 1000, len(data) // 2, len(data) - 1]
    rejected = 0
    for i, n in enumerate(cuts):
        p = tmp_path / ("cut%d" % i)
        p.write_bytes(data[:n])
        rejected += convert(str(p), out)
    assert rejected >= len(cuts) - 1                            # (a PNG cut inside its last chunk may still decode)
    for i in range(60):                                         # byte flips: headers first, then anywhere
        b = bytearray(data)
        for _ in range(1 + i % 3):
            pos = int(rng.integers(0, 64 if i < 30 else len(b)))
            b[pos] ^= int(rng.integers(1, 256))
        p = tmp_path / ("flip%d" % i)
        p.write_bytes(bytes(b))
        convert(str(p), out)                                    # either outcome is fine; no sanitizer report is the point


def test_hostile_headers(asan_tools, tmp_path):
    import struct
    import zlib
    out = str(tmp_path / "o.ppm")

    def bmp(w, h, bpp, off=54, ncol=0, body=b"\0" * 64, comp=0):
        hdr = struct.pack("<2sIHHI", b"BM", 54 + len(body), 0, 0, off) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, bpp, comp, len(body), 0, 0, ncol, 0)
        return hdr + body

    def png(w, h, depth, ctype, idat, interlace=0):
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")

    cases = {
        "bmp_huge": bmp(60000, 60000, 24), "bmp_neg": bmp(-5, 5, 24), "bmp_zero": bmp(0, 0, 24), "bmp_off": bmp(4, 4, 24, off=1 << 30),
        "bmp_pal": bmp(4, 4, 8, ncol=100000), "bmp_pal_idx": bmp(4, 4, 8, off=54 + 8, ncol=2, body=b"\1\2\3\0\4\5\6\0" + b"\xff" * 16),
        "bmp_topdown": bmp(4, -4, 32, body=b"\7" * 64), "bmp_rle": bmp(4, 4, 8, comp=1), "bmp_16": bmp(4, 4, 16),
        "png_huge": png(1 << 30, 1 << 30, 8, 2, zlib.compress(b"\0" * 10)), "png_zero": png(0, 0, 8, 0, zlib.compress(b"")),
        "png_short": png(16, 16, 8, 2, zlib.compress(b"\0" * 20)), "png_badfilter": png(2, 2, 8, 0, zlib.compress(b"\x09ab\x09cd")),
        "png_interlaced": png(4, 4, 8, 0, zlib.compress(b"\0" * 40), interlace=1), "png_depth": png(4, 4, 3, 0, zlib.compress(b"\0" * 40)),
        "png_garbage_zlib": png(4, 4, 8, 0, b"\x78\x9c" + bytes(range(200))), "png_pal_no_plte": png(4, 4, 8, 3, zlib.compress(b"\0" * 20)),
        "pnm_huge": b"P6\n99999 99999\n255\nabc", "pnm_neg": b"P5\n-3 4\n255\n" + b"x" * 20, "pnm_trunc": b"P5\n", "pnm_max": b"P5\n2 2\n65535\nabcdefgh",
    }
    ok = {}
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        ok[name] = convert(str(p), out)
    assert ok["bmp_topdown"] == 0 and ok["bmp_pal_idx"] == 0
    for name in ("bmp_huge", "bmp_neg", "bmp_zero", "bmp_off", "bmp_pal", "bmp_rle", "bmp_16", "png_huge", "png_zero", "png_short",
                 "png_badfilter", "png_interlaced", "png_depth", "png_garbage_zlib", "png_pal_no_plte", "pnm_huge", "pnm_neg", "pnm_trunc", "pnm_max"):
        assert ok[name] == 1, name
