"""Headless CLI (replacement of main(), SAD.cpp:63-82) and its dependency-free PNG/PNM reader/writer.
CPU tests cover the file I/O (`--convert` touches no GPU); the GPU test runs config 1 end to end from the
reference's own PNG pair (tests/golden/tsukuba_*.png are byte copies of test_picture/left.png, right.png)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, read_pnm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "opencv_amd", "csrc", "sadbm_cli")


def run(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True, timeout=120)


def need_cli():
    if not os.path.exists(CLI):
        pytest.skip("sadbm_cli not built (make -C opencv_amd/csrc)")


@pytest.mark.parametrize("side", ["left", "right"])
def test_png_decoder_matches_committed_ppm(side, tmp_path):
    """PNG -> PPM equals the PPM fixture that tests/golden/make_golden.py decoded with PIL/libpng."""
    need_cli()
    out = str(tmp_path / "o.ppm")
    r = run("--convert", "%s/tsukuba_%s.png" % (GOLDEN, side), out)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open("%s/tsukuba_%s.ppm" % (GOLDEN, side), "rb").read()


def test_png_writer_round_trip_and_grey(tmp_path):
    need_cli()
    p1, p2, p3 = str(tmp_path / "a.png"), str(tmp_path / "b.ppm"), str(tmp_path / "c.png")
    assert run("--convert", GOLDEN + "/tsukuba_left.ppm", p1).returncode == 0
    assert run("--convert", p1, p2).returncode == 0
    assert open(p2, "rb").read() == open(GOLDEN + "/tsukuba_left.ppm", "rb").read()
    assert run("--convert", GOLDEN + "/tsukuba_disp_w7_d30.pgm", p3).returncode == 0
    g = str(tmp_path / "g.pgm")
    assert run("--convert", p3, g).returncode == 0
    assert np.array_equal(read_pnm(g), read_pnm(GOLDEN + "/tsukuba_disp_w7_d30.pgm"))


def test_png_variants_against_pil(tmp_path):
    """Colour types 0/2/3/4/6, 16-bit, sub-byte depths, every filter type (PIL picks them adaptively on noise + ramps)."""
    need_cli()
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    H, W = 37, 53
    ramp = (np.add.outer(np.arange(H), np.arange(W)) * 3 % 256).astype(np.uint8)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    rgb = np.stack([ramp, noise, ramp[::-1]], -1)
    cases = {
        "grey8": Image.fromarray(ramp, "L"),
        "rgb8": Image.fromarray(rgb, "RGB"),
        "rgba8": Image.fromarray(np.concatenate([rgb, noise[..., None]], -1), "RGBA"),
        "la8": Image.fromarray(np.stack([noise, ramp], -1), "LA"),
        "pal8": Image.fromarray(rgb, "RGB").quantize(200),
        "pal4": Image.fromarray(rgb, "RGB").quantize(13),
        "bw1": Image.fromarray(noise > 127),
        "grey16": Image.fromarray(noise.astype(np.uint16) << 8 | ramp),
    }
    for name, im in cases.items():
        src, dst = str(tmp_path / (name + ".png")), str(tmp_path / (name + ".pnm"))
        im.save(src, optimize=(name != "rgb8"))
        r = run("--convert", src, dst)
        assert r.returncode == 0, (name, r.stderr)
        got = read_pnm(dst)
        if name == "grey16":
            exp = noise                                             # high byte, as imread without IMREAD_ANYDEPTH
        elif name == "bw1":
            exp = (noise > 127).astype(np.uint8) * 255
        elif im.mode in ("L", "LA"):
            exp = np.asarray(im.convert("L")) if name != "la8" else noise
        else:
            exp = np.asarray(im.convert("RGB"))                     # read_pnm returns PPM as stored: RGB
        assert np.array_equal(got, exp), name


def test_bad_files_are_rejected(tmp_path):
    need_cli()
    bad = tmp_path / "bad.png"
    data = bytearray(open(GOLDEN + "/tsukuba_left.png", "rb").read())
    data[200] ^= 0x55                                               # breaks an IDAT CRC
    bad.write_bytes(bytes(data))
    r = run("--convert", str(bad), str(tmp_path / "x.ppm"))
    assert r.returncode == 1 and "CRC" in r.stderr
    r = run("--convert", str(tmp_path / "missing.png"), str(tmp_path / "x.ppm"))
    assert r.returncode == 1 and "cannot open" in r.stderr
    assert run("onlyone.png").returncode == 2


def test_bmp_reader_on_the_reference_pictures(tmp_path):
    """cat.bmp / dog.bmp (byte copies of the reference's test_picture/ files): the BMP reader's pixels equal the PIL decode
    recorded by tests/golden/make_filter_golden.py; BMP -> PNG -> BMP keeps every pixel."""
    need_cli()
    import hashlib
    import json
    meta = json.load(open(GOLDEN + "/filters_golden.json"))
    for n in ("cat", "dog"):
        ppm, png, bmp = str(tmp_path / (n + ".ppm")), str(tmp_path / (n + ".png")), str(tmp_path / (n + "2.bmp"))
        assert run("--convert", "%s/%s.bmp" % (GOLDEN, n), ppm).returncode == 0
        rgb = read_pnm(ppm)
        assert list(rgb.shape) == meta[n + "_shape"]
        assert hashlib.sha256(np.ascontiguousarray(rgb[:, :, ::-1]).tobytes()).hexdigest() == meta[n + "_bgr_sha256"]
        assert run("--convert", ppm, png).returncode == 0 and run("--convert", png, bmp).returncode == 0
        back = str(tmp_path / "back.ppm")
        assert run("--convert", bmp, back).returncode == 0
        assert np.array_equal(read_pnm(back), rgb)


@pytest.mark.gpu
def test_cli_filter_programs_on_the_reference_pictures(tmp_path):
    """`sadbm_cli --gaus` / `--hybrid` = main() of gaus_filter.cpp:42-70 / hybrid_image.cpp:102-143 on the reference's own
    pictures, with the programs' own parameters, against the oracle outputs committed under tests/golden/."""
    need_cli()
    import hashlib
    import json
    meta = json.load(open(GOLDEN + "/filters_golden.json"))

    def bgr_sha(path):
        ppm = str(tmp_path / "t.ppm")
        assert run("--convert", path, ppm).returncode == 0
        return hashlib.sha256(np.ascontiguousarray(read_pnm(ppm)[:, :, ::-1]).tobytes()).hexdigest()

    low, high = str(tmp_path / "low.bmp"), str(tmp_path / "high.png")
    r = run("--gaus", GOLDEN + "/cat.bmp", "-o", low, "--high", high)
    assert r.returncode == 0, r.stderr
    assert bgr_sha(low) == meta["gaus_cat_low_sha256"] == bgr_sha(GOLDEN + "/gaus_cat_low.png")
    assert bgr_sha(high) == meta["gaus_cat_high_sha256"]
    mix, low, high = str(tmp_path / "mix.bmp"), str(tmp_path / "l.bmp"), str(tmp_path / "h.bmp")
    r = run("--hybrid", GOLDEN + "/dog.bmp", GOLDEN + "/cat.bmp", "-o", mix, "--low", low, "--high", high)
    assert r.returncode == 0, r.stderr
    assert bgr_sha(low) == meta["hybrid_dog_low_sha256"]
    assert bgr_sha(high) == meta["hybrid_cat_high_sha256"]
    assert bgr_sha(mix) == meta["hybrid_dog_cat_sha256"] == bgr_sha(GOLDEN + "/hybrid_dog_cat.png")


@pytest.mark.gpu
def test_cli_config1_from_reference_pngs(tmp_path):
    """BASELINE config 1 end to end: the reference's PNG pair, defaults winSize=7 DSR=30, output == golden map."""
    need_cli()
    out = str(tmp_path / "d.pgm")
    r = run(GOLDEN + "/tsukuba_left.png", GOLDEN + "/tsukuba_right.png", "-o", out)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_pnm(out), read_pnm(GOLDEN + "/tsukuba_disp_w7_d30.pgm"))
    out2 = str(tmp_path / "d.png")
    r = run(GOLDEN + "/tsukuba_left.ppm", GOLDEN + "/tsukuba_right.ppm", "--raw", "-o", out2)
    assert r.returncode == 0, r.stderr
    back = str(tmp_path / "back.pgm")
    assert run("--convert", out2, back).returncode == 0
    assert np.array_equal(read_pnm(back), read_pnm(GOLDEN + "/tsukuba_loc_w7_d30.pgm"))
