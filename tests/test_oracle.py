"""CPU tests that pin the oracle (oracle/) -- the restatement of /root/reference/SAD.cpp:13-61.

The reference has no tests or golden outputs ("parity unpinned"), so the oracle is pinned by
(1) the hand-derivable known-answer tests K1-K9 of SURVEY.md section 4, (2) the SHA-256 vectors of
SURVEY.md section 4 T2 and the committed fixtures in tests/golden/, and (3) agreement of three
independent formulations (literal loops, sliding box sums, numpy integral image).
"""
import numpy as np
import pytest

import oracle
from helpers import GOLDEN, golden_meta, noise_pair, read_pnm, sha, tsukuba_bgr

# SURVEY.md section 4 T2 (restatement-derived vectors recorded by the survey session)
T2 = {
    "blueL": "aa2f80186b58933c38d191c0b08c2421e2ed08fa37a7aed0f9b5339c9298a1d2",
    "blueR": "512166acafe1fb20237a5c86e2e63143d949bc71eeaa22b9f26097e7a05ac0d4",
    "disp": "7da39f56141fbb2fc045c375b5be9fe84e8620beabec3009c99a5194075ea166",
    "loc": "06b755a9c88b300c1929157d3746cf7a78c025d53a4e8f69ba6afa7c2bd941b3",
    "hist": [1772, 337, 2825, 430, 12336, 29893, 12953, 1149, 13259, 614, 7894, 5537, 2938, 661, 6471, 521, 744, 332,
             744, 314, 587, 298, 490, 391, 584, 334, 468, 309, 425, 327],
    "rds_u8": ("6ba71a1f0e687185732fcca2aaa3a84aa7e8f3960f7d4fab5ffc333dd0328427",
               "4fe03e1baa4c24f44500b03030365c6e7c4b81b831f3de8b1227b3fa0849bd7b",
               "dc4dc5014857bce5c6de94478d86211225bb49eeb531a542c5634f25afd3d06b",
               "0dbaaf104191c84999eb576c77f26d6b44f6e0fd7eda90a92b8e5323b07cd575"),
    "rds_bin": ("22dd757e875429f7d66a29b46cd67486bae8dd2a4aff456779a9842dc668df64",
                "ec967bd777f89e764c680067c32e8378a177ac88553e479e66b6cd4190a8149f",
                "611e796db8857a439640a4215fd254d80e82ee3022ad5cc02d2f777a7ec52556",
                "85b1b4e62fe95c1daf24a64dbc306a320cb5e896e14269ea9215539e95d77924"),
}


def test_splitmix64_known_answers():
    assert [oracle.splitmix64(1, i) for i in (1, 2, 3)] == [0x910A2DEC89025CC1, 0xBEEB8DA1658EEC67, 0xF893A2EEFB32555E]


def test_tsukuba_golden_config1():
    """BASELINE config 1: test_picture/left.png + right.png, winSize=7, DSR=30, fed as BGR."""
    L, R = tsukuba_bgr()
    assert L.shape == (288, 384, 3)
    assert sha(L[:, :, 0]) == T2["blueL"] and sha(R[:, :, 0]) == T2["blueR"]
    disp = oracle.literal(L, R, 7, 30)
    loc = oracle.literal(L, R, 7, 30, oracle.RAW_LOC)
    assert sha(disp) == T2["disp"] and sha(loc) == T2["loc"]
    assert np.bincount(loc[:281, :377].ravel(), minlength=30).tolist() == T2["hist"]
    assert loc[0, :40].tolist() == list(range(1, 30)) + [5] * 11
    assert not disp[-7:, :].any() and not disp[:, -7:].any()
    assert np.array_equal(disp, read_pnm(GOLDEN + "/tsukuba_disp_w7_d30.pgm"))
    assert np.array_equal(loc, read_pnm(GOLDEN + "/tsukuba_loc_w7_d30.pgm"))
    assert np.array_equal(disp, (loc.astype(np.int32) * 16 & 255).astype(np.uint8))      # K6 wrap
    m = golden_meta()
    assert m["tsukuba_disp_sha256"] == T2["disp"]


@pytest.mark.parametrize("binary,key", [(0, "rds_u8"), (1, "rds_bin")])
def test_rds_golden(binary, key):
    L, R = oracle.rds(1, 320, 240, 4, 20, binary)
    hl, hr, hd, hraw = T2[key]
    assert sha(L) == hl and sha(R) == hr
    if not binary:
        assert L[0, :8].tolist() == [145, 190, 248, 113, 113, 195, 224, 133]
        assert R[0, :8].tolist() == [113, 195, 224, 133, 73, 203, 103, 154]
    disp = oracle.literal_mt(L, R, 9, 64)
    raw = oracle.literal_mt(L, R, 9, 64, oracle.RAW_LOC)
    assert sha(disp) == hd and sha(raw) == hraw
    assert raw[0, :8].tolist() == [1, 2, 3, 4, 4, 4, 4, 4]
    assert (raw[100, 60:70] == 4).all() and (raw[100, 100:110] == 20).all()
    tag = "bin" if binary else "u8"
    assert np.array_equal(disp, read_pnm(GOLDEN + "/rds_seed1_320x240_w9_d64_%s_disp.pgm" % tag))


def test_k1_identical():
    L, _ = noise_pair(3, 40, 50)
    loc = oracle.literal(L, L, 7, 30, oracle.RAW_LOC)
    assert not loc.any()


def test_k2_constant():
    L = np.full((30, 60), 10, np.uint8)
    R = np.full((30, 60), 200, np.uint8)
    loc = oracle.literal(L, R, 5, 16, oracle.RAW_LOC)
    exp = np.zeros_like(loc)
    for j in range(60 - 5):
        exp[:25, j] = j + 1 if j < 15 else 0
    assert np.array_equal(loc, exp)


def test_k4_pure_shift():
    d0 = 6
    L, R = noise_pair(5, 48, 96, shift=d0)
    loc = oracle.literal(L, R, 7, 20, oracle.RAW_LOC)
    for j in range(96 - 7 - d0):      # columns whose right window lies inside the shifted copy
        exp = j + 1 if j < d0 else d0
        assert (loc[:41, j] == exp).all(), j


def test_k5_k7_zero_frame_and_degenerate():
    L, R = noise_pair(7, 20, 24)
    out = oracle.literal(L, R, 5, 8)
    assert not out[15:, :].any() and not out[:, 19:].any()
    for (h, w, win) in ((5, 30, 5), (30, 5, 5), (4, 4, 7), (1, 1, 1)):
        a, b = noise_pair(1, h, w)
        assert not oracle.literal(a, b, win, 8).any()
        assert not oracle.box(a, b, win, 8).any()


def test_k8_three_channel_is_plane0():
    L, R = noise_pair(11, 30, 40, ch=3)
    a = oracle.literal(L, R, 7, 16)
    b = oracle.literal(np.ascontiguousarray(L[:, :, 0]), np.ascontiguousarray(R[:, :, 0]), 7, 16)
    assert np.array_equal(a, b)
    L4, R4 = noise_pair(12, 30, 40, ch=4)
    assert np.array_equal(oracle.literal(L4, R4, 3, 9), oracle.literal(L4[:, :, 0].copy(), R4[:, :, 0].copy(), 3, 9))


@pytest.mark.parametrize("win", [1, 3, 7, 9, 15, 21])
@pytest.mark.parametrize("D", [1, 16, 30, 64, 128, 256])
def test_three_formulations_agree(win, D):
    H, W = 40, 72
    for binary in (False, True):
        L, R = noise_pair(win * 1000 + D, H, W, shift=3, binary=binary)
        a = oracle.literal_mt(L, R, win, D, oracle.RAW_LOC)
        assert np.array_equal(a, oracle.box(L, R, win, D, oracle.RAW_LOC))
        assert np.array_equal(a, oracle.numpy_restatement(L, R, win, D, raw=True))
        assert np.array_equal(oracle.literal_mt(L, R, win, D), (a.astype(np.int32) * 16 & 255).astype(np.uint8))


def test_literal_rows_and_mt_match_single_thread():
    L, R = noise_pair(21, 50, 64)
    full = oracle.literal(L, R, 7, 30)
    assert np.array_equal(full, oracle.literal_mt(L, R, 7, 30, threads=3))
    parts = [oracle.literal_rows(L, R, 7, 30, r0, min(r0 + 13, 50)) for r0 in range(0, 50, 13)]
    assert np.array_equal(full, np.concatenate(parts, axis=0))


def test_strided_roi_views():
    big_l, big_r = noise_pair(31, 60, 100)
    L = big_l[5:45, 10:80]
    R = big_r[5:45, 10:80]
    a = oracle.literal(L, R, 7, 16)
    b = oracle.literal(np.ascontiguousarray(L), np.ascontiguousarray(R), 7, 16)
    assert np.array_equal(a, b)


def test_errors():
    L, R = noise_pair(1, 10, 10)
    with pytest.raises(ValueError):
        oracle.literal(L, R, 0, 8)
    with pytest.raises(ValueError):
        oracle.literal(L, R, 3, 0)


def test_product_side_generator_matches_oracle_generator():
    """opencv_amd.synth.rds (what bench.py feeds the GPU) == the oracle's rds, including the SURVEY check values."""
    from opencv_amd import synth
    for args in ((1, 320, 240, 4, 20, 0), (1, 320, 240, 4, 20, 1), (1000, 1242, 375, 5, 60, 0), (7, 33, 17, 40, 50, 0), (2, 1920, 1080, 8, 40, 0)):
        a, b = synth.rds(*args), oracle.rds(*args)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), args
    assert int(synth.splitmix64_stream(1, 3)[2]) == 0xf893a2eefb32555e
