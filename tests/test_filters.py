"""Gaussian low/high-pass path (gaus_filter.cpp, hybrid_image.cpp).  Oracle = oracle/filter_oracle.c
(parity with OpenCV itself is unpinned, see its header).  CPU: the oracle against an independent numpy formulation and
the product's host-side kernel generator; GPU: HIP kernels bit-exact against the oracle."""
import numpy as np
import pytest

import opencv_amd as oa
from oracle import filters as F


def test_gaussian_kernel_quirk_and_host_generator():
    for size, sigma in ((35, 3), (31, 10), (5, 1), (1, 1)):
        k = F.gaussian_kernel(size, sigma)
        assert np.array_equal(k, F.gaussian_kernel_np(size, sigma))
        assert np.array_equal(k, oa.filters.GetgaussianKernel(size, sigma))           # product host code == oracle
        assert abs(k.sum() - 1) < 1e-12 and k[size // 2, size // 2] == k.max()
    # the integer division makes plateaus: with sigma=10 every tap with r^2 < 200 has the centre weight (hybrid_image.cpp:26)
    k = F.gaussian_kernel(31, 10)
    assert k[15, 15] == k[15, 28] == k[6, 6] and k[5, 5] < k[15, 15] and k[15, 30] < k[15, 15]


def test_filter_oracle_matches_numpy_formulation():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (47, 61, 3), dtype=np.uint8)
    for k in (F.gaussian_kernel(35, 3), F.gaussian_kernel(31, 10), rng.standard_normal((3, 5))):
        assert np.array_equal(F.filter2d(img, k), F.filter2d_np(img, k))
    grey = img[:, :, 0].copy()
    assert np.array_equal(F.filter2d(grey, F.gaussian_kernel(7, 2)), F.filter2d_np(grey, F.gaussian_kernel(7, 2)))
    one = np.zeros((1, 1))
    one[0, 0] = 1.0
    assert np.array_equal(F.filter2d(img, one), img)                                  # identity kernel


def test_highpass_and_blend_definitions():
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    hp = F.highpass(a, b, 12)
    exp = np.clip(np.minimum(a.astype(int) + 12, 255) - b, 0, 255).astype(np.uint8)
    assert np.array_equal(hp, exp)
    bl = F.blend(a, b, 0.6)
    assert np.array_equal(bl[0], a[0]) and np.array_equal(bl[:, 0], a[:, 0]) and np.array_equal(bl[-1], a[-1])
    inner = (0.6 * a[1:-1, 1:-1].astype(np.float64) + (1.0 - 0.6) * b[1:-1, 1:-1]).astype(int)
    assert np.array_equal(bl[1:-1, 1:-1], inner.astype(np.uint8))


def test_filter_oracle_reproduces_the_committed_outputs_on_the_reference_pictures():
    """tests/golden/{gaus,hybrid}_*.png were produced by tests/golden/make_filter_golden.py from cat.bmp / dog.bmp."""
    import json
    import os
    from helpers import GOLDEN, sha
    Image = pytest.importorskip("PIL.Image")
    meta = json.load(open(os.path.join(GOLDEN, "filters_golden.json")))
    cat = np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, "cat.bmp")).convert("RGB"))[:, :, ::-1])
    dog = np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, "dog.bmp")).convert("RGB"))[:, :, ::-1])
    assert sha(cat) == meta["cat_bgr_sha256"] and sha(dog) == meta["dog_bgr_sha256"]
    k = F.gaussian_kernel(31, 10)
    low = F.filter2d(dog, k)
    high = F.highpass(cat, F.filter2d(cat, k), 12)
    assert sha(low) == meta["hybrid_dog_low_sha256"] and sha(high) == meta["hybrid_cat_high_sha256"]
    mix = F.blend(low, high, 0.6)
    assert sha(mix) == meta["hybrid_dog_cat_sha256"]
    stored = np.asarray(Image.open(os.path.join(GOLDEN, "hybrid_dog_cat.png")).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(stored, mix)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(361, 410, 3), (97, 131, 1), (40, 33, 4), (16, 16, 3), (5, 300, 3)])
def test_filter2d_hip_bit_exact(shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if shape[2] == 1:
        img = img[:, :, 0].copy()
    for k in (F.gaussian_kernel(35, 3), F.gaussian_kernel(31, 10), rng.standard_normal((3, 7)) * 0.3, np.full((1, 1), 2.5)):
        got = oa.filters.filter2D(_dev(img), k).cpu().numpy()
        assert np.array_equal(got, F.filter2d(img, k)), (shape, k.shape)


@pytest.mark.gpu
def test_gaus_filter_and_hybrid_programs_end_to_end():
    """gaus_filter.cpp:57-62 and hybrid_image.cpp:120-138 on random stand-ins for the 410x361 BGR test pictures."""
    rng = np.random.default_rng(11)
    dog = rng.integers(0, 256, (361, 410, 3), dtype=np.uint8)
    cat = rng.integers(0, 256, (361, 410, 3), dtype=np.uint8)
    # gaus_filter.cpp: 35x35, sigma 3, low = filter2D, high = src - low
    k = oa.filters.GetgaussianKernel(35, 3)
    low = oa.filters.filter2D(_dev(dog), k)
    high = oa.filters.highpass(_dev(dog), low)
    ref_low = F.filter2d(dog, F.gaussian_kernel(35, 3))
    assert np.array_equal(low.cpu().numpy(), ref_low)
    assert np.array_equal(high.cpu().numpy(), F.highpass(dog, ref_low, 0))
    # hybrid_image.cpp: 31x31, sigma 10; low(dog), high(cat + 12), blend alpha 0.6
    k = oa.filters.GetgaussianKernel(31, 10)
    dst_low = oa.filters.filter2D(_dev(dog), k)
    dst_tmp = oa.filters.filter2D(_dev(cat), k)
    dst_high = oa.filters.highpass(_dev(cat), dst_tmp, add=12)
    hybrid = oa.filters.blend(dst_low, dst_high, 0.6)
    r_low = F.filter2d(dog, F.gaussian_kernel(31, 10))
    r_high = F.highpass(cat, F.filter2d(cat, F.gaussian_kernel(31, 10)), 12)
    assert np.array_equal(hybrid.cpu().numpy(), F.blend(r_low, r_high, 0.6))
    # pitched views
    import torch
    big = torch.zeros((100, 140, 3), dtype=torch.uint8, device="cuda")
    big[10:90, 20:120] = _dev(dog[:80, :100])
    out = oa.filters.filter2D(big[10:90, 20:120], k)
    assert np.array_equal(out.cpu().numpy(), F.filter2d(dog[:80, :100].copy(), F.gaussian_kernel(31, 10)))


@pytest.mark.gpu
def test_host_level_filter_programs_match_the_oracle():
    """sadbm_gaus_filter_host / sadbm_hybrid_image_host (what sadbm_cli --gaus / --hybrid call) on 1-, 3- and 4-channel
    host images of odd sizes, against the oracle's definition of each step."""
    rng = np.random.default_rng(11)
    for shape in ((37, 53), (61, 47, 3), (40, 33, 4)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = rng.integers(0, 256, shape, dtype=np.uint8)
        low, high = oa.filters.gaus_filter_host(a)                       # 35 x 35, sigma 3 (gaus_filter.cpp:10-11)
        r_low = F.filter2d(a, F.gaussian_kernel(35, 3))
        assert np.array_equal(low, r_low), shape
        assert np.array_equal(high, F.highpass(a, r_low, 0)), shape
        if len(shape) == 3:
            l2, h2, mix = oa.filters.hybrid_image_host(a, b)             # 31 x 31, sigma 10, += 12, alpha 0.6
            k = F.gaussian_kernel(31, 10)
            r_l = F.filter2d(a, k)
            r_h = F.highpass(b, F.filter2d(b, k), 12)
            assert np.array_equal(l2, r_l) and np.array_equal(h2, r_h), shape
            assert np.array_equal(mix, F.blend(r_l, r_h, 0.6)), shape
    with pytest.raises(oa.SadbmError) as e:
        oa.filters.hybrid_image_host(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8))   # add() indexes three channels
    assert e.value.code == -2
