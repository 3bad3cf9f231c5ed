"""CPU tests of the launch policy (sadbm_plan_describe: what run_pipeline decides before it touches the GPU): the kernels the
BASELINE configurations get, and the invariants of frame groups, bands and grids over a few thousand parameter sets."""
import numpy as np
import pytest

import opencv_amd as oa


def test_baseline_configurations_get_the_documented_kernels():
    p = oa.plan(288, 384, 3, 1, 7, 30)                      # the reference's own call: BGR pair, winSize 7, DSR 30 (SAD.cpp:9,11,65-66)
    assert p["kernel"] == "sad_strip_otf_kernel<7,2>+direct" and p["prep"] == 0 and p["scratch"] == 0 and p["threads"] == 128   # round 3: one launch
    p = oa.plan(288, 384, 1, 256, 7, 30)                    # config 1 as a batch of grey frames: read in place, two frames per group
    assert p["kernel"] == "sad_strip_otf_kernel<7,2>+direct" and p["scratch"] == 0 and p["FG"] == 2 and p["bx"] == 3
    p = oa.plan(1080, 1920, 1, 64, 9, 64)                   # config 2
    assert p["kernel"] == "sad_strip_otf_kernel<9,4>+direct" and p["prep"] == 0 and p["FG"] == 2 and p["bx"] == 15 and p["tail"] > 0
    p = oa.plan(2160, 3840, 1, 32, 15, 128)                 # config 3
    assert p["kernel"] == "sad_strip_kernel<15>+direct" and p["threads"] == 512 and p["FG"] == 1 and p["bx"] == 15 and p["prep"] == 0 and p["scratch"] == 0
    p = oa.plan(4320, 7680, 1, 4, 21, 256)                  # config 4
    assert p["kernel"] == "sad_strip_kernel<21>+direct" and p["threads"] == 1024 and p["bx"] == 30 and p["prep"] == 0
    p = oa.plan(375, 1242, 1, 512, 7, 128)                  # config 5: 8 frames side by side fill 39 blocks of 256 columns
    assert p["kernel"] == "sad_strip_kernel<7>+direct" and p["FG"] == 8 and p["Wp"] == 1248 and p["bx"] == 39
    p = oa.plan(2160, 3840, 3, 32, 15, 128)                 # BGR frames keep the prep pass (channel 0 -> padded planes)
    assert p["kernel"] == "sad_strip_kernel<15>" and p["prep"] == 1 and p["scratch"] > 0
    assert oa.plan(200, 300, 1, 8, 15, 128)["kernel"] == "sad_strip_kernel<15>"      # rows shorter than 384 bytes: planes (more than two edge rows possible)
    assert oa.plan(1080, 1920, 1, 1, 3, 30)["matcher"] == "quad"
    assert oa.plan(100, 100, 1, 1, 31, 30)["kernel"] == "sad_quad_lds_kernel<31,8>"       # 26 ... 49: the quad matcher (no strip instantiation)
    assert oa.plan(100, 100, 1, 1, 50, 30)["matcher"] == "wide"
    assert oa.plan(300, 300, 1, 1, 255, 30)["kernel"] == "sad_wide_kernel"
    assert oa.plan(100, 100, 1, 1, 50, 30, oa.FORCE_GENERIC)["matcher"] == "generic"
    assert oa.plan(7, 100, 1, 1, 7, 30)["matcher"] == "none"        # rows <= winSize: all-zero result, nothing to launch
    assert oa.plan(1080, 1920, 1, 1, 7, 30, oa.KERNEL_QUAD)["kernel"] == "sad_quad_lds_kernel<7,8>"
    assert oa.plan(1080, 1920, 1, 32, 7, 100)["kernel"] == "sad_strip_otf_kernel<7,8>+direct"     # D % 16 != 0 up to 128: poisoned sums
    assert oa.plan(1080, 1920, 1, 32, 7, 112)["kernel"] == "sad_strip_kernel<7>+direct"


def test_plan_rejects_what_the_calls_reject():
    for args, code in (((10, 10, 2, 1, 7, 30), -2), ((10, 10, 1, 1, 0, 30), -4), ((10, 10, 1, 1, 7, 0), -4), ((10, 10, 1, 1, 7, 257, oa.RAW_LOC), -4),
                       ((5000, 5000, 1, 1, 4097, 30), -4), ((10, 10, 1, 0, 7, 30), -3)):
        with pytest.raises(oa.SadbmError) as e:
            oa.plan(*args)
        assert e.value.code == code, args


def test_plan_beyond_256_disparities_and_255_columns():
    """The reference limits neither winSize nor DSR (SAD.cpp:9,11): beyond 256 disparities the column-sum matcher runs (any window up to
    255 columns), beyond 255 columns the brute force; a search wider than the frame is the search over [0, cols)."""
    assert oa.plan(400, 700, 1, 1, 7, 300)["matcher"] == "wide"
    assert oa.plan(400, 700, 1, 1, 21, 5000)["matcher"] == "wide"
    assert oa.plan(400, 200, 1, 1, 7, 300)["kernel"] == oa.plan(400, 200, 1, 1, 7, 200)["kernel"]      # D clamps to cols
    assert oa.plan(400, 700, 1, 1, 300, 30)["matcher"] == "generic"
    assert oa.plan(100, 100, 1, 1, 5000, 30)["matcher"] == "none"


def test_strip_plans_cover_the_job():
    rng = np.random.default_rng(4)
    n_strip = 0
    for _ in range(3000):
        win = int(rng.integers(4, 26))
        D = int(rng.choice([1, 8, 16, 17, 30, 32, 33, 64, 65, 100, 128, 129, 200, 256]))
        rows = int(rng.choice([win + 1, int(rng.integers(win + 1, 400)), 480, 1080, 2160, 4320]))
        cols = int(rng.choice([win + 1, int(rng.integers(win + 1, 700)), 384, 640, 1242, 1920, 2048, 3840, 7680]))
        n = int(rng.choice([1, 1, 2, 3, 7, 32, 64, 256, 1000]))
        ch = int(rng.choice([1, 1, 3, 4]))
        p = oa.plan(rows, cols, ch, n, win, D)
        if not p["matcher"].startswith("strip"):
            continue
        n_strip += 1
        nvalid = rows - win
        FG, Wp, groups, bx, TY, nby = p["FG"], p["Wp"], p["groups"], p["bx"], p["TY"], p["nby"]
        what = (rows, cols, ch, n, win, D, p)
        assert FG >= 1 and groups * FG >= n > (groups - 1) * FG, what                   # every frame in exactly one group
        assert Wp >= cols and (FG == 1 or Wp % 16 == 0), what                            # strips never straddle two frames
        assert bx * 256 >= (FG - 1) * Wp + cols > (bx - 1) * 256, what                   # column blocks cover the group, none is empty
        assert (FG - 1) * Wp + cols < 65536, what                                        # 16-bit column fields
        assert TY >= 1 and nby * TY >= nvalid > (nby - 1) * TY, what                     # bands cover the computed rows, none is empty
        assert p["blocks"] == bx * nby * groups, what
        assert 0 <= p["tail"] <= (p["blocks"] + 7) // 8 // 2, what
        assert p["lds"] <= 160 * 1024 and p["threads"] in (64, 128, 256, 512, 1024), what
        Deff = min(D, cols)                                                              # disparities >= cols never compete (SAD.cpp:44)
        assert 16 * (p["threads"] // 64) >= Deff, what                                   # the block's disparity slots cover the search
        if p["matcher"] == "strip_otf":
            assert Deff <= 128 and p["lds"] <= 16 * 1024, what
            assert (p["prep"] == 0) == p["kernel"].endswith("+direct") and (p["prep"] == 1 or p["scratch"] == 0), what
            assert p["prep"] == 1 or ch == 1 or (ch == 3 and p["threads"] == 128 and win <= 8), what          # BGR in place: where that variant is built
        elif p["kernel"].endswith("+direct"):
            assert ch == 1 and p["prep"] == 0 and p["scratch"] == 0 and cols >= 384 and p["threads"] >= 512 and p["lds"] <= 80 * 1024 * (2 if p["threads"] == 1024 else 1), what
        else:
            assert p["prep"] == 1 and p["scratch"] > 2 * (rows - 1) * cols * groups, what
    assert n_strip > 2000


def test_bench_traffic_entries_describe_the_kernels_that_run():
    """bench.py takes roofline.traffic from bench_traffic.json, keyed by (workload, kernel, pairs per launch).  Every workload's CURRENT
    kernel (what the launch policy picks today) must have an entry -- a renamed or re-routed kernel fails here, on CPU, not in the
    round-end bench -- and every entry from round 2 on must still be what its profiles/ file says: the file exists, holds a PMC
    section of that kernel, and 2 x FETCH_SIZE + WRITE_SIZE of that section (KB; gfx950 tallies 16-byte reads at half size) is the entry."""
    import json
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, "bench_traffic.json")) as f:
        tj = json.load(f)
    for name, (W, H, win, D, _seed, _bg, _fg, pairs) in bench.WORKLOADS.items():
        if name.startswith("wide_"):
            continue                                           # profiling-only workload (SADBM_BENCH_ALLOW_NO_TRAFFIC)
        key = "%s|%s|pairs=%d" % (name, oa.plan(H, W, 1, pairs, win, D)["kernel"], pairs)
        assert key in tj, key
    if not os.path.isdir(os.path.join(root, "profiles")):
        pytest.skip("profiles/ does not travel to the GPU box (.gpurunignore)")
    checked = 0
    for key, ent in tj.items():
        m = re.match(r"(profiles/r0[2-9]\S+\.txt)", ent["source"])
        if not m:
            continue                                           # round-1 entries: files since superseded, kernels no longer the default
        path = os.path.join(root, m.group(1))
        assert os.path.exists(path), (key, path)
        kernel = key.split("|")[1].replace("+direct", "")
        base = kernel.split("<")[0] + "<" + kernel.split("<")[1].rstrip(">").replace(",", ", ")    # rocprofv3 prints "<7, 2, true>", "<15, 24>"
        lines = open(path).read().splitlines()
        starts = [i for i, l in enumerate(lines) if l.startswith(base) and "calls=" not in l]
        assert starts, (key, base, path)
        sect = lines[starts[0] + 1:]
        fetch = next(float(re.search(r"mean=([0-9.e+]+)", l).group(1)) for l in sect if l.strip().startswith("FETCH_SIZE"))
        write = next(float(re.search(r"mean=([0-9.e+]+)", l).group(1)) for l in sect if l.strip().startswith("WRITE_SIZE"))
        assert abs((2 * fetch + write) * 1024 - ent["hbm_bytes_per_launch"]) <= 0.002 * ent["hbm_bytes_per_launch"], (key, fetch, write)
        checked += 1
    assert checked >= 9
