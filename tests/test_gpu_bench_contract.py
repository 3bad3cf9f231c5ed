"""bench.py's contract on the GPU box: one JSON line with the fields the driver reads, `--gpus N` honoured (ranks are
spawned when no launcher did; a mismatching WORLD_SIZE is refused), every pair of the last step verified bit-exact."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=170):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_single_gpu_line_has_the_contract_fields_and_verifies():
    r, lines = run_bench(["--workload", "cfg1_tsukuba_size_w7_d30", "--pairs", "4", "--steps", "3", "--warmup", "1", "--no-extras"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["verified_bit_exact"] is True
    rf = d["roofline"]
    assert rf["bound"] == "valu" and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


def test_gpus_2_spawns_two_ranks_and_refuses_a_mismatching_world():
    r, lines = run_bench(["--gpus", "2", "--workload", "cfg1_tsukuba_size_w7_d30", "--pairs", "4", "--steps", "3", "--warmup", "1", "--no-cpu"],
                         {"SADBM_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # an N > 1 line verifies itself (rank 0's pairs of the last step against the oracle) and carries the per-rank times
    assert d["verified_bit_exact"] is True, d.get("verified")
    assert d["ranks"]["ms_per_step_min"] <= d["ranks"]["ms_per_step_max"] and abs(d["ranks"]["ms_per_step_max"] - d["ms_per_step"]) < 1e-3
    assert "cpu_baseline" not in d                                    # N = 1 only
    r, lines = run_bench(["--gpus", "2", "--no-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines


@pytest.mark.parametrize("sub_bands", ["1", "3"])
def test_gpus_2_row_bands_line_is_verified_against_the_whole_frame(sub_bands):
    """--split bands at N = 2 (two gloo ranks sharing the box's GPU): the gathered map of the last step equals oracle.box of the
    whole frame -- the single-call result of SAD.cpp:34-59 -- and the line splits the step into band compute and gather."""
    r, lines = run_bench(["--gpus", "2", "--split", "bands", "--sub-bands", sub_bands, "--workload", "cfg2_1080p_w9_d64", "--steps", "3", "--warmup", "1",
                          "--no-cpu"], {"SADBM_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "row-bands2+allgather"
    assert d["verified_bit_exact"] is True, d.get("verified")
    assert d["ranks"]["band_compute_ms_max"] > 0 and d["ranks"]["gather_ms_max"] > 0
