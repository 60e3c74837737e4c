"""bench.py as the driver starts it: `python bench.py --gpus N ...` with NO launcher around it.  For N > 1 the
(GPU-untouched) parent starts its own ranks; on the one-GPU box both ranks share cuda:0 and talk over gloo -- RCCL
refuses two ranks on one device -- so what is covered is the launch path, the per-round asynchronous gather and the
JSON contract, not xGMI."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--nx", "320", "--ny", "240", "--no-cpu", "--no-4k", "--no-sor", "--no-cli", "--no-occ", "--fixed-steps", "1", "--warmup", "1"]


def run_bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_one_gpu_line_has_roofline_keys():
    d = run_bench(["--gpus", "1", "--steps", "8"] + SMALL)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    for k in ("fused_algorithmic_bytes_per_launch", "algorithmic_equivalent_bytes_per_launch", "avg_launch_us", "pairs_per_launch"):
        assert k in r, k
    # quoted on the launch as the job issues it (a lockstep group), with the one-pair launch beside it; bytes scale with the group
    assert r["pairs_per_launch"] > 1 and 0 < r["single_pair"]["frac"] <= 1.0
    assert abs(r["achieved"] * r["avg_launch_us"] * 1e3 - r["fused_algorithmic_bytes_per_launch"]) < 1e-3 * r["fused_algorithmic_bytes_per_launch"]
    # the unit size is the one the library ran (stats.fused); the SURVEY 8(d) figure scales with it; a three-iteration launch
    # carries the two-iteration launch of the same shape beside it
    F = r["iterations_per_launch"]
    assert F in (2, 3) and ("k_tvl1_iter%d" % F) in r["kernel"]
    assert abs(r["algorithmic_equivalent_bytes_per_launch"] - F * r["fused_algorithmic_bytes_per_launch"]) < 1.0
    if F == 3:
        assert 0 < r["two_iterations_per_launch"]["frac"] <= 1.0
    assert d["config"]["arithmetic_mode"] == "tolerance" and d["strict"]["value"] > 0
    assert set(d["loop_ends"]["iterations_per_launch_by_level"]) <= {1, 2, 3}


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks_and_gathers_identical_bytes():
    d = run_bench(["--gpus", "2", "--backend", "gloo", "--steps", "4", "--rounds", "2", "--check"] + SMALL)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["config"]["rounds_per_gpu"] == [3, 1]
    assert d["gather_check"].startswith("ok: 8 payloads")
    assert d["gather_ms"] >= 0.0 and d["value"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_4k_batch_workload_shards_pairs_over_ranks():
    """BASELINE config 5 in miniature: ONE batch of 5 distinct pairs, pair k on rank k mod 2 (3 + 2), strong scaling"""
    d = run_bench(["--gpus", "2", "--backend", "gloo", "--workload", "4k-batch", "--steps", "5", "--rounds", "2", "--check"] + SMALL)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "strong"
    assert d["config"]["workload_name"] == "4k-batch" and d["config"]["pairs_per_gpu"] == 3
    assert d["gather_check"].startswith("ok: 5 payloads")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_4k_batch_workload_on_one_gpu_at_full_size():
    """`bench.py --workload 4k-batch --steps 4 --gpus 1`: BASELINE config 5's size (3840x2160) on the one GPU, the launch
    shape roofline_4k quotes (lockstep groups, non-temporal stores); the line must carry the contract keys and a roofline."""
    d = run_bench(["--gpus", "1", "--workload", "4k-batch", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-sor", "--no-cli", "--no-occ",
                   "--fixed-steps", "1"])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["workload_name"] == "4k-batch" and "3840x2160" in d["config"]["workload"]
    assert 0 < d["roofline"]["frac"] <= 1.0 and "3840x2160" in d["roofline"]["kernel"]
