"""The bench line the driver parses: checked on the committed round-1 run (profiles/r01_bench_n1_final.json), so a later edit of bench.py that
drops a contract key shows up on the CPU, and on the GPU by running bench.py itself with the smallest settings."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
       "roofline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _check(line, with_cpu):
    for k in TOP:
        assert k in line, k
    assert line["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert line["unit"] == "pivots/s" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and line["dtype"] in ("int64", "int32") and "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for k in ROOFLINE:
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(line["value"] - line["pivots_per_solve"] * line["steps"] * line["n_gpus"] / (line["ms_per_step"] * line["steps"] / 1e3)) / line["value"] < 0.02
    if with_cpu:
        for k in CPU:
            assert k in line["cpu_baseline"], k
        assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1


@pytest.mark.parametrize("name", ["r01_bench_n1_final.json", "r02_bench_n1.json", "r03_bench_n1.json"])
def test_committed_bench_line_keeps_the_contract(name):
    line = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    _check(line, with_cpu=True)
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    if not name.startswith("r01"):
        assert line["roofline"]["traffic_source"] and line["cpu_baseline"]["host_cpu"] and set(line["other_configs"]) == {"config2", "config4"}
        assert line["engine"]["candidate_cache"] is True and line["every_search_on_the_device"]["identical_pivot_sequence"] is True
    if name.startswith("r03"):
        # which machine's optimized Block Search the legs reproduce, the CPU leg on one named core, the update kernels priced, int32 scan rows
        assert "Vector<long>.Count = 4" in line["config"]["semantics"] and line["cpu_baseline"]["pinned_to_cpu"] >= 0 and line["potential_update_microbench"]
        assert any(r.get("dtype") == "i32" for r in line["scan_microbench"])
        assert line["sharded"]["identical_pivot_sequence"] is True and line["sharded"]["variants"]["rccl_all_gather"]["us_per_pivot"] < 100


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-microbench",
                          "--no-validator", "--concurrent", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    _check(json.loads(lines[0]), with_cpu=False)


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """`--gpus N` never labels a run with another number of ranks (no GPU needed: the check comes before anything touches one)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "refusing" in out.stderr and not out.stdout.strip()


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_spawns_its_ranks_and_runs_the_sharded_leg():
    """`python bench.py --gpus 2 --backend gloo` without a launcher starts two ranks itself (both on this box's one GPU), prints n_gpus 2 and
    times the arc-sharded config-5 leg with two shard engines exchanging their candidates through shared memory."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                          "--sharded-pivots", "300"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    _check(line, with_cpu=False)
    assert line["n_gpus"] == 2
    sh = line["sharded"]
    assert sh["ranks"] == 2 and sh["variants"]["host_exchange"]["ranks"] == 2 and sh["variants"]["host_exchange"]["pivots"] == 300
    assert sh["identical_pivot_sequence"] is True and sh["single_gpu_same_pivots"]["pivots"] == 300
