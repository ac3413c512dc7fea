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


def test_committed_bench_line_keeps_the_contract():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r01_bench_n1_final.json")).read().strip().splitlines()[-1])
    _check(line, with_cpu=True)
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-microbench",
                          "--no-validator", "--concurrent", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    _check(json.loads(lines[0]), with_cpu=False)
