"""Scan-only timing of config 5's start-basis arrays and of a uniform random graph of the same size (dispatch mode)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
assert ns5.begin() == 0
it = ns5.internal(); ms = it["search_arc_num"]; n = g5.node_count + 1
rng = np.random.default_rng(1)
def run(label, src, tgt):
    eng = M.PivotEngine(n, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, flags=M.ENGINE_DISPATCH)
    eng.upload(src, tgt, it["cost"][:ms], it["state"][:ms], it["pi"] + rng.integers(-1000, 1000, n))
    nb = eng.stats()["bytes_per_scan"]
    w = eng.bench_scan(reps=20); c = eng.bench_scan(reps=8, cold=True, flush_bytes=512 << 20)
    print(f"{label}: warm {w[0]/1e3:.1f} us ({nb/w[0]:.0f} GB/s), cold {c[0]/1e3:.1f} us ({nb/c[0]:.0f} GB/s)", flush=True)
run("config 5 arrays, generator order", it["source"][:ms], it["target"][:ms])
run("uniform random end points", rng.integers(0, n, ms, dtype=np.int32), rng.integers(0, n, ms, dtype=np.int32))
