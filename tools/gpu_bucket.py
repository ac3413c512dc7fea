"""Feasibility: scan time of config 5's arrays when the arcs are stably sorted by target-node range (timing only; tie-breaks ignored)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
assert ns5.begin() == 0
it = ns5.internal(); ms = it["search_arc_num"]; n = g5.node_count + 1
src, tgt, cost, state, pi = it["source"][:ms], it["target"][:ms], it["cost"][:ms], it["state"][:ms], it["pi"]
rng = np.random.default_rng(1)
pi = pi + rng.integers(-1000, 1000, n)       # not all equal
def run(label, order):
    eng = M.PivotEngine(n, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, flags=M.ENGINE_DISPATCH)
    eng.upload(src[order], tgt[order], cost[order], state[order], pi)
    nb = eng.stats()["bytes_per_scan"]
    w = eng.bench_scan(reps=20); c = eng.bench_scan(reps=8, cold=True, flush_bytes=512 << 20)
    print(f"{label}: warm {w[0]/1e3:.1f} us ({nb/w[0]:.0f} GB/s), cold {c[0]/1e3:.1f} us ({nb/c[0]:.0f} GB/s)", flush=True)
    del eng
run("generator order", np.arange(ms))
for R in (524288, 262144, 131072, 65536, 16384):
    run(f"bucketed by target range of {R} nodes", np.argsort(tgt // R, kind="stable"))
run("bucketed by source range of 131072 nodes AND target range", np.lexsort((tgt // 131072, src // 131072)))
run("random order", rng.permutation(ms))
