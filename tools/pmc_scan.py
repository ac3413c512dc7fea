"""Few-dispatch workload for rocprofv3 --pmc: scan-only dispatches at several sizes, then a bounded resident solve."""
import sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M

what = sys.argv[1] if len(sys.argv) > 1 else "scan"
rng = np.random.default_rng(7)
if what == "scan":
    g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True)
    assert ns.begin() == 0
    it = ns.internal(); ms = it["search_arc_num"]
    eng = M.PivotEngine(g.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, flags=M.ENGINE_DISPATCH)
    eng.upload(it["source"][:ms], it["target"][:ms], it["cost"][:ms], it["state"][:ms], it["pi"])
    print("config3 warm", eng.bench_scan(reps=20), eng.stats()["bytes_per_scan"])
    print("config3 cold", eng.bench_scan(reps=10, cold=True, flush_bytes=512 << 20))
    del eng
    for m_s, n in ((64_000_000, 2_001), (64_000_000, 1_000_001)):
        a = dict(src=rng.integers(0, n, m_s, dtype=np.int32), tgt=rng.integers(0, n, m_s, dtype=np.int32),
                 cost=rng.integers(-10 ** 4, 10 ** 4, m_s, dtype=np.int64), state=rng.integers(-1, 2, m_s, dtype=np.int8),
                 pi=rng.integers(-10 ** 9, 1, n, dtype=np.int64))
        eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, flags=M.ENGINE_DISPATCH)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        print(m_s, n, "cold", eng.bench_scan(reps=5, cold=True, flush_bytes=512 << 20), eng.stats()["bytes_per_scan"])
        del eng, a
else:
    g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_pivot_limit(20000)
    ns.solve()
    m = ns.get_metrics()
    print("resident", m["iterations"], m["engine"]["resident_requests"], m["engine"]["resident_launches"], m["engine"]["bytes_per_scan"])
