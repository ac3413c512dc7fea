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
elif what == "rc5":
    # config 5's arrays: the RC layout's scan (cold repetitions: each preceded by flush_kernel streaming 512 MiB = the calibration dispatch)
    # and, in the same run, the gathering scan over the bucketed layout
    import os
    g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
    ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
    assert ns5.begin() == 0
    it = ns5.internal(); ms = it["search_arc_num"]
    for env in ({}, {"MCF_HIP_RC": "0"}):
        os.environ.update(env)
        eng = M.PivotEngine(g5.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, flags=M.ENGINE_DISPATCH)
        for k in env:
            os.environ.pop(k)
        eng.upload(it["source"][:ms], it["target"][:ms], it["cost"][:ms], it["state"][:ms], it["pi"])
        st = eng.stats()
        print("rc" if st["rc_layout"] else "gather", "warm", eng.bench_scan(reps=10), "cold", eng.bench_scan(reps=6, cold=True, flush_bytes=512 << 20),
              st["scan_bytes_read"], st["bytes_per_scan"])
        del eng
    # a stretch of the solve: scans with their inline shifts, and the list updates
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_pivot_limit(3000)
    ns.solve()
    print("solve", ns.get_metrics()["iterations"])
else:
    g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_pivot_limit(20000)
    ns.solve()
    m = ns.get_metrics()
    print("resident", m["iterations"], m["engine"]["resident_requests"], m["engine"]["resident_launches"], m["engine"]["bytes_per_scan"])
