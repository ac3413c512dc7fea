"""Workload for rocprofv3 --kernel-trace --stats: config 5's first pivots with one dispatch per search over the RC layout (scan_rc_kernel per pivot,
update_rc_kernel for the long lists), then the scan micro-benchmark's warm and cold repetitions on the same arrays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, M.ENGINE_DISPATCH)
ns.set_pivot_limit(20000).prepare()
ns.solve()
m = ns.get_metrics()
print("pivots", m["iterations"], "us/pivot", m["loop_us"] / m["iterations"], "scan launches", m["engine"]["scan_launches"], "bytes read per scan", m["engine"]["scan_bytes_read"])
it = ns.internal(); ms = it["search_arc_num"]
eng = M.PivotEngine(g5.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, flags=M.ENGINE_DISPATCH)
eng.upload(it["source"][:ms], it["target"][:ms], it["cost"][:ms], it["state"][:ms], it["pi"])
print("bench_scan warm", eng.bench_scan(reps=50), "cold", eng.bench_scan(reps=20, cold=True, flush_bytes=512 << 20))
