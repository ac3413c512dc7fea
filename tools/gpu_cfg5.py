"""First pivots of config 5 on one GPU: per-pivot time and in-kernel time of the resident tile loop."""
import sys
sys.path.insert(0, ".")
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
for flags, name in ((0, "resident"), (M.ENGINE_DISPATCH, "dispatch")):
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True)
    ns.set_device(0, 64, 0, flags).set_pivot_limit(3000).prepare()
    ns.solve()
    m = ns.get_metrics(); it = max(m["iterations"], 1); e = m["engine"]
    print(f"{name}: {m['loop_us']/it:.1f} us/pivot, search {m['pivot_search_us']/it:.1f}, in-kernel {e['resident_scan_ns']/max(1,e['resident_requests'])/1e3:.1f}, grid {e['scan_workgroups']}x{e['scan_threads']}", flush=True)
