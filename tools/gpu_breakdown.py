"""Per-pivot host-side breakdown of one config-3 solve (resident mode)."""
import sys
sys.path.insert(0, ".")
import mincostflow_amd as M
g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
for rep in range(2):
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0).prepare()
    assert ns.solve() == 1
    m = ns.get_metrics(); e = m["engine"]; it = m["iterations"]
    print(f"pivots {it}  loop {m['loop_us']/it:.2f} us/pivot: search {m['pivot_search_us']/it:.2f} (post {e['host_launch_ns']/it/1e3:.2f} + wait {e['host_wait_ns']/it/1e3:.2f}; "
          f"device in-kernel {e['resident_scan_ns']/max(1,e['resident_requests'])/1e3:.2f}), tree {m['tree_update_us']/it:.2f}, potential {m['potential_update_us']/it:.2f}, "
          f"avg subtree {m['potential_nodes']/it:.0f}", flush=True)
