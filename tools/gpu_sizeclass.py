"""Where a solve's time goes by size of the moved subtree (MCF_NS_DEBUG histogram + the candidate cache's own counters): config 3 by default,
`config5` as first argument for BASELINE config 5 (optionally a pivot limit as second argument)."""
import os
import sys
sys.path.insert(0, ".")
os.environ["MCF_NS_DEBUG"] = "1"
os.environ["MCF_HIP_CAND_DEBUG"] = "1"
import mincostflow_amd as M
which = sys.argv[1] if len(sys.argv) > 1 else "config3"
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000) if which == "config5" else M.netgen_like(13502460, 100_000, 300_000, 316, 316)
for rep in range(1 if which == "config5" else 2):
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    if limit:
        ns.set_pivot_limit(limit)
    ns.prepare()
    st = ns.solve()
    m = ns.get_metrics(); e = m["engine"]; it = m["iterations"]
    print(f"status {st} pivots {it}  loop {m['loop_us']/1e3:.1f} ms = {m['loop_us']/it:.2f} us/pivot: search {m['pivot_search_us']/it:.2f}, tree {m['tree_update_us']/it:.2f}, "
          f"potential {m['potential_update_us']/it:.2f}; avg subtree {m['potential_nodes']/it:.0f}; device requests {e['resident_requests']} (in-kernel {e['resident_scan_ns']/max(1,e['resident_requests'])/1e3:.2f} us each), "
          f"phases shift/values/scan {e['phase_shift_ns']/max(1,e['resident_requests'])/1e3:.2f}/{e['phase_values_ns']/max(1,e['resident_requests'])/1e3:.2f}/{e['phase_scan_ns']/max(1,e['resident_requests'])/1e3:.2f} us, shift lists {e['shift_lists']}, "
          f"host-decided {e['host_decided']}, resident launches {e['resident_launches']}, update launches {e['update_launches']}, rc recomputes {e['rc_recomputes']}", flush=True)
    print("reduced costs on the device that differ from cost + pi[s] - pi[t]:", ns.check_reduced_costs(), flush=True)
    del ns
