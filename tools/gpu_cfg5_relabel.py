"""Config 5's whole solve with the node relabelling at the solver's own policy and forced to every x * n walked nodes (MCF_NS_RENUMBER=x)."""
import os, sys, time
sys.path.insert(0, ".")
import mincostflow_amd as M
g = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
for every in (sys.argv[1:] or [None, "512", "256"]):
    if every in (None, "policy"): os.environ.pop("MCF_NS_RENUMBER", None)
    else: os.environ["MCF_NS_RENUMBER"] = every
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0).prepare()
    t0 = time.perf_counter(); st = ns.solve(); dt = time.perf_counter() - t0
    m = ns.get_metrics(); it = m["iterations"]; e = m["engine"]
    print(f"MCF_NS_RENUMBER={every}: status {st} {it} pivots in {dt:.2f} s | per pivot us: search {m['pivot_search_us']/it:.2f} potential {m['potential_update_us']/it:.2f} tree {m['tree_update_us']/it:.2f} | "
          f"relabellings {e['renumberings']} launches {e['resident_launches']} reloads in grid {e['rc_reloads_in_grid']} | mismatching reduced costs {ns.check_reduced_costs()}", flush=True)
    del ns
