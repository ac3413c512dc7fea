"""Config 5 with growing pivot limits: per-pivot cost and the share of long potential lists as the solve goes on (resident RC grid vs one dispatch per search)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
flags = M.ENGINE_DISPATCH if (len(sys.argv) > 1 and sys.argv[1] == "dispatch") else 0
for pivots in (100_000, 400_000, 1_000_000, 2_000_000, 4_000_000):
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, flags)
    ns.set_pivot_limit(pivots).prepare()
    t0 = time.time()
    st = ns.solve()
    m = ns.get_metrics(); n = max(m["iterations"], 1); e = m["engine"]
    print(f"limit {pivots}: status {st} {n} pivots in {time.time()-t0:.1f} s, {m['loop_us']/n:.1f} us/pivot | search {m['pivot_search_us']/n:.1f} pot {m['potential_update_us']/n:.1f} | "
          f"avg subtree {m['potential_nodes']/n:.0f} | resident launches {e['resident_launches']} requests {e['resident_requests']} updates {e['update_launches']} scans {e['scan_launches']}", flush=True)
    del ns
    if st != 0:
        break
