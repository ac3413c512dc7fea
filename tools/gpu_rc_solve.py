"""Config 5 on one GPU: the first pivots with the gathering scan and with the RC layout (same pivots), then a longer stretch of the RC solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mincostflow_amd as M

g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)

def solve(label, env, pivots):
    for k, v in env.items():
        os.environ[k] = str(v)
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    ns.set_pivot_limit(pivots).record_trace(min(pivots, 4000)).prepare()
    ns.solve()
    m = ns.get_metrics(); n = max(m["iterations"], 1); e = m["engine"]
    print(f"{label} {pivots}: {m['loop_us']/n:.1f} us/pivot | search {m['pivot_search_us']/n:.1f} pot {m['potential_update_us']/n:.1f} tree {m['tree_update_us']/n:.2f} | "
          f"avg subtree {m['potential_nodes']/n:.0f} | scans {e['scan_launches']} updates {e['update_launches']} inline {e['inline_updates']}", flush=True)
    for k in env:
        os.environ.pop(k, None)
    return ns.trace()

a = solve("gather", {"MCF_HIP_RC": 0}, 4000)
b = solve("rc    ", {}, 4000)
assert np.array_equal(a, b)
solve("rc    ", {}, int(sys.argv[1]) if len(sys.argv) > 1 else 300000)
