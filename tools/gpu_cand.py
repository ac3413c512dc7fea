"""Candidate-cache timing on config 3: solve with MCF_ENGINE_CANDIDATES under a few settings; MCF_HIP_CAND_DEBUG prints the host-side breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCF_HIP_CAND_DEBUG"] = "1"
import mincostflow_amd as M

g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
def run(flags, label, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    best = None
    for rep in range(2):
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, flags).prepare()
        assert ns.solve() == 1
        m = ns.get_metrics()
        best = m if best is None or m["loop_us"] < best["loop_us"] else best
        del ns
    it = best["iterations"]
    e = best["engine"]
    print(f"{label}: {best['loop_us']/1e3:.1f} ms, {it/(best['loop_us']/1e6)/1e3:.1f} k pivots/s, {best['loop_us']/it:.2f} us/pivot | search {best['pivot_search_us']/it:.2f} pot {best['potential_update_us']/it:.2f} "
          f"tree {best['tree_update_us']/it:.2f} | device requests {e['resident_requests']} host {e['host_decided']} async {e['async_refreshes']}", flush=True)
    for k in (env or {}):
        os.environ.pop(k, None)

run(0, "plain resident")
run(M.ENGINE_CANDIDATES, "candidates default")
for nodes in (16, 96, 200):
    run(M.ENGINE_CANDIDATES, f"candidates nodes={nodes}", {"MCF_HIP_CAND_NODES": nodes})
for low in (4, 32):
    run(M.ENGINE_CANDIDATES, f"candidates refresh_low={low}", {"MCF_HIP_CAND_REFRESH": low})
