"""Config 5 on one GPU, RC layout from the resident grid: every search on the device against the candidate cache on top of it (same pivots),
over stretches of the solve and over the whole of it.  python tools/gpu_rc_cand.py [pivots ...] (0 = to the end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mincostflow_amd as M

g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)


def solve(label, env, pivots):
    for k, v in env.items():
        os.environ[k] = str(v)
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    if pivots:
        ns.set_pivot_limit(pivots)
    ns.record_trace(int(os.environ.get("RC_TRACE", "20000"))).prepare()          # RC_TRACE=3000000: the whole solve's entering arcs are compared
    t0 = time.perf_counter()
    ns.solve()
    wall = time.perf_counter() - t0
    m = ns.get_metrics(); n = max(m["iterations"], 1); e = m["engine"]
    print(f"{label} {n} pivots: {wall:.2f} s solve, {m['loop_us']/n:.2f} us/pivot | search {m['pivot_search_us']/n:.2f} pot {m['potential_update_us']/n:.2f} "
          f"tree {m['tree_update_us']/n:.2f} | avg subtree {m['potential_nodes']/n:.0f} | host-decided {e['host_decided']} requests {e['resident_requests']} "
          f"async {e['async_refreshes']} updates {e['update_launches']} inline {e['inline_updates']} cost {ns.get_total_cost() if not pivots else '-'}", flush=True)
    for k in env:
        os.environ.pop(k, None)
    return ns.trace()


for pivots in [int(x) for x in sys.argv[1:]] or [100000]:
    b = solve("candidate", {}, pivots)
    if os.environ.get("RC_CAND_ONLY") != "1":
        plain = {"MCF_HIP_CANDIDATES": 0}
        if os.environ.get("RC_PLAINEST") == "1":      # ... and none of the host driver's short cuts either: every walk is the subtree's, every list is named and shifted arc by arc
            plain.update({"MCF_NS_RELOAD": 0, "MCF_NS_SMALLER_SIDE": 0, "MCF_HIP_RC_RECOMPUTE": 0})
        a = solve("device   ", plain, pivots)
        assert np.array_equal(a, b)
        print(f"entering arcs compared: {len(a)} of each run, identical", flush=True)
