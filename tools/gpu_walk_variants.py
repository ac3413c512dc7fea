"""Config 3 with the big walks handed to the register-resident candidate grid in its three ways: node ids (MCF_NS_RUNS=0), runs of consecutive
ids (default), a reload of the bound potentials inside the grid from N nodes on (MCF_HIP_SHIFT_RELOAD=N).  Best of three solves each."""
import os, sys
sys.path.insert(0, ".")
import mincostflow_amd as M
g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
cases = [("node ids", {"MCF_NS_RUNS": "0", "MCF_HIP_SHIFT_RELOAD": "0"}), ("runs", {"MCF_HIP_SHIFT_RELOAD": "0"}),
         ("runs, reload from 32768 nodes", {"MCF_HIP_SHIFT_RELOAD": "32768"}), ("runs, reload from 8192 nodes", {"MCF_HIP_SHIFT_RELOAD": "8192"}),
         ("node ids", {"MCF_NS_RUNS": "0", "MCF_HIP_SHIFT_RELOAD": "0"}), ("runs", {"MCF_HIP_SHIFT_RELOAD": "0"})]
for label, env in cases:
    os.environ.update(env)
    best = None
    for rep in range(3):
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0).prepare()
        ns.solve()
        m = ns.get_metrics(); it = m["iterations"]; e = m["engine"]
        row = (m["loop_us"] / 1e3, m["potential_update_us"] / it, m["pivot_search_us"] / it, e["resident_scan_ns"] / max(1, e["resident_requests"]) / 1e3, e["rc_reloads_in_grid"], e["shift_lists"])
        if best is None or row[0] < best[0]: best = row
        del ns
    print(f"{label:32s}: {best[0]:.1f} ms = {190580 / best[0]:.1f} k pivots/s | potential {best[1]:.2f} search {best[2]:.2f} us/pivot | in-kernel {best[3]:.2f} us/request | reloads {best[4]} shift lists {best[5]}", flush=True)
    for k in env: os.environ.pop(k)
