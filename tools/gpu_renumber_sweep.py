"""Config 3 with the node relabelling forced to several intervals (MCF_NS_RENUMBER = walked nodes between two relabellings, in multiples of
the node count; unset = the solver's own cost-aware policy): solve time, relabellings, share of walk steps that left the id order."""
import os, sys
sys.path.insert(0, ".")
import mincostflow_amd as M
g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
for every in (None, "128", "64", "32", "16", "8", "0"):
    if every is None: os.environ.pop("MCF_NS_RENUMBER", None)
    else: os.environ["MCF_NS_RENUMBER"] = every
    best = None
    for rep in range(3):
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0).prepare()
        ns.solve()
        m = ns.get_metrics(); it = m["iterations"]
        row = (m["loop_us"] / 1e3, m["engine"]["renumberings"], m["potential_update_us"] / it, m["pivot_search_us"] / it)
        if best is None or row[0] < best[0]: best = row
        del ns
    print(f"MCF_NS_RENUMBER={every}: best of 3: {best[0]:.1f} ms, {best[1]} relabellings, potential {best[2]:.2f} us/pivot, search {best[3]:.2f}", flush=True)
