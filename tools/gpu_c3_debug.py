import os, sys
sys.path.insert(0, "/root/repo")
import mincostflow_amd as M
g = M.netgen_like(13502460, 100_000, 300_000, 1000, 1000)
for k in range(2):
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    ns.prepare(); ns.solve()
    m = ns.get_metrics(); print(m["iterations"], m["loop_us"] / m["iterations"], flush=True)
    del ns
