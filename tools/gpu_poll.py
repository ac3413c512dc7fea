"""A/B of the resident poll parameters inside one process (env vars are read at engine creation)."""
import os, sys, json
sys.path.insert(0, ".")
import mincostflow_amd as M
g3 = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
g2 = M.netgen_like(13502460, 10_000, 30_000, 100, 100)
for rep in (1, 2, 3):
    for replicas, sleep in ((8, 1), (16, 1), (4, 1), (8, 0), (8, 3), (16, 0)):
        os.environ["MCF_HIP_POLL_REPLICAS"], os.environ["MCF_HIP_POLL_SLEEP"] = str(replicas), str(sleep)
        out = []
        for g, rule in ((g3, M.PivotRule.BestEligible), (g2, M.PivotRule.BlockSearch)):
            ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True).prepare()
            assert ns.solve() == 1
            m = ns.get_metrics(); e = m["engine"]; it = m["iterations"]
            out.append((round(m["loop_us"] / it, 2), round(e["host_wait_ns"] / e["searches"] / 1e3, 2), round(e["resident_scan_ns"] / max(e["resident_requests"], 1) / 1e3, 2)))
        print(f"replicas={replicas} sleep={sleep}: config3 us/pivot, wait, in-kernel = {out[0]} | config2 = {out[1]}", flush=True)
