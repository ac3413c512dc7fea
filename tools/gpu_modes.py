"""Dispatch mode vs resident mode: per-pivot cost on the BASELINE configs.  Not a test."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M

cases = (("config2_block_i32", M.netgen_like(13502460, 10_000, 30_000, 100, 100), M.PivotRule.BlockSearch),
         ("config3_best_i64", M.netgen_like(13502460, 100_000, 300_000, 316, 316), M.PivotRule.BestEligible),
         ("config4_best", M.assignment(42, 1000, 1, 100), M.PivotRule.BestEligible))
out = {}
for name, g, rule in cases:
    for mode, flags in (("dispatch", M.ENGINE_SAMPLE_KERNEL_TIME | M.ENGINE_DISPATCH), ("resident", M.ENGINE_RESIDENT), ("candidates", M.ENGINE_CANDIDATES)):
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True).set_device(0, 0, 0, flags)
        ns.prepare()
        if name.startswith("netgen_1M"):
            # bounded: stepwise would need the engine; just run the full solve only in resident mode? too long -> skip dispatch
            if mode == "dispatch":
                continue
        t = time.perf_counter()
        st = ns.solve()
        wall = time.perf_counter() - t
        m = ns.get_metrics(); e = m["engine"]
        it = max(m["iterations"], 1)
        r = dict(status=st, cost=ns.get_total_cost(), pivots=m["iterations"], loop_ms=m["loop_us"] / 1e3, wall_ms=wall * 1e3,
                 us_per_pivot=m["loop_us"] / it, search_us=m["pivot_search_us"] / it, tree_us=m["tree_update_us"] / it,
                 pot_us=m["potential_update_us"] / it, wait_us=e["host_wait_ns"] / max(e["searches"], 1) / 1e3,
                 post_or_launch_us=e["host_launch_ns"] / max(e["searches"], 1) / 1e3, inline=e["inline_updates"], staged=e["update_launches"],
                 grid=e["scan_workgroups"], resident=e["resident"], resident_launches=e["resident_launches"],
                 resident_requests=e["resident_requests"], in_kernel_scan_us=e["resident_scan_ns"] / max(e["resident_requests"], 1) / 1e3,
                 kernel_us=e["timed_scan_ns"] / max(e["timed_scans"], 1) / 1e3, host_decided=e["host_decided"], avg_subtree=m["potential_nodes"] / it)
        out[f"{name}/{mode}"] = r
        print(name, mode, json.dumps(r), flush=True)
        del ns
json.dump(out, open("gpurun_out/modes.json", "w"), indent=1)
