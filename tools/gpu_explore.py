"""GPU exploration: per-pivot round trip vs scan grid size, scan-only bandwidth, solve timings.  Not a test."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mincostflow_amd as M

out = {}
rng = np.random.default_rng(1)


def soa(m_s, n):
    return dict(src=rng.integers(0, n, m_s, dtype=np.int32), tgt=rng.integers(0, n, m_s, dtype=np.int32),
                cost=rng.integers(-10 ** 4, 10 ** 4, m_s, dtype=np.int64), state=rng.integers(-1, 2, m_s, dtype=np.int8),
                pi=rng.integers(-10 ** 9, 1, n, dtype=np.int64))


# 1. round trip of one search (launch -> records in pinned memory -> host merge) vs grid
m_s, n = 400_000, 100_001
a = soa(m_s, n)
rt = {}
for wg in (0, 32, 64, 128, 256, 391):
    eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, scan_workgroups=wg, flags=M.ENGINE_TIME_EVERY_KERNEL)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for _ in range(200):
        eng.find_entering()
    eng.reset_stats()
    t = time.perf_counter()
    N = 3000
    for _ in range(N):
        eng.find_entering()
    dt = (time.perf_counter() - t) / N
    st = eng.stats()
    # with a 30-node inline patch
    nodes = np.arange(30, dtype=np.int32)
    t = time.perf_counter()
    for _ in range(N):
        eng.update_potential(nodes, 1)
        eng.find_entering()
    dt2 = (time.perf_counter() - t) / N
    nodes = np.arange(500, dtype=np.int32)
    t = time.perf_counter()
    for _ in range(N):
        eng.update_potential(nodes, 1)
        eng.find_entering()
    dt3 = (time.perf_counter() - t) / N
    rt[wg] = dict(grid=st["scan_workgroups"], python_roundtrip_us=dt * 1e6, kernel_us=st["timed_scan_ns"] / max(st["timed_scans"], 1) / 1e3,
                  host_wait_us=st["host_wait_ns"] / st["searches"] / 1e3, host_launch_us=st["host_launch_ns"] / st["searches"] / 1e3,
                  with_inline30_us=dt2 * 1e6, with_staged500_us=dt3 * 1e6)
    print("roundtrip", wg, rt[wg], flush=True)
out["roundtrip_400k"] = rt

# 2. scan-only bandwidth
bw = {}
for m_s, n in ((400_000, 100_001), (1_000_000, 2001), (1_000_000, 100_001), (8_000_000, 1_000_001), (64_000_000, 1_000_001), (64_000_000, 2001)):
    for width in (64, 32):
        a = soa(m_s, n)
        if width == 32:
            a["pi"] = (a["pi"] // 4).astype(np.int64)
        eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, int_width=width)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        bytes_ = eng.stats()["bytes_per_scan"]
        warm = eng.bench_scan(reps=30)
        cold = eng.bench_scan(reps=10, cold=True, flush_bytes=512 << 20)
        key = f"{m_s}/{n}/i{width}"
        bw[key] = dict(bytes=bytes_, warm_avg_us=warm[0] / 1e3, warm_min_us=warm[1] / 1e3, cold_avg_us=cold[0] / 1e3,
                       warm_GBs=bytes_ / warm[0], cold_GBs=bytes_ / cold[0], grid=eng.stats()["scan_workgroups"])
        print("scan", key, bw[key], flush=True)
        del eng
out["scan_bw"] = bw

# 3. solves
sol = {}
for name, g, rule in (("config2_block_i32", M.netgen_like(13502460, 10_000, 30_000, 100, 100), M.PivotRule.BlockSearch),
                      ("config2_best", M.netgen_like(13502460, 10_000, 30_000, 100, 100), M.PivotRule.BestEligible),
                      ("config3_best", M.netgen_like(13502460, 100_000, 300_000, 316, 316), M.PivotRule.BestEligible),
                      ("config3_block", M.netgen_like(13502460, 100_000, 300_000, 316, 316), M.PivotRule.BlockSearch),
                      ("config4_best", M.assignment(42, 1000, 1, 100), M.PivotRule.BestEligible)):
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True).set_device(0, 0, 0, M.ENGINE_SAMPLE_KERNEL_TIME)
    ns.prepare()
    st = ns.solve()
    m = ns.get_metrics()
    e = m["engine"]
    sol[name] = dict(status=st, cost=ns.get_total_cost(), pivots=m["iterations"], loop_ms=m["loop_us"] / 1e3, setup_ms=m["setup_us"] / 1e3,
                     us_per_pivot=m["loop_us"] / max(m["iterations"], 1), search_us=m["pivot_search_us"] / max(m["iterations"], 1),
                     tree_us=m["tree_update_us"] / max(m["iterations"], 1), pot_us=m["potential_update_us"] / max(m["iterations"], 1),
                     kernel_us=e["timed_scan_ns"] / max(e["timed_scans"], 1) / 1e3, inline=e["inline_updates"], staged=e["update_launches"],
                     width=m["int_width"], grid=e["scan_workgroups"], wait_us=e["host_wait_ns"] / max(e["searches"], 1) / 1e3,
                     launch_us=e["host_launch_ns"] / max(e["searches"], 1) / 1e3)
    print("solve", name, sol[name], flush=True)
out["solves"] = sol
json.dump(out, open("gpurun_out/explore.json", "w"), indent=1)
