"""RC layout (reduced costs kept per arc) against the gathering scan on config 5's arrays: scan-only kernel times (warm / cold) for a few
geometries, then the first pivots of the solve in both layouts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mincostflow_amd as M

SEED = 13502460
g5 = M.netgen_like(SEED, 1_000_000, 8_000_000, 1000, 1000)
ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
assert ns5.begin() == 0
it = ns5.internal()
ms = it["search_arc_num"]
arrs = (it["source"][:ms], it["target"][:ms], it["cost"][:ms], it["state"][:ms], it["pi"])

def scan(label, env):
    for k, v in env.items():
        os.environ[k] = str(v)
    eng = M.PivotEngine(g5.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, flags=M.ENGINE_DISPATCH)
    eng.upload(*arrs)
    st = eng.stats()
    warm = eng.bench_scan(reps=30)
    cold = eng.bench_scan(reps=10, cold=True, flush_bytes=512 << 20)
    rd = st["scan_bytes_read"]
    print(f"{label:44s} grid {st['scan_workgroups']:5d} rc={st['rc_layout']} read {rd/1e6:6.1f} MB | warm {warm[0]/1e3:6.1f} us (min {warm[1]/1e3:6.1f}) {rd/warm[0]/8000:5.2f} of peak | "
          f"cold {cold[0]/1e3:6.1f} us {rd/cold[0]/8000:5.2f} of peak | SURVEY bytes {st['bytes_per_scan']/1e6:.0f} MB -> {st['bytes_per_scan']/cold[0]:.0f} GB/s equivalent", flush=True)
    f = eng.find_entering()
    del eng
    for k in env:
        os.environ.pop(k, None)
    return f

ref = scan("gathering scan, bucketed (round 1)", {"MCF_HIP_RC": 0})
if len(sys.argv) > 1 and sys.argv[1] == "threads":
    for threads in (256, 512, 1024):
        for wg in (256, 512, 1024, 2048):
            for unroll in (1, 2):
                got = scan(f"RC layout {threads} threads unroll {unroll} max workgroups {wg}", {"MCF_HIP_RC": 1, "MCF_HIP_RC_THREADS": threads, "MCF_HIP_UNROLL": unroll, "MCF_HIP_MAXWG": wg})
                assert got == ref, (got, ref)
    sys.exit(0)
for unroll in (1, 2, 4):
    for wg in (1024, 2048, 4096, 8192):
        got = scan(f"RC layout unroll {unroll} max workgroups {wg}", {"MCF_HIP_RC": 1, "MCF_HIP_UNROLL": unroll, "MCF_HIP_MAXWG": wg})
        assert got == ref, (got, ref)

def solve(label, env, pivots=4000):
    for k, v in env.items():
        os.environ[k] = str(v)
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    ns.set_pivot_limit(pivots).record_trace(pivots).prepare()
    ns.solve()
    m = ns.get_metrics(); n = max(m["iterations"], 1)
    e = m["engine"]
    print(f"{label}: {m['loop_us']/n:.1f} us/pivot | search {m['pivot_search_us']/n:.1f} pot {m['potential_update_us']/n:.1f} tree {m['tree_update_us']/n:.2f} | "
          f"avg subtree {m['potential_nodes']/n:.0f} nodes | scans {e['scan_launches']} updates {e['update_launches']} inline {e['inline_updates']}", flush=True)
    tr = ns.trace()
    for k in env:
        os.environ.pop(k, None)
    return tr

a = solve("first pivots, gathering scan", {"MCF_HIP_RC": 0})
b = solve("first pivots, RC layout     ", {})
assert np.array_equal(a, b)
print("identical pivots")
