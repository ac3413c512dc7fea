"""Config 5 as R arc shards driven by one host thread on ONE GPU (mcf_ns_set_shard_group): every shard with its own candidate cache against
every search on the device (MCF_HIP_CANDIDATES=0), first N pivots, same pivots.  python tools/gpu_shard_cand.py [pivots] [R ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mincostflow_amd as M

g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
pivots = int(sys.argv[1]) if len(sys.argv) > 1 else 200000


def solve(label, shards, env):
    for k, v in env.items():
        os.environ[k] = str(v)
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    if shards > 1:
        ns.set_shard_group([0] * shards)
    ns.set_pivot_limit(pivots).record_trace(20000).prepare()
    ns.solve()
    m = ns.get_metrics(); n = max(m["iterations"], 1); e = m["engine"]
    print(f"{label} R={shards}: {m['loop_us']/n:.2f} us/pivot | search {m['pivot_search_us']/n:.2f} pot {m['potential_update_us']/n:.2f} | shard 0: candidates {e['candidates']} "
          f"resident {e['resident']} host-decided {e['host_decided']} requests {e['resident_requests']} scans {e['scan_launches']}", flush=True)
    for k in env:
        os.environ.pop(k, None)
    return ns.trace()


ref = solve("one engine      ", 1, {})
for r in [int(x) for x in sys.argv[2:]] or [2, 3]:
    a = solve("caches per shard", r, {})
    b = solve("device only     ", r, {"MCF_HIP_CANDIDATES": 0})
    assert np.array_equal(a, ref) and np.array_equal(b, ref)
