"""K independent config-3 solves in flight on ONE GPU (one host thread and one resident grid each): aggregate pivots/s.
  python tools/gpu_concurrent.py shared 1 2 4      MCF_ENGINE_SHARE_DEVICE grids of 256 workgroups (two fit a CU)
  python tools/gpu_concurrent.py split 1 2 4 8     every solver gets 256 / K workgroups of its own (MCF_NS_RESIDENT_WORKGROUPS): CUs partitioned per instance
GPU_MAX_HW_QUEUES must be at least K (set before HIP starts): a resident grid never leaves its hardware queue."""
import json, os, sys, threading, time
as_json = "--json" in sys.argv               # one JSON object on the last line (bench.py runs this script in a process of its own)
argv = [a for a in sys.argv[1:] if a != "--json"]
mode = argv[0] if argv else "shared"
K_LIST = [int(x) for x in argv[1:]] or [1, 2, 4]
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, max(K_LIST))))
sys.path.insert(0, ".")
import numpy as np
import mincostflow_amd as M
SEED = 13502460
flags = M.ENGINE_SHARE_DEVICE if mode == "shared" else 0
ref = None
out = {}
for K in K_LIST:
    if mode == "split":
        os.environ["MCF_NS_RESIDENT_WORKGROUPS"] = str(max(8, 256 // K // 8 * 8))
    # K different instances of the same shape (solver 0 has the headline instance: bench.py compares its pivot count with the timed solve's);
    # MCF_CONC_SAME=1: the same instance K times -- the solves then hit their big subtrees in lock-step, which costs a third of the throughput
    gs = [M.netgen_like(SEED + (0 if os.environ.get("MCF_CONC_SAME") else k), 100_000, 300_000, 316, 316) for k in range(K)]
    def mk(g):
        return M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, flags).record_trace(1 << 18).prepare()
    for rep in range(2):
        cs = [mk(g) for g in gs]
        t0 = time.perf_counter()
        def run(c, k):
            # MCF_CONC_PIN=<first cpu>: solver k's host thread stays on cpu <first cpu> + k (sched_setaffinity with pid 0 binds the calling thread)
            if os.environ.get("MCF_CONC_PIN"):
                try: os.sched_setaffinity(0, {int(os.environ["MCF_CONC_PIN"]) + k})
                except OSError: pass
            c.solve()
        th = [threading.Thread(target=run, args=(c, k)) for k, c in enumerate(cs)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        ms = [c.get_metrics() for c in cs]
    if ref is None:
        ref = cs[0].trace().copy()
    same = bool(np.array_equal(cs[0].trace(), ref))
    pv = sum(m["iterations"] for m in ms)
    e = ms[0]["engine"]; it = ms[0]["iterations"]
    print(f"{mode} K={K}: {pv/dt/1e3:.1f} k pivots/s aggregate, {dt*1e3:.0f} ms wall, each solve {sum(m['loop_us'] for m in ms)/K/1e3:.0f} ms | solve 0 per pivot us: total {ms[0]['loop_us']/it:.2f} search {ms[0]['pivot_search_us']/it:.2f} "
          f"potential {ms[0]['potential_update_us']/it:.2f} tree {ms[0]['tree_update_us']/it:.2f} | workgroups {e['scan_workgroups']} rc_layout {e['rc_layout']} in-kernel {e['resident_scan_ns']/max(1,e['resident_requests'])/1e3:.2f} us/request, "
          f"requests {e['resident_requests']}, host-decided {e['host_decided']}, launches {e['resident_launches']}, relabellings {e['renumberings']} | solve 0's pivots identical to the first run's: {same}", flush=True)
    out[str(K)] = {"solves_in_flight": K, "workgroups_per_solve": int(e["scan_workgroups"]), "pivots_per_s": pv / dt, "seconds": dt, "solve_ms_each": sum(m["loop_us"] for m in ms) / K / 1e3,
                   "all_optimal": all(c.status == M.SolverStatus.Optimal for c in cs), "all_resident": all(m["engine"]["resident"] == 1 for m in ms),
                   "reduced_costs_kept_per_arc": bool(e["rc_layout"]), "in_kernel_us_per_request": e["resident_scan_ns"] / max(1, e["resident_requests"]) / 1e3,
                   "pivots_of_each_solve": [int(m["iterations"]) for m in ms], "first_solver_same_pivots_as_the_first_run": same}
    del cs
if as_json:
    print(json.dumps(out), flush=True)
