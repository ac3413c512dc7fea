"""K independent config-3 solves in flight on ONE GPU (one host thread and one resident grid each): aggregate pivots/s for K = 1, 2, 4, 8.
GPU_MAX_HW_QUEUES must be at least K (set before HIP starts): a resident grid never leaves its hardware queue."""
import os, sys, threading, time
K_LIST = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(K_LIST)))
sys.path.insert(0, ".")
import mincostflow_amd as M
SEED = 13502460
flags = int(os.environ.get("CONC_FLAGS", str(M.ENGINE_SHARE_DEVICE)))
for K in K_LIST:
    gs = [M.netgen_like(SEED + k, 100_000, 300_000, 316, 316) for k in range(K)]
    def mk(g):
        return M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, flags).prepare()
    for rep in range(2):
        cs = [mk(g) for g in gs]
        t0 = time.perf_counter()
        th = [threading.Thread(target=c.solve) for c in cs]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        ms = [c.get_metrics() for c in cs]
    pv = sum(m["iterations"] for m in ms)
    e = ms[0]["engine"]; it = ms[0]["iterations"]
    print(f"K={K}: {pv/dt/1e3:.1f} k pivots/s aggregate, {dt*1e3:.0f} ms wall, each solve {sum(m['loop_us'] for m in ms)/K/1e3:.0f} ms | solve 0 per pivot us: total {ms[0]['loop_us']/it:.2f} search {ms[0]['pivot_search_us']/it:.2f} "
          f"potential {ms[0]['potential_update_us']/it:.2f} tree {ms[0]['tree_update_us']/it:.2f} | in-kernel {e['resident_scan_ns']/max(1,e['resident_requests'])/1e3:.2f} us/request, requests {e['resident_requests']}, resident {e['resident']} shift_grid {e['shift_grid']}", flush=True)
    del cs
