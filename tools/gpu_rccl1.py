"""The RCCL exchange of a RESIDENT shard engine with a world of one rank (all this box has): config 5's first pivots through mcf_ns_set_sharding,
in the main thread / in a second thread, before and after torch has initialised the device, and with the resident grid leaving more CUs."""
import os, sys, time, threading
sys.path.insert(0, ".")
import mincostflow_amd as M
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)

def leg(tag, env):
    os.environ.update(env)
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0)
    ns.set_sharding(M.comm_unique_id(), 0, 1).set_pivot_limit(P).record_trace(P).prepare()
    t0 = time.perf_counter(); ns.solve(); dt = time.perf_counter() - t0
    m = ns.get_metrics(); e = m["engine"]
    print(tag, env, f"{dt / max(1, m['iterations']) * 1e6:.1f} us per pivot | resident {e['resident']} candidates {e['candidates']} workgroups {e['scan_workgroups']} host-decided {e['host_decided']} requests {e['resident_requests']}", flush=True)
    for k in env: os.environ.pop(k)

def in_thread(tag, env):
    th = threading.Thread(target=leg, args=(tag, env), daemon=True); th.start(); th.join(timeout=120.0)
    if th.is_alive(): print(tag, "still running after 120 s", flush=True); os._exit(1)

leg("main thread", {"MCF_NS_RCCL_FREE_CUS": "8"})
in_thread("second thread", {"MCF_NS_RCCL_FREE_CUS": "8"})
import torch
torch.cuda.set_device(0); x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
leg("main thread, torch up", {"MCF_NS_RCCL_FREE_CUS": "8"})
in_thread("second thread, torch up", {"MCF_NS_RCCL_FREE_CUS": "8"})
in_thread("second thread, torch up", {"MCF_HIP_RCCL_RESIDENT": "0"})
