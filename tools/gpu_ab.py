"""A/B of two builds of the library in one box (boxes differ by more than most changes): build the baseline commit into
mincostflow_amd/libmcf_hip_base.so (git archive <commit> mincostflow_amd/csrc include | tar -x -C /tmp/b && make -C /tmp/b/mincostflow_amd/csrc),
then run this; MCF_AB_LIB names the library file a child process loads."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import mincostflow_amd._lib as L
    if os.environ.get("MCF_AB_LIB"):
        L.LIB_PATH = os.path.join(os.path.dirname(L.__file__), os.environ["MCF_AB_LIB"])
    import mincostflow_amd as M
    g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, 0).prepare()
    assert ns.solve() == 1
    m = ns.get_metrics(); it = m["iterations"]; e = m["engine"]
    print(f"{sys.argv[1]}: loop {m['loop_us']/it:.2f} us/pivot, wait {e['host_wait_ns']/it/1e3:.2f}, in-kernel {e['resident_scan_ns']/max(1,e['resident_requests'])/1e3:.2f}, grid {e['scan_workgroups']}x{e['scan_threads']}", flush=True)
else:
    for rep in range(3):
        for name, env in (("base", {"MCF_AB_LIB": "libmcf_hip_base.so"}), ("new", {})):
            subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))
