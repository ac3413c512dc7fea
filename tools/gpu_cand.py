"""Candidate-cache timing on config 3: solves under a few settings of the knobs; MCF_HIP_CAND_DEBUG prints the host-side breakdown."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import mincostflow_amd as M
g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
flags = int(sys.argv[1]); label = sys.argv[2]
best = None
for rep in range(3):
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64, 0, flags).prepare()
    assert ns.solve() == 1
    m = ns.get_metrics()
    best = m if best is None or m["loop_us"] < best["loop_us"] else best
    del ns
it = best["iterations"]; e = best["engine"]
print(f"{label}: {best['loop_us']/1e3:.1f} ms, {it/(best['loop_us']/1e6)/1e3:.1f} k pivots/s, {best['loop_us']/it:.2f} us/pivot | search {best['pivot_search_us']/it:.2f} pot {best['potential_update_us']/it:.2f} "
      f"tree {best['tree_update_us']/it:.2f} | device requests {e['resident_requests']} host {e['host_decided']} async {e['async_refreshes']}", flush=True)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def run(flags, label, env=None):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in (env or {}).items()})
    out = subprocess.run([sys.executable, "-c", CHILD, str(flags), label], env=e, capture_output=True, text=True)
    lines = [ln for ln in (out.stdout + out.stderr).splitlines() if ln.startswith(label) or ln.startswith("[cand]")]
    print("\n".join(lines[-2:]), flush=True)

run(128, "every search on the device")
run(0, "candidates default", {"MCF_HIP_CAND_DEBUG": 1})
for piece, lines in ((1024, 192), (2048, 384), (512, 96)):
    run(0, f"walk piece {piece} stream lines {lines}", {"MCF_NS_WALK_PIECE": piece, "MCF_HIP_STREAM_LINES": lines})
for nodes in (48, 200):
    run(0, f"nodes={nodes}", {"MCF_HIP_CAND_NODES": nodes})
for low in (2, 12):
    run(0, f"refresh_low={low}", {"MCF_HIP_CAND_REFRESH": low})
