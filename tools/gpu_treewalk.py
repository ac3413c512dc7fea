"""SURVEY.md 8f-2, measured: FindJoinNode + FindLeavingArc (NS.cs:925-1010) on the host and on one / two lanes of the device, over real trees
of a config-3 solve.  Solves config 3 on the GPU for its pivot sequence, replays the first P pivots on the host-only stepwise driver
(mcf_ns_begin / mcf_ns_apply_pivot), dumps that spanning tree and the K entering arcs that followed, and runs tools/treewalk on the dump."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mincostflow_amd as M

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "treewalk")
if not os.path.exists(exe):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(HERE, "treewalk.hip"), "-o", exe])

g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).record_trace(1 << 20)
assert ns.solve() == 1
trace = ns.trace()
print(f"config 3: {len(trace)} pivots", flush=True)
for P in (20_000, 100_000, 170_000):
    K = 20_000
    rep = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible)
    assert rep.begin() == 0
    for a in trace[:P]:
        assert not rep.apply_pivot(int(a))
    t, it = rep.tree(), rep.internal()
    arcs = np.array([a for a in trace[P:P + K] if it["state"][a] != 0], np.int32)      # entering arcs that are non-basic in THIS tree
    path = f"/tmp/treewalk_{P}.bin"
    with open(path, "wb") as f:
        np.array([len(t["parent"]), len(it["source"]), len(arcs)], np.int32).tofile(f)
        for k in ("parent", "pred_arc", "succ_num"):
            t[k].astype(np.int32).tofile(f)
        t["pred_dir"].astype(np.int8).tofile(f)
        it["source"].astype(np.int32).tofile(f); it["target"].astype(np.int32).tofile(f)
        t["flow"].astype(np.int64).tofile(f); t["upper"].astype(np.int64).tofile(f)
        it["state"].astype(np.int8).tofile(f)
        arcs.tofile(f)
    print(f"--- tree after {P} pivots, the {len(arcs)} entering arcs that followed", flush=True)
    subprocess.check_call([exe, path])
