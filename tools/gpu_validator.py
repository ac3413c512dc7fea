"""Validator at config 5's shape: where the time goes (MCF_VAL_SKIP: 1 = no scatter, 2 = no potential gathers; both are measurement aids
that give wrong results) and the grid size (MCF_VAL_G), for NETGEN-like arc order and for uniformly random end points."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
rng = np.random.default_rng(8)
n, m = 1_000_000, 8_000_000
g = M.netgen_like(13502460, n, m, 1000, 1000)
flow = np.where(rng.random(m) < 0.12, rng.integers(0, 1000, m), 0).astype(np.int64)
pi = -rng.integers(0, 10 ** 7, n).astype(np.int64)
for label, src, tgt in (("netgen-like", g.source, g.target), ("uniform", rng.integers(0, n, m, dtype=np.int32), rng.integers(0, n, m, dtype=np.int32))):
    v = M.SolutionValidator(n, m).upload_network(src, tgt, g.lower, g.upper, g.cost, g.supply).upload_solution(flow, pi)
    for skip in ("0", "1", "2", "3"):
        os.environ["MCF_VAL_SKIP"] = skip
        for grp in ("512", "1024"):
            os.environ["MCF_VAL_G"] = grp
            r = [v.run(0, 0)["kernel_us"] for _ in range(8)][2:]
            print(f"{label} skip {skip} groups {grp}: {np.mean(r):.1f} us  -> {360e6/np.mean(r)/1e3:.0f} GB/s", flush=True)
