"""Scan-only bandwidth sweep at 64M arcs; variants selected with MCF_HIP_UNROLL / MCF_HIP_NT / MCF_HIP_MAXWG env vars."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
rng = np.random.default_rng(7)
m_s = 64_000_000
for n in (2_001, 100_001):
    a = dict(src=rng.integers(0, n, m_s, dtype=np.int32), tgt=rng.integers(0, n, m_s, dtype=np.int32),
             cost=rng.integers(-10 ** 4, 10 ** 4, m_s, dtype=np.int64), state=rng.integers(-1, 2, m_s, dtype=np.int8),
             pi=rng.integers(-10 ** 9, 1, n, dtype=np.int64))
    for unroll in ("1", "2", "4"):
        for nt in ("0", "1"):
            for wg in ("1024", "2048", "4096", "8192"):
                if unroll == "1" and nt == "1":
                    continue
                os.environ["MCF_HIP_UNROLL"], os.environ["MCF_HIP_NT"], os.environ["MCF_HIP_MAXWG"] = unroll, nt, wg
                eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, flags=M.ENGINE_DISPATCH)
                eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
                b = eng.stats()["bytes_per_scan"]
                warm = eng.bench_scan(reps=8)
                cold = eng.bench_scan(reps=4, cold=True, flush_bytes=512 << 20)
                print(f"n={n} unroll={unroll} nt={nt} maxwg={wg} grid={eng.stats()['scan_workgroups']}: warm {warm[1]/1e3:.1f} us {b/warm[1]:.0f} GB/s | cold {cold[1]/1e3:.1f} us {b/cold[1]:.0f} GB/s", flush=True)
                del eng
