import sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
rng = np.random.default_rng(7)
for m_s, n, w in ((64_000_000, 2_001, 64), (64_000_000, 16_000, 64), (64_000_000, 2_001, 32), (1_002_000, 2_001, 64)):
    a = dict(src=rng.integers(0, n, m_s, dtype=np.int32), tgt=rng.integers(0, n, m_s, dtype=np.int32),
             cost=rng.integers(-10 ** 4, 10 ** 4, m_s, dtype=np.int64), state=rng.integers(-1, 2, m_s, dtype=np.int8),
             pi=rng.integers(-10 ** 8, 1, n, dtype=np.int64))
    for wg in (0, 512):
        eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, int_width=w, flags=M.ENGINE_DISPATCH, scan_workgroups=wg)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        b = eng.stats()["bytes_per_scan"]
        warm = eng.bench_scan(reps=10); cold = eng.bench_scan(reps=5, cold=True, flush_bytes=512 << 20)
        print(f"m_s={m_s} n={n} i{w} grid={eng.stats()['scan_workgroups']}x{eng.stats()['scan_threads']}: warm {warm[0]/1e3:.1f} us {b/warm[0]:.0f} GB/s ({b/warm[0]/80:.1f}%) | cold {cold[0]/1e3:.1f} us {b/cold[0]:.0f} GB/s ({b/cold[0]/80:.1f}%)", flush=True)
        del eng
