"""Resident mode: host-side and device-side share of one search as a function of the patch-list length."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
rng = np.random.default_rng(3)
m_s, n = 400_000, 100_001
a = dict(src=rng.integers(0, n, m_s, dtype=np.int32), tgt=rng.integers(0, n, m_s, dtype=np.int32),
         cost=rng.integers(-10 ** 4, 10 ** 4, m_s, dtype=np.int64), state=rng.integers(-1, 2, m_s, dtype=np.int8),
         pi=rng.integers(-10 ** 9, 1, n, dtype=np.int64))
eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, flags=0)
eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
for _ in range(50): eng.find_entering()
for k in (0, 1, 30, 300, 1000, 3000, 12000, 50000):
    nodes = rng.choice(n, size=k, replace=False).astype(np.int32)
    vals = a["pi"][nodes]
    reps = 200 if k <= 3000 else 40
    eng.park(); eng.reset_stats()
    eng.find_entering()
    t_find = 0.0
    for _ in range(reps):
        if k: eng.set_potential(nodes, vals)
        t1 = time.perf_counter()
        eng.find_entering()
        t_find += time.perf_counter() - t1
    eng.park()
    s = eng.stats()
    print(f"k={k}: find_entering {t_find/reps*1e6:.1f} us host wall, device {s['resident_scan_ns']/max(1,s['resident_requests'])/1e3:.2f} us/request "
          f"({s['resident_requests']} requests)", flush=True)
