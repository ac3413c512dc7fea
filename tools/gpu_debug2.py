import sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
from oracle import ns_oracle as O
n = 2
for costs in ([-4, -7, -8, 0], [-4, -7, -8, -9], [0, -7, -8, 0], [0, 0, -3, -8], [-8, -7, -3, -1], [-1, -2, -3, -4, -5, -6, -7, -8]):
    m_s = len(costs)
    src = np.zeros(m_s, np.int32); tgt = np.ones(m_s, np.int32)
    for rule, opt in ((M.PivotRule.BlockSearch, True), (M.PivotRule.BlockSearch, False), (M.PivotRule.FirstEligible, True), (M.PivotRule.BestEligible, True)):
        for w in (64, 32):
            eng = M.PivotEngine(n, m_s, m_s, rule=rule, optimized=opt, block_size=374, int_width=w)
            eng.upload(src, tgt, np.array(costs, np.int64), np.ones(m_s, np.int8), np.zeros(n, np.int64))
            print(costs, "rule", rule, "opt", opt, "w", w, "->", eng.find_entering(), flush=True)
