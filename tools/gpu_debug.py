import sys
import numpy as np
sys.path.insert(0, ".")
import mincostflow_amd as M
from oracle import ns_oracle as O

g = M.netgen_like(13502460, 2000, 8000, 40, 40)
p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
for flags in (M.ENGINE_NO_INLINE_UPDATE, 0):
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).record_trace(1 << 20)
    ns.set_device(0, 64, 0, flags)
    st = ns.solve()
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BEST)
    st_o, tr_o = o.solve(trace_cap=1 << 20)
    tr = ns.trace()
    n = min(len(tr), len(tr_o))
    diff = np.nonzero(tr[:n] != tr_o[:n])[0]
    print("flags", flags, "status", st, st_o, "len", len(tr), len(tr_o), "first diff", diff[:5], flush=True)
    if len(diff):
        i = int(diff[0])
        print("  at", i, "gpu", tr[i - 2:i + 3], "oracle", tr_o[i - 2:i + 3])
        # replay oracle to pivot i and evaluate both candidates
        o2 = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BEST); o2.init()
        for k in range(i):
            o2.apply_pivot(int(tr_o[k]))
        a = o2.internal_arrays()
        for e in (int(tr[i]), int(tr_o[i])):
            rc = int(a["state"][e]) * (int(a["cost"][e]) + int(a["pi"][a["src"][e]]) - int(a["pi"][a["tgt"][e]]))
            print("   arc", e, "state", a["state"][e], "src", a["src"][e], "tgt", a["tgt"][e], "rc", rc, "m_s", o2.search_arc_num)
        print("   last subtree of previous pivot", o2.last_subtree, "sigma", o2.last_sigma)

# block opt random case
rng = np.random.default_rng(1234 + 64 + 10 * 2 + 1)
def soa(m_s, n, cs, ps, extra=7):
    cap = m_s + extra
    return dict(src=rng.integers(0, n, cap, dtype=np.int32), tgt=rng.integers(0, n, cap, dtype=np.int32),
                cost=rng.integers(-cs, cs + 1, cap, dtype=np.int64), state=rng.integers(-1, 2, cap, dtype=np.int8),
                pi=rng.integers(-ps, 1, n, dtype=np.int64))
for m_s, n, span in [(1023, 40, 3), (1024, 300, 2), (4097, 5000, 4)]:
    for trial in range(6):
        a = soa(m_s, n, span, span * 3)
        block = int(rng.integers(1, 700))
        for opt in (True, False):
            eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BlockSearch, optimized=opt, block_size=block)
            eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
            na = 0
            for it in range(8):
                f, e, c, na2 = O.scan_block(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"], block, opt, na)
                f2, e2, c2 = eng.find_entering()
                ok = (f, e, c) == (f2, e2, c2) and (not f or eng.next_arc == na2)
                if not ok:
                    pos = lambda x: (x - (0 if na >= m_s else na)) % m_s
                    print("BLOCK MISMATCH m_s", m_s, "block", block, "opt", opt, "it", it, "na", na, "oracle", (f, e, c, na2), "gpu", (f2, e2, c2, eng.next_arc),
                          "pos", pos(e), pos(e2), "rank", pos(e) // block, pos(e2) // block, flush=True)
                    break
                if f:
                    na = na2
                else:
                    break
                na = int(rng.integers(0, m_s + 1)); eng.next_arc = na
print("done")
