"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.
Bit-exact: this is integer work.  Run with `pytest -m gpu` on an MI355X."""
import ctypes as C

import time

import numpy as np
import pytest

import kat_data as K
import mincostflow_amd as M
from helpers import fixtures, load, problem_from_dict, validate_solution
from mincostflow_amd import _lib as L
from oracle import ns_oracle as O

pytestmark = pytest.mark.gpu

RULES = {O.RULE_FIRST: M.PivotRule.FirstEligible, O.RULE_BEST: M.PivotRule.BestEligible, O.RULE_BLOCK: M.PivotRule.BlockSearch}


def _random_soa(rng, m_s, n, cost_span, pi_span, extra=7):
    cap = m_s + extra
    return dict(src=rng.integers(0, n, cap, dtype=np.int32), tgt=rng.integers(0, n, cap, dtype=np.int32),
                cost=rng.integers(-cost_span, cost_span + 1, cap, dtype=np.int64),
                state=rng.integers(-1, 2, cap, dtype=np.int8),
                pi=rng.integers(-pi_span, 1, n, dtype=np.int64))


def _oracle_scan(rule, optimized, a, m_s, block, next_arc, vw=4):
    if rule == O.RULE_BEST:
        f, e, c = O.scan_best(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"]); return f, e, c, next_arc
    if rule == O.RULE_FIRST:
        return O.scan_first(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"], next_arc)
    return O.scan_block(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"], block, optimized, next_arc, vector_width=vw)


# default = resident grid, with the candidate cache where it applies (Best Eligible, register-resident arcs, sparse graph)
MODES = [pytest.param(M.ENGINE_NO_CANDIDATES, id="resident"), pytest.param(M.ENGINE_DISPATCH, id="dispatch"), pytest.param(0, id="candidates"),
         pytest.param(M.ENGINE_SHARE_DEVICE | M.ENGINE_NO_CANDIDATES, id="resident-shared"),      # resident grid without register-resident potentials
         pytest.param(M.ENGINE_SHARE_DEVICE, id="candidates-shared"),
         # reduced costs kept per arc (what large sparse instances use), forced onto these sizes: one dispatch per search, the resident grid
         # with the arcs in LDS, and the resident grid streaming them from memory
         pytest.param("rc", id="rc-layout"), pytest.param("rc-resident", id="rc-resident-lds"), pytest.param("rc-stream", id="rc-resident-stream"),
         # ... and the candidate cache on top of either resident RC grid (Best Eligible on sparse graphs; the other rules run as above)
         pytest.param("rc-cand", id="rc-candidates-lds"), pytest.param("rc-cand-stream", id="rc-candidates-stream")]
CAND_MODES = (0, M.ENGINE_SHARE_DEVICE)


def _mode_flags(mode, monkeypatch):
    """Engine flags of a MODES entry; the "rc*" entries force the RC layout (MCF_HIP_RC=1) onto any size."""
    if isinstance(mode, str):
        monkeypatch.setenv("MCF_HIP_RC", "1")
        if mode in ("rc-stream", "rc-cand-stream"):
            monkeypatch.setenv("MCF_HIP_RC_LDS", "0")
        return M.ENGINE_DISPATCH if mode == "rc" else (0 if mode.startswith("rc-cand") else M.ENGINE_NO_CANDIDATES)
    return mode


# (rule, EnableOptimizedPivot flavour, Vector<long>.Count of the reference's host: 4 = x64, 0 = not hardware accelerated; only the optimized
# Block Search reads it -- BlockSearchPivotOptimized.cs:74-99)
RULE_CASES = [pytest.param(O.RULE_BEST, True, 4, id="best"), pytest.param(O.RULE_BLOCK, True, 4, id="block-opt-v4"), pytest.param(O.RULE_BLOCK, True, 0, id="block-opt-v0"),
              pytest.param(O.RULE_BLOCK, True, 8, id="block-opt-v8"), pytest.param(O.RULE_BLOCK, False, 4, id="block-plain"), pytest.param(O.RULE_FIRST, True, 4, id="first")]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("width", [64, 32])
@pytest.mark.parametrize("rule,optimized,vw", RULE_CASES)
def test_scan_matches_oracle_on_random_arrays(width, rule, optimized, vw, mode, monkeypatch):
    """Ragged sizes, heavy ties (tiny cost range), random patches between searches (short lists, long lists), in both
    engine modes: the resident grid fed through the mailbox and one dispatch per search."""
    rc_layout = isinstance(mode, str)
    rc_resident = rc_layout and mode != "rc"
    mode = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(1234 + width + 10 * rule + optimized + (0 if vw == 4 else 1000 + vw))
    for m_s, n, span in [(1, 2, 3), (3, 2, 2), (4, 5, 2), (5, 3, 50), (1023, 40, 3), (1024, 300, 2), (1025, 7, 10 ** 6),
                         (4097, 5000, 4), (100003, 20000, 10 ** 4), (2 ** 20 + 5, 3000, 10 ** 5),
                         # potentials too many for LDS: register-resident potentials (<= 512 threads per workgroup) and, at 600k arcs, without
                         (50001, 17000, 3), (400003, 100001, 10 ** 4), (600000, 50000, 5),
                         (400003, 300000, 7)]:     # more nodes than bits in the patch bitmap of the resident grid: aliased bits
        pi_span = 10 ** 9 if width == 64 and span > 100 else span * 3
        a = _random_soa(rng, m_s, n, span, pi_span)
        block = int(rng.integers(1, max(2, min(m_s, 700))))
        eng = M.PivotEngine(n, len(a["src"]), m_s, rule=RULES[rule], optimized=optimized, int_width=width, block_size=block, flags=mode, vector_width=vw)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        next_arc = 0
        for it in range(12):
            f, e, c, na = _oracle_scan(rule, optimized, a, m_s, block, next_arc, vw)
            f2, e2, c2 = eng.find_entering()
            assert f2 == f, (m_s, it)
            if f:
                assert (e2, c2) == (e, c), (m_s, it, rule, optimized, vw, e2, e, c2, c, next_arc, block)
                if rule != O.RULE_BEST:
                    assert eng.next_arc == na, (m_s, it, eng.next_arc, na, next_arc, block, e)
                next_arc = na
            # random patches: 0..2 state writes; a potential list that is empty / small (inline) / large (staged)
            k_st = int(rng.integers(0, 3))
            arcs = rng.choice(len(a["src"]), size=min(k_st, len(a["src"])), replace=False).astype(np.int32)
            vals = rng.integers(-1, 2, len(arcs)).astype(np.int8)
            a["state"][arcs] = vals
            eng.patch_state(arcs, vals)
            k = int(rng.choice([0, 1, 5, 20, 21, 22, 96, 97, min(n, 3000), 4097, 4098, 6000]))   # every patch path of the resident grid
            k = min(k, n)
            nodes = rng.choice(n, size=k, replace=False).astype(np.int32)
            sigma = int(rng.integers(-span - 1, span + 2))
            a["pi"][nodes] += sigma
            eng.update_potential(nodes, sigma)
            if it == 5 and rule != O.RULE_BEST:
                next_arc = int(rng.integers(0, m_s + 1))      # includes next_arc == m_s (BlockSearchPivotOptimized.cs:63,102)
                eng.next_arc = next_arc
        assert np.array_equal(eng.download_pi(), a["pi"])
        assert np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])
        assert eng.check_reduced_costs() == (0, -1)      # RC layout: every arc's stored reduced cost is cost + pi[source] - pi[target] (0 arcs off; trivially so elsewhere)
        st = eng.stats()
        assert st["searches"] == 12 and st["resident"] == (0 if mode == M.ENGINE_DISPATCH else 1)
        if mode not in CAND_MODES:
            assert st["candidates"] == 0
        if rc_resident:
            assert st["rc_layout"] == 1 and st["resident"] == 1 and st["resident_requests"] + st["host_decided"] >= 12 and st["scan_bytes_read"] == 9 * m_s
            assert st["candidates"] == int(mode == 0 and rule == O.RULE_BEST and 2 * m_s <= 24 * n)
            assert st["candidates"] or st["resident_requests"] >= 12
        elif rc_layout:
            # (short lists ride in the scan's arguments only while their arc lists fit: with hundreds of arcs per node that is a matter of the draw)
            assert st["rc_layout"] == 1 and (st["update_launches"] > 0 or n < 100) and (st["inline_updates"] > 0 or 2 * m_s > 400 * n) and st["scan_bytes_read"] == 9 * m_s
        elif mode == M.ENGINE_DISPATCH:
            assert (st["inline_updates"] > 0 and st["update_launches"] > 0) or n < 97
        elif not rc_layout and mode in CAND_MODES and rule == O.RULE_BEST and m_s <= 1 << 20 and 2 * m_s <= 24 * n:
            assert st["candidates"] == 1 and st["resident_requests"] + st["host_decided"] >= 12
        else:
            assert st["resident_requests"] >= 12 and st["update_launches"] <= 1    # the patches queued after the last search


def test_scan_edge_cases():
    # empty search range, nothing eligible, everything in the basis
    eng = M.PivotEngine(3, 4, 0, rule=M.PivotRule.BestEligible)
    eng.upload(np.zeros(4, np.int32), np.zeros(4, np.int32), np.zeros(4, np.int64), np.zeros(4, np.int8), np.zeros(3, np.int64))
    assert eng.find_entering() == (False, -1, 0)
    for rule in RULES.values():
        eng = M.PivotEngine(4, 9, 9, rule=rule)
        src = np.arange(9, dtype=np.int32) % 4
        tgt = (src + 1) % 4
        eng.upload(src, tgt, np.full(9, 5, np.int64), np.ones(9, np.int8), np.zeros(4, np.int64))   # all rc = +5
        assert eng.find_entering()[0] is False
        eng.upload(src, tgt, np.full(9, -5, np.int64), np.zeros(9, np.int8), np.zeros(4, np.int64))  # all tree arcs
        assert eng.find_entering()[0] is False
        eng.upload(src, tgt, np.full(9, -5, np.int64), -np.ones(9, np.int8), np.zeros(4, np.int64))  # upper, rc = +5
        assert eng.find_entering()[0] is False
    # all reduced costs equal: the lowest arc index wins (strict '<', NetworkSimplex.cs:1653)
    eng = M.PivotEngine(2, 5000, 5000, rule=M.PivotRule.BestEligible)
    eng.upload(np.zeros(5000, np.int32), np.ones(5000, np.int32), np.full(5000, -7, np.int64), np.ones(5000, np.int8), np.zeros(2, np.int64))
    assert eng.find_entering() == (True, 0, -7)
    eng.patch_state([0, 1, 2], [0, 0, -1])
    assert eng.find_entering() == (True, 3, -7)
    # int64 values that need all 64 bits
    big = np.int64(3) << 60
    eng = M.PivotEngine(2, 2, 2, rule=M.PivotRule.BestEligible)
    eng.upload([0, 1], [1, 0], [-big, -big - 1], [1, 1], [0, 0])
    assert eng.find_entering() == (True, 1, int(-big - 1))
    with pytest.raises(M.McfError) as ei:      # int32 engine refuses what does not fit
        e32 = M.PivotEngine(2, 2, 2, int_width=32)
        e32.upload([0, 1], [1, 0], [-big, 1], [1, 1], [0, 0])
    assert ei.value.code == L.ERR_OVERFLOW
    with pytest.raises(M.McfError) as ei:
        eng.upload([0, 7], [1, 0], [1, 1], [1, 1], [0, 0])     # end point outside the node range
    assert ei.value.code == L.ERR_INVALID


def _solve_both(p, sem, rule, int_width=0, flags=0, supply_type=O.GEQ, block_size=0, auto=True, config=None, vw=4):
    """auto: the reference's auto-configuration (its default, NetworkSimplex.cs:90) on both sides; config: dict of OptimizationConfig
    fields given to SetOptimizationConfig on both sides (switches auto-configuration off, NetworkSimplex.cs:557-561)."""
    o = O.Oracle(p, sem, rule, supply_type=supply_type, block_size=block_size, auto_config=auto and config is None, config=config, vector_width=vw)
    st_o, tr_o = o.solve(trace_cap=4_000_000)
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
    ns.set_pivot_rule(RULES[rule]).enable_optimized_pivot(sem == O.SEM_CSHARP_OPT).set_supply_type(supply_type).set_vector_width(vw)
    if config is not None:
        ns.set_optimization_config(M.block_config(**config))
    elif not auto:
        ns.set_auto_configuration(False)
    ns.set_device(0, int_width, block_size, flags).record_trace(4_000_000)
    st = ns.solve()
    return o, st_o, tr_o, ns, st


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["netgen_8_08a", "netgen_8_10a", "transport_40x30", "circulation_100_0_10", "assignment_50x50",
                                  "SimpleProblemIllustration2NonSparse", "AURV19V6", "grid_5x5", "star_graph"])
def test_solve_is_pivot_for_pivot_identical(name, mode, monkeypatch):
    """Same entering arc at every pivot, hence same flows and potentials, for both C# flavours of every rule."""
    rc_layout = isinstance(mode, str)
    rc_resident = rc_layout and mode != "rc"
    mode = _mode_flags(mode, monkeypatch)
    p = load(name)
    # the optimized Block Search as the reference runs it on x64 (Vector<long>.Count = 4), on NEON (2) and without hardware vectors (0)
    for sem, rule, vw in [(O.SEM_CSHARP_OPT, O.RULE_BLOCK, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 0), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 2), (O.SEM_CSHARP, O.RULE_BLOCK, 4),
                          (O.SEM_CSHARP_OPT, O.RULE_BEST, 4), (O.SEM_CSHARP, O.RULE_BEST, 4), (O.SEM_CSHARP_OPT, O.RULE_FIRST, 4), (O.SEM_CSHARP, O.RULE_FIRST, 4)]:
        if name == "AURV19V6" and rule == O.RULE_FIRST:
            continue        # 117k pivots, nothing new
        o, st_o, tr_o, ns, st = _solve_both(p, sem, rule, flags=mode, vw=vw)
        tr = ns.trace()
        assert st == st_o == O.OPTIMAL
        assert len(tr) == len(tr_o) == o.n_pivots, (name, sem, rule, vw, len(tr), len(tr_o))
        assert np.array_equal(tr, tr_o), (name, sem, rule, vw, int(np.argmax(tr != tr_o)))
        assert ns.get_total_cost() == o.total_cost
        assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
        m = ns.get_metrics()
        assert m["iterations"] == o.n_pivots and m["search_arc_num"] == o.search_arc_num
        assert rule != O.RULE_BLOCK or m["block_size"] == o.initial_block_size
        # the rest of SolverMetrics (OptimizationTypes.cs:53-59): only the plain BlockSearchPivot counts arcs and reports block sizes
        assert m["total_arcs_checked"] == o.arcs_checked and m["config_flags"] == (o.config_flags if sem == O.SEM_CSHARP else m["config_flags"])
        if sem == O.SEM_CSHARP and rule == O.RULE_BLOCK:
            assert (m["initial_block_size"], m["final_block_size"]) == (o.initial_block_size, o.block_size)
            assert m["reference_selects_cached_pivot"] == int(o.would_cache)
        else:
            assert (m["initial_block_size"], m["final_block_size"], m["total_arcs_checked"]) == (0, 0, 0)
        assert m["engine"]["resident"] == (0 if mode == M.ENGINE_DISPATCH else 1)
        if not rc_layout and mode in CAND_MODES and rule == O.RULE_BEST and 2 * o.search_arc_num <= 24 * (p.n + 1):
            assert m["engine"]["candidates"] == 1 and m["engine"]["host_decided"] + m["engine"]["resident_requests"] >= o.n_pivots


@pytest.mark.parametrize("name,path,want", fixtures(), ids=[f[0] for f in fixtures()])
def test_every_fixture_reaches_the_sol_cost(name, path, want):
    """The reference's fixture sweep (PerformanceComparisonReport.cs:253-268): GetTotalCost() == .sol 's' line, through
    EnableOptimizedPivot(true) + Block Search, plus the validator's optimality certificate."""
    p = load(path)
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply).enable_optimized_pivot(True)
    assert ns.solve() == M.SolverStatus.Optimal
    assert ns.get_total_cost() == want
    validate_solution(p, ns.flows(), ns.potentials())
    if p.m <= 10000:
        nb = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
        nb.set_pivot_rule(M.PivotRule.BestEligible)
        assert nb.solve() == M.SolverStatus.Optimal and nb.get_total_cost() == want


@pytest.mark.parametrize("kat", K.CSHARP_KATS, ids=[k[0] for k in K.CSHARP_KATS])
def test_csharp_unit_test_answers_on_gpu(kat):
    name, d, status, cost, flows = kat
    p = problem_from_dict(d)
    seen = set()
    for optimized in (False, True):                 # OptimizationTests.cs:14-69: same status, cost AND flows
        for rule in RULES.values():                 # OptimizationTests.cs:75-120: every rule validates
            ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
            ns.set_pivot_rule(rule).enable_optimized_pivot(optimized)
            assert ns.solve() == status and ns.status == status
            if status != M.SolverStatus.Optimal:
                with pytest.raises(M.McfError):
                    ns.get_flow(0)
                continue
            if cost is not None:
                assert ns.get_total_cost() == cost
            if flows is not None:
                assert ns.flows().tolist() == flows
            validate_solution(p, ns.flows(), ns.potentials())
            if rule == M.PivotRule.BlockSearch:
                seen.add((ns.get_total_cost(), tuple(ns.flows().tolist())))
    assert len(seen) <= 1


@pytest.mark.parametrize("case", K.LEMON_TABLE, ids=[f"lemon-{c[0]}" for c in K.LEMON_TABLE])
def test_status_parity_on_the_lemon_table_networks(case):
    """LEMON's 12-node / negative-cost networks through the C# semantics: whatever the C# port answers (including its
    documented deviations D7/D9: unbounded detection, LEQ/GEQ slack) the GPU path answers too, pivot for pivot."""
    cid, d, stype, _, _ = case
    p = problem_from_dict(d)
    for sem, rule in [(O.SEM_CSHARP_OPT, O.RULE_BLOCK), (O.SEM_CSHARP, O.RULE_BEST), (O.SEM_CSHARP_OPT, O.RULE_FIRST)]:
        o, st_o, tr_o, ns, st = _solve_both(p, sem, rule, supply_type=stype)
        assert st == st_o, (cid, sem, rule, st, st_o)
        assert np.array_equal(ns.trace(), tr_o[: len(ns.trace())]) and len(ns.trace()) in (len(tr_o), len(tr_o) - 0)
        if st == M.SolverStatus.Optimal:
            assert ns.get_total_cost() == o.total_cost and np.array_equal(ns.flows(), o.flow())


def test_degenerate_graphs():
    # no arcs at all: only the n root links are searched
    ns = M.NetworkSimplex(3, np.zeros(0, np.int32), np.zeros(0, np.int32)).set_problem(supply=[0, 0, 0])
    assert ns.solve() == M.SolverStatus.Optimal and ns.get_total_cost() == 0
    ns = M.NetworkSimplex(2, np.zeros(0, np.int32), np.zeros(0, np.int32)).set_problem(supply=[1, -1])
    assert ns.solve() == M.SolverStatus.Infeasible            # nothing connects the two nodes
    # Solve() is single-shot (D11)
    with pytest.raises(M.McfError) as ei:
        ns.solve()
    assert ei.value.code == L.ERR_STATE
    # self loop and parallel arcs
    ns = M.NetworkSimplex(2, [0, 0, 0, 1], [1, 1, 0, 1]).set_problem(upper=[5, 5, 9, 9], cost=[3, 2, -1, 4], supply=[7, -7])
    p = O.Problem(2, 4, [0, 0, 0, 1], [1, 1, 0, 1], [0] * 4, [5, 5, 9, 9], [3, 2, -1, 4], [7, -7])
    o = O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK)
    st_o, _ = o.solve()
    assert ns.solve() == st_o
    if st_o == 1:
        assert ns.get_total_cost() == o.total_cost and np.array_equal(ns.flows(), o.flow())


def test_int32_and_int64_device_paths_agree_and_overflow_is_caught():
    g = M.netgen_like(13502460, 3000, 12000, 50, 50)
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    traces = []
    for w in (32, 64, 0):
        o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK, int_width=w)
        assert st == st_o == O.OPTIMAL and np.array_equal(ns.trace(), tr_o)
        traces.append(ns.trace())
        if w == 0:
            assert ns.get_metrics()["int_width"] == 32      # (max|c|+1)*n*4 < 2^31
    assert np.array_equal(traces[0], traces[1])
    pa = load("AURV19V6")                                    # needs int64 ("never int")
    ns = M.NetworkSimplex(pa.n, pa.src, pa.tgt).set_problem(pa.lower, pa.upper, pa.cost, pa.supply).set_device(0, 32)
    with pytest.raises(M.McfError) as ei:
        ns.solve()
    assert ei.value.code == L.ERR_OVERFLOW
    ns = M.NetworkSimplex(pa.n, pa.src, pa.tgt).set_problem(pa.lower, pa.upper, pa.cost, pa.supply)
    assert ns.solve() == 1 and ns.get_metrics()["int_width"] == 64


def test_staged_update_path_equals_inline_path():
    p = load("netgen_8_10a")
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, flags=M.ENGINE_NO_INLINE_UPDATE)
    assert st == 1 and np.array_equal(ns.trace(), tr_o)
    e = ns.get_metrics()["engine"]
    assert e["inline_updates"] == 0 and e["update_launches"] > 1000
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, flags=M.ENGINE_SAMPLE_KERNEL_TIME)
    e = ns.get_metrics()["engine"]
    assert np.array_equal(ns.trace(), tr_o) and e["inline_updates"] > 1000
    assert e["resident"] or (e["timed_scans"] > 10 and e["timed_scan_ns"] > 0)


@pytest.mark.parametrize("rule,optimized,vw", RULE_CASES)
def test_sharded_engines_resolve_like_one_engine(rule, optimized, vw):
    """Arc shards on separate engines + the MINLOC resolve step give the single-engine answer (same device; the RCCL
    all-gather only moves the 16-byte records)."""
    rng = np.random.default_rng(99 + rule)
    m_s, n, world = 50021, 3000, 3
    a = _random_soa(rng, m_s, n, 5, 12, extra=0)
    block = 173
    one = M.PivotEngine(n, m_s, m_s, rule=RULES[rule], optimized=optimized, block_size=block, vector_width=vw)
    one.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    shards = []
    for r in range(world):
        e = M.PivotEngine(n, m_s, m_s, rule=RULES[rule], optimized=optimized, block_size=block, shard=M.shard_range(m_s, r, world), vector_width=vw)
        e.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        shards.append(e)
    for it in range(10):
        na0 = one.next_arc
        want = one.find_entering()
        assert want == _oracle_scan(rule, optimized, a, m_s, block, na0, vw)[:3]
        cands = [e.find_entering_local() for e in shards]
        got = [e.resolve(cands) for e in shards]
        assert all(g == want for g in got), (it, want, got)
        assert all(e.next_arc == one.next_arc for e in shards)
        if want[0]:
            arcs = np.array([want[1]], np.int32); vals = np.array([0], np.int8)
            nodes = rng.choice(n, size=40, replace=False).astype(np.int32)
            a["state"][arcs] = vals
            a["pi"][nodes] += -3
            for e in [one] + shards:
                e.patch_state(arcs, vals)
                e.update_potential(nodes, -3)


@pytest.mark.parametrize("layout", ["lds-potentials", "gathering", "rc"])
def test_sharded_solve_through_the_rccl_exchange(layout, monkeypatch):
    """mcf_ns_set_sharding + mcf_engine_find_entering_sharded on the one GPU of this box (world size 1: the shard engine stays resident with
    its candidate cache, its candidate goes through ncclAllGather on a stream of its own and the MINLOC still runs); more ranks are covered by
    tests/test_sharded_gloo.py and the same-device shard tests.  Every rule, on each of the three layouts (potentials in LDS, in registers, RC)."""
    if layout == "rc":
        monkeypatch.setenv("MCF_HIP_RC", "1")
    if layout == "lds-potentials":
        p = load("netgen_8_10a")
    else:
        g = M.netgen_like(7, 20_000, 60_000, 100, 100)
        p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    for sem, rule, vw in [(O.SEM_CSHARP_OPT, O.RULE_BEST, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 0), (O.SEM_CSHARP, O.RULE_BLOCK, 4),
                          (O.SEM_CSHARP_OPT, O.RULE_FIRST, 4)]:
        if layout != "lds-potentials" and rule == O.RULE_FIRST:
            continue        # hundreds of thousands of pivots, nothing new
        o = O.Oracle(p, sem, rule, auto_config=True, vector_width=vw)
        st_o, tr_o = o.solve(trace_cap=1 << 22)
        ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
        ns.set_pivot_rule(RULES[rule]).enable_optimized_pivot(sem == O.SEM_CSHARP_OPT).set_vector_width(vw).record_trace(1 << 22)
        ns.set_sharding(M.comm_unique_id(), 0, 1)
        assert ns.solve() == st_o == 1
        assert np.array_equal(ns.trace(), tr_o) and ns.get_total_cost() == o.total_cost
        e = ns.get_metrics()["engine"]
        # the engine keeps its resident grid and, for Best Eligible on these sparse graphs, its candidate cache: the collective runs on a
        # stream of its own over records in pinned host memory
        assert e["resident"] == 1 and e["comm_ranks"] == 1 and e["resident_requests"] > 0
        assert e["candidates"] == (1 if rule == O.RULE_BEST else 0) and (rule != O.RULE_BEST or e["host_decided"] > 0)
        assert e["rc_layout"] == (1 if layout == "rc" else 0)


def test_engines_kept_alive_do_not_share_a_hardware_queue_with_a_resident_grid():
    """HIP spreads the streams of one priority over four hardware queues and a resident grid never leaves its queue: with a handful of engines
    alive in the process, a grid on its engine's ordinary stream ended up on one queue with another engine's work -- the collective of a sharded
    solve then waited for the grid's idle timeout at every pivot (33 ms per pivot in a bench run that kept its timed solvers alive).  Grids run
    on a stream of their own priority that lives only as long as the grid; the collective has another."""
    g = M.netgen_like(7, 20_000, 60_000, 100, 100)
    idle = []
    for k in range(6):             # engines that have searched, are parked and stay alive (their ordinary streams keep their queues)
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_pivot_limit(50)
        assert ns.solve() == M.SolverStatus.NotSolved
        idle.append(ns)
    ref = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_pivot_limit(400).record_trace(400)
    ref.solve()
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_pivot_limit(400).record_trace(400)
    ns.set_sharding(M.comm_unique_id(), 0, 1).prepare()
    t0 = time.perf_counter()
    ns.solve()
    dt = time.perf_counter() - t0
    e = ns.get_metrics()["engine"]
    assert e["resident"] == 1 and e["comm_ranks"] == 1
    assert np.array_equal(ns.trace(), ref.trace())
    assert dt < 4.0, f"{dt:.2f} s for 400 pivots through the collective: is it waiting behind a resident grid?"       # 10 - 20 ms when not (13 s when it was)
    del idle


def test_solvers_that_share_a_device_by_workgroups_take_the_same_pivots():
    """mcf_ns_set_device_share: K solvers in flight on one device, each with a resident grid of 256 / K workgroups on CUs of its own.  With so few
    workgroups the arcs of an instance this size no longer fit the registers and the engine keeps reduced costs per arc instead -- same pivots.
    (This process runs with HIP's default of four hardware queues: four grids are resident, and that is what the test asks for.)"""
    import threading
    g = M.netgen_like(13502460, 30_000, 100_000, 170, 170)
    ref = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).record_trace(1 << 18)
    assert ref.solve() == M.SolverStatus.Optimal
    want = ref.trace().copy()
    del ref
    K = 4
    cs = [M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device_share(256 // K).record_trace(1 << 18).prepare()
          for _ in range(K)]
    th = [threading.Thread(target=c.solve) for c in cs]
    [t.start() for t in th]
    [t.join() for t in th]
    for c in cs:
        assert c.status == M.SolverStatus.Optimal and np.array_equal(c.trace(), want)
        e = c.get_metrics()["engine"]
        assert e["resident"] == 1 and e["scan_workgroups"] == 256 // K and c.check_reduced_costs() == 0
    with pytest.raises(M.McfError):
        cs[0].set_device_share(3)


def test_pivot_limit_stops_early():
    p = load("netgen_8_10a")
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply).set_pivot_limit(100)
    assert ns.solve() == M.SolverStatus.NotSolved and ns.get_metrics()["iterations"] == 100


def test_full_size_configs_certified_optimal():
    """BASELINE.json configs 2-4 at full size.  The oracle's Best-Eligible would take minutes here, so parity is checked
    through size-independent properties: the validator's optimality certificate (primal = dual, complementary slackness,
    conservation) and equality of the optimal cost with the oracle's Block-Search solve of the same instance."""
    cases = [
        ("config2", M.netgen_like(13502460, 10_000, 30_000, 100, 100), M.PivotRule.BlockSearch, 32),
        ("config3", M.netgen_like(13502460, 100_000, 300_000, 316, 316), M.PivotRule.BestEligible, 64),
        ("config4", M.assignment(42, 1000, 1, 100), M.PivotRule.BestEligible, 64),
    ]
    for name, g, rule, width in cases:
        p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True).set_device(0, width)
        assert ns.solve() == M.SolverStatus.Optimal, name
        cost = validate_solution(p, ns.flows(), ns.potentials())
        assert cost == ns.get_total_cost()
        o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK)
        assert o.solve()[0] == O.OPTIMAL and o.total_cost == cost, name
        m = ns.get_metrics()
        assert m["int_width"] == width and m["iterations"] > 0


def test_candidate_cache_answers_most_searches_and_changes_nothing():
    """MCF_ENGINE_CANDIDATES on a mid-size NETGEN-like solve: identical pivot sequence, most searches answered without a
    device request."""
    g = M.netgen_like(13502460, 20_000, 70_000, 140, 140)
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, flags=0)
    assert st == st_o == 1 and np.array_equal(ns.trace(), tr_o)
    assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
    e = ns.get_metrics()["engine"]
    assert e["candidates"] == 1 and e["host_decided"] > 2 * e["resident_requests"], e
    # potential updates given as += sigma (the C# host's call) keep the cache consistent too
    rng = np.random.default_rng(11)
    m_s, n = 50_000, 9_000
    a = _random_soa(rng, m_s, n, 40, 400, extra=0)
    eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, flags=0)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for it in range(300):
        f, e_, c = eng.find_entering()
        assert (f, e_, c) == O.scan_best(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"]), it
        if not f:
            break
        a["state"][e_] = 0
        eng.patch_state([e_], [0])
        nodes = rng.choice(n, size=int(rng.choice([1, 1, 2, 3, 30, 700])), replace=False).astype(np.int32)
        sigma = int(rng.integers(-5, 6))
        a["pi"][nodes] += sigma
        if it % 2:
            eng.update_potential(nodes, sigma)
        else:
            eng.set_potential(nodes, a["pi"][nodes])
    # several long lists in a row without a search in between (repeats are squeezed out before the list can outgrow the mailbox)
    for rep in range(7):
        nodes = rng.choice(n, size=8000, replace=False).astype(np.int32)
        a["pi"][nodes] += rep + 1
        if rep % 2:
            eng.update_potential(nodes, rep + 1)
        else:
            eng.set_potential(nodes, a["pi"][nodes])
    assert eng.find_entering() == O.scan_best(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"])
    st2 = eng.stats()
    assert st2["host_decided"] > 50 and st2["resident_requests"] > 5, st2
    assert np.array_equal(eng.download_pi(), a["pi"]) and np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])


def test_sigma_and_value_updates_can_be_mixed():
    rng = np.random.default_rng(3)
    m_s, n = 30000, 5000
    a = _random_soa(rng, m_s, n, 50, 500, extra=0)
    eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for it in range(6):
        nodes = rng.choice(n, size=int(rng.integers(1, 900)), replace=False).astype(np.int32)
        if it % 2 == 0:
            a["pi"][nodes] += 7
            eng.set_potential(nodes, a["pi"][nodes])          # values (what the C++ host driver sends)
        else:
            a["pi"][nodes] -= 3
            eng.update_potential(nodes, -3)                   # += sigma (what a C# host sends)
        assert eng.find_entering() == O.scan_best(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"])
    assert np.array_equal(eng.download_pi(), a["pi"])


def test_bench_scan_reports_sane_durations():
    rng = np.random.default_rng(5)
    m_s, n = 1 << 20, 100_000
    a = _random_soa(rng, m_s, n, 10 ** 4, 10 ** 9, extra=0)
    eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    avg, mn = eng.bench_scan(reps=10)
    assert 500 < mn <= avg < 5e6
    f, e, c = eng.find_entering()
    assert (f, e, c) == O.scan_best(m_s, a["state"], a["cost"], a["src"], a["tgt"], a["pi"])


def _random_problem(rng, n, m, supply_kind):
    src = rng.integers(0, n, m).astype(np.int32)
    tgt = rng.integers(0, n, m).astype(np.int32)
    lower = np.where(rng.random(m) < 0.15, rng.integers(0, 4, m), 0).astype(np.int64)
    upper = (lower + rng.integers(0, 12, m)).astype(np.int64)
    upper[rng.random(m) < 0.1] = O.INF_CAP
    cost = rng.integers(-6 if supply_kind == "negative" else 0, 20, m).astype(np.int64)
    supply = np.zeros(n, np.int64)
    k = max(1, n // 4)
    s = rng.integers(1, 9, k)
    supply[rng.choice(n, k, replace=False)] += s
    supply[rng.choice(n, k, replace=False)] -= rng.permutation(s)
    if supply_kind == "excess":
        supply[rng.integers(0, n)] += 3          # unbalanced: whatever the C# solver makes of it, both sides must agree
    return O.Problem(n, m, src, tgt, lower, upper, cost, supply)


@pytest.mark.parametrize("seed", range(12))
def test_random_small_networks_all_outcomes(seed):
    """Small random instances with lower bounds, infinite capacities, negative costs, self loops, parallel arcs, LEQ and GEQ:
    status (Optimal / Infeasible / Unbounded), pivot sequence, cost, flows and potentials equal the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    seen = set()
    for trial in range(10):
        n = int(rng.integers(2, 40))
        m = int(rng.integers(1, 160))
        kind = ["balanced", "negative", "excess"][trial % 3]
        p = _random_problem(rng, n, m, kind)
        stype = O.LEQ if trial % 4 == 3 else O.GEQ
        sem, rule = [(O.SEM_CSHARP_OPT, O.RULE_BLOCK), (O.SEM_CSHARP, O.RULE_BEST), (O.SEM_CSHARP_OPT, O.RULE_FIRST), (O.SEM_CSHARP, O.RULE_BLOCK)][trial % 4]
        flags = [M.ENGINE_NO_CANDIDATES, M.ENGINE_DISPATCH, 0][trial % 3]
        o, st_o, tr_o, ns, st = _solve_both(p, sem, rule, supply_type=stype, flags=flags)
        assert st == st_o, (seed, trial, st, st_o)
        assert np.array_equal(ns.trace(), tr_o[: len(ns.trace())])
        seen.add(st)
        if st == M.SolverStatus.Optimal:
            assert len(ns.trace()) == len(tr_o) and ns.get_total_cost() == o.total_cost
            assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
    assert seen       # at least something ran


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [pytest.param(M.ENGINE_DISPATCH, id="dispatch"), pytest.param(0, id="resident")])
@pytest.mark.parametrize("bucket", [700, 4096])
def test_bucketed_layout_keeps_every_tie_break(mode, bucket, monkeypatch):
    """Arcs stored sorted by target-node range (MCF_HIP_BUCKET_NODES forces it on small inputs): same arc, same reduced cost, same state
    and potentials as the oracle, with tiny cost ranges so that ties between arcs of different ranges decide most searches."""
    monkeypatch.setenv("MCF_HIP_BUCKET_NODES", str(bucket))
    if mode == 0:
        monkeypatch.setenv("MCF_HIP_RESIDENT", "1")      # above 1M arcs the engine would pick one dispatch per search by itself
    rng = np.random.default_rng(77 + bucket)
    for m_s, n, span, width in [(50001, 17000, 2, 64), (400003, 100001, 3, 64), (300000, 20000, 10 ** 4, 32), (1_300_000, 30000, 2, 64)]:
        a = _random_soa(rng, m_s, n, span, span * 3)
        eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BestEligible, optimized=True, int_width=width, flags=mode)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        for it in range(10):
            f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
            f2, e2, c2 = eng.find_entering()
            assert f2 == f and (not f or (e2, c2) == (e, c)), (m_s, it, e2, e, c2, c)
            arcs = rng.choice(m_s, size=int(rng.integers(0, 3)), replace=False).astype(np.int32)
            vals = rng.integers(-1, 2, len(arcs)).astype(np.int8)
            a["state"][arcs] = vals
            eng.patch_state(arcs, vals)
            k = int(rng.choice([0, 1, 5, 97, 3000]))
            nodes = rng.choice(n, size=k, replace=False).astype(np.int32)
            sigma = int(rng.integers(-span - 1, span + 2))
            a["pi"][nodes] += sigma
            eng.update_potential(nodes, sigma)
        assert np.array_equal(eng.download_pi(), a["pi"])
        assert np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])


@pytest.mark.gpu
def test_bucketed_layout_solves_pivot_for_pivot(monkeypatch):
    monkeypatch.setenv("MCF_HIP_BUCKET_NODES", "2048")
    p = load("netgen_8_14a")           # 16384 nodes: the potentials do not fit LDS, so the tile loop (and with it the bucketed layout) is used
    for flags in (M.ENGINE_DISPATCH,):
        o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, flags=flags)
        assert st == st_o == O.OPTIMAL and np.array_equal(ns.trace(), tr_o)
        assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())


@pytest.mark.gpu
def test_bucketed_layout_on_arc_shards(monkeypatch):
    """Every shard stores ITS arcs in bucketed order (positions are shard-local, ids global): the resolved answer is still the oracle's."""
    monkeypatch.setenv("MCF_HIP_BUCKET_NODES", "900")
    rng = np.random.default_rng(4242)
    m_s, n, world = 200003, 20000, 3
    a = _random_soa(rng, m_s, n, 2, 6, extra=0)
    shards = []
    for r in range(world):
        e = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, optimized=True, shard=M.shard_range(m_s, r, world), flags=M.ENGINE_NO_CANDIDATES)
        e.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        shards.append(e)
    for it in range(10):
        f, arc, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        cands = [e.find_entering_local() for e in shards]
        got = [e.resolve(cands) for e in shards]
        assert all(g[0] == f and (not f or (g[1], g[2]) == (arc, c)) for g in got), (it, got, arc, c)
        arcs = rng.choice(m_s, size=3, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 3).astype(np.int8)
        a["state"][arcs] = vals
        nodes = rng.choice(n, size=int(rng.choice([1, 40, 500])), replace=False).astype(np.int32)
        a["pi"][nodes] -= 2
        for e in shards:
            e.patch_state(arcs, vals)
            e.update_potential(nodes, -2)
    full = np.zeros(m_s, np.int8)
    for e in shards:
        st = e.download_state()
        b, en = M.shard_range(m_s, shards.index(e), world)
        full[b:en] = st[b:en]
    assert np.array_equal(full, a["state"][:m_s])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["registers", "rc-lds", "rc-stream"])
def test_arc_shards_keep_their_own_candidate_caches(layout, monkeypatch):
    """Best Eligible on a sparse graph: every shard engine keeps a candidate cache over ITS arcs (list + heap + adjacency of [begin, end)),
    answers mcf_engine_search_end_local from the host when it can and asks its own grid when it cannot; the holders' MINLOC of the shard
    candidates is the oracle's entering arc after every kind of change (single nodes, lists of a pivot's size, lists too big to evaluate,
    state writes inside and outside a shard)."""
    if layout != "registers":
        monkeypatch.setenv("MCF_HIP_RC", "1")
        if layout == "rc-stream":
            monkeypatch.setenv("MCF_HIP_RC_LDS", "0")
    rng = np.random.default_rng(77)
    m_s, n, world = 180_007, 40_000, 3
    a = _random_soa(rng, m_s, n, 40, 300, extra=0)
    shards = []
    for r in range(world):
        e = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, optimized=True, shard=M.shard_range(m_s, r, world), resident_workgroups=64)
        e.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        assert e.stats()["candidates"] == 1 and e.stats()["resident"] == 1
        shards.append(e)
    for it in range(400):
        f, arc, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        for e in shards:
            e.search_begin()
        cands = [e.search_end_local() for e in shards]
        got = [e.resolve(cands) for e in shards]
        assert all(g[0] == f and (not f or (g[1], g[2]) == (arc, c)) for g in got), (it, got, arc, c)
        if not f:
            break
        # a pivot-like change: the entering arc joins the basis, another arc leaves it, a subtree's potentials move by the entering arc's reduced cost
        leave = int(rng.integers(0, m_s))
        arcs = np.array([arc, leave], np.int32); vals = np.array([0, int(rng.choice([-1, 1]))], np.int8)
        a["state"][arcs] = vals
        k = int(rng.choice([1, 1, 1, 2, 5, 30, 90, 400, 5000]))
        nodes = rng.choice(n, size=k, replace=False).astype(np.int32)
        sigma = int(c) if it % 3 else -int(c)
        a["pi"][nodes] += sigma
        for e in shards:
            e.patch_state(arcs, vals)
            e.shift_potential(nodes, a["pi"][nodes], sigma)
    st = [e.stats() for e in shards]
    assert sum(x["host_decided"] for x in st) > 0 and all(x["host_decided"] + x["resident_requests"] >= it for x in st), st
    for r, e in enumerate(shards):
        assert np.array_equal(e.download_pi(), a["pi"])
        b, en = M.shard_range(m_s, r, world)
        assert np.array_equal(e.download_state()[b:en], a["state"][b:en])


RC_MODES = [pytest.param("rc", id="rc-layout"), pytest.param("rc-resident", id="rc-resident-lds"), pytest.param("rc-stream", id="rc-resident-stream"),
            pytest.param("rc-cand", id="rc-candidates-lds"), pytest.param("rc-cand-stream", id="rc-candidates-stream")]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", RC_MODES + [pytest.param(0, id="candidates-shift-grid")])
def test_reload_of_the_bound_potentials_replaces_node_lists(mode, monkeypatch):
    """mcf_engine_bind_potentials + mcf_engine_reload_potentials (RC layout): after changes to a large part of the caller's array the engine is told
    'reload' instead of being handed the nodes; it copies the array by itself and computes every reduced cost again.  Mixed with ordinary lists
    and state writes, with further small lists between the reload and the search, in every RC engine mode: the oracle's answer every time."""
    flags = _mode_flags(mode, monkeypatch)
    if mode == 0:
        monkeypatch.setenv("MCF_HIP_SHIFT_RELOAD", "8192")       # the register-resident grid's own threshold (32768 nodes) is more than this graph has
    rng = np.random.default_rng(4711)
    m_s, n = 120_003, 25_000
    a = _random_soa(rng, m_s, n, 30, 200, extra=0)
    eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=flags)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    assert eng.reload_threshold() == 0                      # nothing bound yet: lists only
    eng.bind_potentials(a["pi"])
    lo = eng.reload_threshold()
    shift_grid = mode == 0         # the register-resident candidate grid: the reload is a copy + a grid-wide barrier + a gather inside the grid
    assert lo == (8192 if shift_grid else max(1024, n // 16) + 1)
    for it in range(40):
        f, arc, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        assert eng.find_entering() == (f, arc, c), it
        arcs = rng.choice(m_s, size=2, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 2).astype(np.int8)
        a["state"][arcs] = vals
        eng.patch_state(arcs, vals)
        kind = it % 4
        if kind == 0:                                        # an ordinary list
            nodes = rng.choice(n, size=int(rng.choice([1, 30, 2000])), replace=False).astype(np.int32)
            sigma = int(rng.integers(-9, 10))
            a["pi"][nodes] += sigma
            eng.shift_potential(nodes, a["pi"][nodes], sigma)
        else:                                                # a large part of the array changes in place; the engine only hears "reload"
            k = int(rng.choice([lo, n // 2, n - 1]))
            nodes = rng.choice(n, size=k, replace=False)
            a["pi"][nodes] += int(rng.integers(-9, 10))
            if kind == 2:                                    # ... after a list that the reload makes redundant
                few = rng.choice(n, size=5, replace=False).astype(np.int32)
                a["pi"][few] -= 1
                eng.shift_potential(few, a["pi"][few], -1)
                a["pi"][nodes] += 2
            eng.reload_potentials(k)
            if kind == 3:                                    # ... and a small list on top of it before the search
                few = rng.choice(n, size=3, replace=False).astype(np.int32)
                a["pi"][few] += 4
                eng.shift_potential(few, a["pi"][few], 4)
    f, arc, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
    assert eng.find_entering() == (f, arc, c)
    st = eng.stats()
    # a resident grid carries the reloads out itself (the workgroups copy the bound array, meet at a grid-wide barrier and compute their arcs'
    # reduced costs again); with one dispatch per search the array is copied and rc_init_kernel runs
    assert st["rc_layout"] == (0 if shift_grid else 1) and st["shift_grid"] == (1 if shift_grid else 0) and st["rc_recomputes"] + st["rc_reloads_in_grid"] >= 25
    assert st["rc_reloads_in_grid"] >= (25 if mode != "rc" else 0) and (mode != "rc" or st["rc_reloads_in_grid"] == 0)
    assert eng.check_reduced_costs() == (0, -1)
    assert np.array_equal(eng.download_pi(), a["pi"]) and np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])
    eng.bind_potentials(None)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", RC_MODES)
@pytest.mark.parametrize("name", ["netgen_8_10a", "AURV19V6"])
def test_solve_that_reloads_the_potentials_after_long_walks(name, mode, monkeypatch):
    """mcf_ns_solve over the RC layout with the reload threshold lowered to 48 nodes (default: a sixteenth of the nodes): every longer walk
    moves the potentials without writing a list and the engines reload _pi -- same pivots, flows and potentials as the oracle."""
    flags = _mode_flags(mode, monkeypatch)
    monkeypatch.setenv("MCF_HIP_RC_RECOMPUTE", "8")
    p = load(name)
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, int_width=64, flags=flags)      # reloads copy int64 potentials: 64-bit engines only
    assert st == st_o == O.OPTIMAL and np.array_equal(ns.trace(), tr_o)
    assert ns.get_total_cost() == o.total_cost
    assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
    m = ns.get_metrics()
    assert m["engine"]["rc_layout"] == 1 and m["engine"]["rc_recomputes"] + m["engine"]["rc_reloads_in_grid"] > 50
    assert mode == "rc" or m["engine"]["rc_reloads_in_grid"] > 50


@pytest.mark.gpu
def test_shift_grid_reloads_the_potentials_inside_the_grid(monkeypatch):
    """The register-resident candidate grid with its reload threshold lowered from 32768 to 48 nodes: every longer walk of the solve names no
    nodes, the workgroups copy the bound _pi, meet at their grid-wide barrier and gather their end points again (cmd 3) -- the oracle's pivots,
    flows and potentials; with relabellings in between (the bound array is permuted in place) and without."""
    monkeypatch.setenv("MCF_HIP_SHIFT_RELOAD", "48")
    g = M.netgen_like(7, 20_000, 60_000, 100, 100)
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    for renumber in ("0", "0.5"):
        monkeypatch.setenv("MCF_NS_RENUMBER", renumber)
        o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, int_width=64, flags=0)
        assert st == st_o == O.OPTIMAL
        assert np.array_equal(ns.trace(), tr_o), int(np.argmax(ns.trace()[: len(tr_o)] != tr_o[: len(ns.trace())]))
        assert ns.get_total_cost() == o.total_cost
        assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
        e = ns.get_metrics()["engine"]
        assert e["shift_grid"] == 1 and e["rc_reloads_in_grid"] > 50, e["rc_reloads_in_grid"]
        assert (e["renumberings"] > 0) == (renumber != "0")


@pytest.mark.gpu
def test_register_resident_potentials_pivot_for_pivot():
    """A solve bigger than the bundled fixtures (20k nodes / 60k arcs: the potentials do not fit LDS), so the resident grid keeps the end
    points' potentials in registers and sees every kind of patch list (one node ... more than 4096) in real proportions."""
    g = M.netgen_like(7, 20_000, 60_000, 100, 100)
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    for rule, flags in ((O.RULE_BEST, M.ENGINE_NO_CANDIDATES), (O.RULE_BLOCK, 0), (O.RULE_BEST, 0)):
        o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, rule, flags=flags)
        assert st == st_o == O.OPTIMAL
        assert np.array_equal(ns.trace(), tr_o), int(np.argmax(ns.trace()[: len(tr_o)] != tr_o[: len(ns.trace())]))
        assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
        m = ns.get_metrics()
        assert m["engine"]["resident"] == 1 and m["engine"]["scan_threads"] <= 512
        assert m["engine"]["candidates"] == (1 if rule == O.RULE_BEST and flags == 0 else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("m_s,n", [(400003, 100001), (60001, 16000), (1_300_000, 40000)])
def test_potential_lists_that_arrive_in_pieces(mode, m_s, n, monkeypatch):
    """mcf_engine_append_potential: the list of one pivot handed over in several calls (disjoint nodes); in resident mode the complete lines
    travel at once (apply posts) and the search finishes the list.  Every engine mode, potentials in registers / LDS / memory."""
    rc_layout = isinstance(mode, str)
    rc_resident = rc_layout and mode != "rc"
    mode = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(31 + m_s)
    a = _random_soa(rng, m_s, n, 3, 9)
    eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=mode)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for it in range(8):
        f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        f2, e2, c2 = eng.find_entering()
        assert f2 == f and (not f or (e2, c2) == (e, c)), (it, e2, e, c2, c)
        arcs = rng.choice(m_s, size=2, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 2).astype(np.int8)
        a["state"][arcs] = vals
        eng.patch_state(arcs, vals)
        total = int(rng.choice([3, 5200, 11000, min(n - 1, 15900), min(n - 1, 60000)]))
        nodes = rng.choice(n, size=total, replace=False).astype(np.int32)
        a["pi"][nodes] += int(rng.integers(-4, 5))
        cuts = sorted(set([0, total] + [int(x) for x in rng.integers(0, total + 1, int(rng.integers(0, 5)))]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            eng.append_potential(nodes[lo:hi], a["pi"][nodes[lo:hi]])
    assert np.array_equal(eng.download_pi(), a["pi"])
    assert np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_potential_lists_as_runs_of_consecutive_ids(mode, monkeypatch):
    """mcf_engine_shift_potential_runs: a pivot's list as {first id, length} pairs (bound potentials), in one call or several, runs longer than a
    pair may cover, mixed with ordinary lists before and after in the same pivot (ids after runs, runs after ids: one form per list).  The
    register-resident candidate grid takes the pairs as they are (range-encoded shift lines), every other engine expands them."""
    flags = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(77)
    m_s, n = 400003, 100001
    a = _random_soa(rng, m_s, n, 3, 9)
    eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=flags)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    with pytest.raises(M.McfError):
        eng.shift_potential_runs([0], [5], 1)                    # nothing bound: runs carry no values
    pi = a["pi"]
    eng.bind_potentials(pi)
    for it in range(14):
        f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        f2, e2, c2 = eng.find_entering()
        assert f2 == f and (not f or (e2, c2) == (e, c)), (it, e2, e, c2, c)
        arcs = rng.choice(m_s, size=2, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 2).astype(np.int8)
        a["state"][arcs] = vals
        eng.patch_state(arcs, vals)
        R = int(rng.choice([1, 3, 40, 900, 6000]))
        cut = np.sort(rng.choice(n, size=2 * R, replace=False))
        first, length = cut[0::2].astype(np.int32), (cut[1::2] - cut[0::2]).astype(np.int32)       # disjoint runs [first, first + length)
        sigma = int(rng.integers(-4, 5)) or 1
        for lo, ln in zip(first, length):
            pi[lo:lo + ln] += sigma
        kind = it % 5
        if kind == 3:                                            # a big list of ids first, the runs continue it
            free = np.setdiff1d(np.arange(n), np.concatenate([np.arange(lo, lo + ln) for lo, ln in zip(first, length)]))
            ids = rng.choice(free, size=700, replace=False).astype(np.int32)
            pi[ids] += sigma
            eng.shift_potential(ids, None, sigma)
        pieces = sorted(set([0, R] + [int(x) for x in rng.integers(0, R + 1, int(rng.integers(0, 3)))]))
        for lo, hi in zip(pieces[:-1], pieces[1:]):
            eng.shift_potential_runs(first[lo:hi], length[lo:hi], sigma)
        if kind == 1:                                            # a few nodes on top of it, named one by one
            few = rng.choice(n, size=4, replace=False).astype(np.int32)
            pi[few] += 7
            eng.shift_potential(few, pi[few], 7)
        if kind == 4:                                            # ... or a big list of ids behind the runs
            free = np.setdiff1d(np.arange(n), np.concatenate([np.arange(lo, lo + ln) for lo, ln in zip(first, length)]))
            ids = rng.choice(free, size=900, replace=False).astype(np.int32)
            pi[ids] += sigma
            eng.shift_potential(ids, None, sigma)
    f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
    assert eng.find_entering() == (f, e, c)
    with pytest.raises(M.McfError):
        eng.shift_potential_runs([n - 3], [5], 1)                # runs off the end of the graph
    assert np.array_equal(eng.download_pi(), a["pi"])
    assert np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_search_in_two_halves(mode, monkeypatch):
    """mcf_engine_search_begin / _end: the answer is the blocking call's; patches queued while a search is in flight belong to the next one;
    parking the engine (or reading its statistics) in between keeps the answer."""
    rc_layout = isinstance(mode, str)
    rc_resident = rc_layout and mode != "rc"
    mode = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(606)
    m_s, n = 120007, 30000
    a = _random_soa(rng, m_s, n, 4, 12)
    eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=mode)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    with pytest.raises(M.McfError):
        eng.search_end()                                  # nothing in flight
    for it in range(10):
        want = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        eng.search_begin()
        with pytest.raises(M.McfError):
            eng.search_begin()                            # one at a time
        # queued now, seen by the NEXT search only
        arcs = rng.choice(m_s, size=2, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 2).astype(np.int8)
        nodes = rng.choice(n, size=int(rng.choice([1, 30, 6000])), replace=False).astype(np.int32)
        eng.patch_state(arcs, vals)
        if it % 3 == 1:
            eng.park()
        if it % 3 == 2:
            eng.stats()
        f, e, c = eng.search_end()
        assert f == want[0] and (not f or (e, c) == (want[1], want[2])), (it, e, want)
        a["state"][arcs] = vals
        a["pi"][nodes] -= 3
        eng.append_potential(nodes, a["pi"][nodes])
    assert np.array_equal(eng.download_pi(), a["pi"])
    assert np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [pytest.param(M.ENGINE_NO_CANDIDATES, id="resident"), pytest.param(M.ENGINE_DISPATCH, id="dispatch"),
                                  pytest.param(M.ENGINE_SHARE_DEVICE | M.ENGINE_NO_CANDIDATES, id="resident-shared"), pytest.param(0, id="candidates"),
                                  pytest.param("rc", id="rc-layout"), pytest.param("rc-resident", id="rc-resident-lds"), pytest.param("rc-stream", id="rc-resident-stream"),
         # ... and the candidate cache on top of either resident RC grid (Best Eligible on sparse graphs; the other rules run as above)
         pytest.param("rc-cand", id="rc-candidates-lds"), pytest.param("rc-cand-stream", id="rc-candidates-stream")])
@pytest.mark.parametrize("m_s,n", [(400003, 100001), (60001, 16000)])
def test_state_patch_lists_of_any_length(mode, m_s, n, monkeypatch):
    """mcf_engine_patch_state with 65, 200 and 5000 distinct arcs between two searches, each time with a potential list pending
    ('Queued; ordered before the next search', include/mcf_hip.h): the resident grid receives them through the mailbox (up to 4096)
    or, beyond that, is stopped first; dispatch mode ships them with update_kernel."""
    mode = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(2024 + m_s)
    a = _random_soa(rng, m_s, n, 3, 9)
    eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=mode)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for it, k_st in enumerate([65, 3, 200, 5000, 64, 4097, 1]):
        f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        f2, e2, c2 = eng.find_entering()
        assert f2 == f and (not f or (e2, c2) == (e, c)), (it, k_st, e2, e, c2, c)
        nodes = rng.choice(n, size=int(rng.choice([1, 7, 300, 5000])), replace=False).astype(np.int32)
        a["pi"][nodes] += int(rng.integers(-4, 5))
        if it % 2:
            eng.set_potential(nodes, a["pi"][nodes])      # potentials first, states after ...
        arcs = rng.choice(m_s, size=k_st, replace=False).astype(np.int32)
        vals = rng.integers(-1, 2, k_st).astype(np.int8)
        a["state"][arcs] = vals
        # in several calls, repeating some arcs with their final value
        cut = k_st // 3
        eng.patch_state(arcs[:cut], vals[:cut])
        eng.patch_state(arcs[cut:], vals[cut:])
        eng.patch_state(arcs[:2], vals[:2])
        if not it % 2:
            eng.set_potential(nodes, a["pi"][nodes])      # ... or the other way round
    f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
    assert eng.find_entering() == (f, e, c)
    assert np.array_equal(eng.download_pi(), a["pi"])
    assert np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [pytest.param(0, id="resident"), pytest.param(M.ENGINE_DISPATCH, id="dispatch")])
@pytest.mark.parametrize("name", ["netgen_8_10a", "transport_40x30", "circulation_100_0_10", "AURV19V6", "netgen_8_13a"])
def test_block_search_sizing_of_the_reference_default(name, mode):
    """`new NetworkSimplex(g).Solve()`: plain Block Search, auto-configured (SmallBlocksForDense, adaptive block size) -- and the same
    rule under explicit configurations, including one that makes the block GROW.  Pivot for pivot, block size for block size."""
    p = load(name)
    cases = [dict(auto=True), dict(auto=False),
             dict(config=dict(flags=1)), dict(config=dict(flags=3, min_block_size=10, max_block_size=50)),
             dict(config=dict(flags=1, low_hit_rate_threshold=0.001, high_hit_rate_threshold=0.002, max_block_size=400, block_size_growth_factor=1.3))]
    if name == "netgen_8_13a":
        cases = cases[:1]
    for kw in cases:
        o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP, O.RULE_BLOCK, flags=mode, **kw)
        assert st == st_o == O.OPTIMAL, (name, kw)
        assert np.array_equal(ns.trace(), tr_o), (name, kw, int(np.argmax(ns.trace()[: len(tr_o)] != tr_o[: len(ns.trace())])))
        assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
        m = ns.get_metrics()
        assert (m["initial_block_size"], m["final_block_size"]) == (o.initial_block_size, o.block_size), (name, kw)
        assert m["total_arcs_checked"] == o.arcs_checked and m["config_flags"] == o.config_flags
        assert abs(m["average_arcs_checked_per_pivot"] - o.arcs_checked / max(o.n_pivots, 1)) < 1e-9
    # EnableOptimizedPivot(true) ignores the configuration (BlockSearchPivotOptimized.cs:27-28)
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK, flags=mode, config=dict(flags=3, min_block_size=10))
    assert st == st_o == O.OPTIMAL and np.array_equal(ns.trace(), tr_o)


@pytest.mark.gpu
def test_raw_engine_adapts_its_block_size_per_request():
    """PivotEngine + set_block_config on random arrays: every search travels with its own block size (resident mailbox / kernel arguments)."""
    rng = np.random.default_rng(515)
    m_s, n = 90001, 20000
    for mode in (0, M.ENGINE_DISPATCH):
        a = _random_soa(rng, m_s, n, 40, 400, extra=0)
        cfg = dict(flags=1, consecutive_hits_before_adapt=1, block_size_shrink_factor=0.6, min_block_size=7, min_block_size_ratio=0.0)
        eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BlockSearch, optimized=False, flags=mode)
        c = M.block_config(**cfg)
        eng.set_block_config(c, n - 1)
        block, dmin = C.c_int32(), C.c_int32()
        L.check(L.lib().mcf_block_initial_size(C.byref(c), m_s, n - 1, C.byref(block), C.byref(dmin)))
        counters = (C.c_int32 * 2)(0, 0)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        next_arc, seen = 0, set()
        for it in range(25):
            assert eng.block_size == block.value
            seen.add(block.value)
            f, e, c_, na = _oracle_scan(O.RULE_BLOCK, False, a, m_s, block.value, next_arc)
            f2, e2, c2 = eng.find_entering()
            assert (f2, e2, c2) == (f, e, c_), (it, block.value)
            assert eng.next_arc == na
            # arcs the reference loop examined: from next_arc to the new next_arc inclusive, cyclically
            checked = (na - next_arc) % m_s + 1 if na != next_arc else (1 if block.value == 1 else m_s)
            L.check(L.lib().mcf_block_adapt(C.byref(c), dmin.value, checked, C.byref(block), counters))
            next_arc = na
            a["state"][e] = 0
            eng.patch_state([e], [0])
        assert len(seen) > 3 and eng.stats()["arcs_checked"] > 0


# ------------------------------------------------------------------ arc shards: resident shard engines, in-process shard groups, BASELINE config 5

@pytest.mark.gpu
@pytest.mark.parametrize("rule,optimized,vw", RULE_CASES)
def test_resident_shard_engines_resolve_like_one_engine(rule, optimized, vw):
    """The shard test above with every shard served by its own RESIDENT grid (co-resident on this GPU: MCF_ENGINE_SHARE_DEVICE and a
    grid cap), searches posted to all shards first and collected afterwards (mcf_engine_search_begin / _search_end_local)."""
    rng = np.random.default_rng(199 + rule)
    m_s, n, world = 150_011, 30_000, 3
    a = _random_soa(rng, m_s, n, 5, 12, extra=0)
    block = 173
    one = M.PivotEngine(n, m_s, m_s, rule=RULES[rule], optimized=optimized, block_size=block, flags=M.ENGINE_DISPATCH, vector_width=vw)
    one.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    shards = []
    for r in range(world):
        e = M.PivotEngine(n, m_s, m_s, rule=RULES[rule], optimized=optimized, block_size=block, shard=M.shard_range(m_s, r, world),
                          flags=M.ENGINE_SHARE_DEVICE, resident_workgroups=64, vector_width=vw)
        e.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        assert e.stats()["resident"] == 1
        shards.append(e)
    for it in range(12):
        want = one.find_entering()
        for e in shards:
            e.search_begin()
        cands = [e.search_end_local() for e in shards]
        got = [e.resolve(cands) for e in shards]
        assert all(g == want for g in got), (it, want, got)
        assert all(e.next_arc == one.next_arc for e in shards)
        if want[0]:
            arcs = np.array([want[1], int(rng.integers(0, m_s))], np.int32); vals = np.array([0, int(rng.integers(-1, 2))], np.int8)
            nodes = rng.choice(n, size=int(rng.choice([1, 40, 3000])), replace=False).astype(np.int32)
            for e in [one] + shards:
                e.patch_state(arcs, vals)
                e.update_potential(nodes, -3)
    st = [e.stats() for e in shards]
    assert all(x["resident_requests"] + x["host_decided"] >= 12 for x in st)          # Best Eligible: every shard answers from its own candidate cache when it can
    assert all(x["candidates"] == (1 if rule == O.RULE_BEST else 0) for x in st)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 3, 8])
def test_shard_group_solves_pivot_for_pivot(shards):
    """mcf_ns_set_shard_group: one host thread, R engines (here co-resident on one GPU) each holding an arc shard, host-side MINLOC over
    their answers -- the pivot sequence of the un-sharded solve, for Best Eligible and both Block Search flavours (the plain one with the
    reference's adaptive block size, which every shard has to follow in step)."""
    cases = [(load("netgen_8_14a"), [(O.SEM_CSHARP_OPT, O.RULE_BEST, 4)]),
             (None, [(O.SEM_CSHARP_OPT, O.RULE_BEST, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 0), (O.SEM_CSHARP, O.RULE_BLOCK, 4)])]
    g = M.netgen_like(7, 20_000, 60_000, 100, 100)
    for p, rules in cases:
        if p is None:
            p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
        for sem, rule, vw in rules:
            o = O.Oracle(p, sem, rule, auto_config=True, vector_width=vw)
            st_o, tr_o = o.solve(trace_cap=4_000_000)
            ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
            ns.set_pivot_rule(RULES[rule]).enable_optimized_pivot(sem == O.SEM_CSHARP_OPT).set_vector_width(vw).record_trace(4_000_000)
            ns.set_shard_group([0] * shards)
            assert ns.solve() == st_o == O.OPTIMAL
            assert np.array_equal(ns.trace(), tr_o), (shards, sem, rule, vw, int(np.argmax(ns.trace()[: len(tr_o)] != tr_o[: len(ns.trace())])))
            assert ns.get_total_cost() == o.total_cost
            assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
            if sem == O.SEM_CSHARP:
                m = ns.get_metrics()
                assert (m["initial_block_size"], m["final_block_size"], m["total_arcs_checked"]) == (o.initial_block_size, o.block_size, o.arcs_checked)


CONFIG5 = (13502460, 1_000_000, 8_000_000, 1000, 1000)


@pytest.mark.gpu
def test_config5_solves_end_to_end_and_shards_follow_the_same_pivots():
    """BASELINE.json configs[4] on ONE GPU, end to end: NETGEN-like 1M nodes / 8M arcs, Best Eligible, int64, through mcf_ns_solve with
    the layout the engine picks by itself at this size (reduced costs kept per arc, streamed by a resident grid of 256 workgroups).  Checked by
      * the device validator's optimality certificate (mcf_ns_validate: conservation, bounds, complementary slackness, primal = dual),
      * the optimal cost of the CPU oracle's Block-Search solve of the same instance (tests/golden/config5_cost.json; 15-20 CPU minutes,
        hence golden -- generated by tests/golden/make_config5_cost.py),
      * the oracle's own Best-Eligible pivots for the first 1500 searches, arc for arc,
      * and the same instance as 2 and 3 arc shards (mcf_ns_set_shard_group on this GPU) for the first 3000 pivots against the
        un-sharded trace."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5_cost.json")))
    g = M.netgen_like(*CONFIG5)
    assert int(g.source.astype(np.int64).sum()) == gold["checksum_source"] and int(g.cost.sum()) == gold["checksum_cost"]      # the same instance
    keep = 3000
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64).record_trace(keep)
    assert ns.solve() == M.SolverStatus.Optimal
    m = ns.get_metrics()
    assert m["search_arc_num"] == gold["search_arc_num"] == 9_000_000
    # the whole Best-Eligible pivot sequence of this instance was recorded arc for arc in round 2 -- the default engine against the plainest
    # path (every search on the device, every walk the subtree's, every list shifted arc by arc): 2 071 293 identical entering arcs
    # (profiles/r02_config5_whole_pivot_sequence_identical.txt).  A search that ever read a stale reduced cost shows up here as another count
    assert m["iterations"] == 2_071_293, m["iterations"]
    assert ns.check_reduced_costs() == 0
    # the grid leaves for a relabelling of the nodes (a handful per solve: how many depends on the host's speed, the policy weighs their cost)
    # and for nothing else: no list, however long, and no reload of _pi stops it
    e5 = m["engine"]
    assert e5["resident_launches"] <= 2 * e5["renumberings"] + 4 and e5["renumberings"] <= 40 and e5["rc_reloads_in_grid"] > 10_000, (e5["resident_launches"], e5["renumberings"], e5["rc_reloads_in_grid"])
    assert m["engine"]["resident"] == 1 and m["engine"]["rc_layout"] == 1 and m["engine"]["scan_workgroups"] == 256     # what the engine chooses at this size
    cost = ns.get_total_cost()
    assert cost == gold["total_cost"], (cost, gold["total_cost"])
    v = ns.validate()
    assert v["valid"] == 1 and v["objective"] == v["dual_cost"] == cost, v
    trace = ns.trace()
    assert len(trace) == keep
    del ns
    # the oracle's Best Eligible on the same instance: 9M arcs per search on one core, so only the first searches
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BEST)
    assert o.init()
    for it in range(1500):
        f, e = o.find_entering()
        assert f and e == trace[it], (it, e, int(trace[it]))
        o.apply_pivot(e)
    del o
    for shards in (2, 3):
        nsg = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64)
        nsg.set_shard_group([0] * shards).set_pivot_limit(keep).record_trace(keep)
        assert nsg.solve() == M.SolverStatus.NotSolved
        assert np.array_equal(nsg.trace(), trace), (shards, int(np.argmax(nsg.trace() != trace)))
        del nsg


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [pytest.param(M.ENGINE_NO_CANDIDATES, id="resident"), pytest.param(M.ENGINE_DISPATCH, id="dispatch"), pytest.param(0, id="candidates-or-default"),
                                  pytest.param("rc", id="rc-layout"), pytest.param("rc-resident", id="rc-resident-lds"), pytest.param("rc-stream", id="rc-resident-stream"),
         # ... and the candidate cache on top of either resident RC grid (Best Eligible on sparse graphs; the other rules run as above)
         pytest.param("rc-cand", id="rc-candidates-lds"), pytest.param("rc-cand-stream", id="rc-candidates-stream")])
def test_patch_arcs_rewrites_end_points_and_costs(mode, monkeypatch):
    """mcf_engine_patch_arcs (artificial arcs re-pointed by a warm start): source / target / cost of some arcs change between two searches;
    in the RC layout the per-arc reduced costs and the nodes' arc lists are rebuilt, and so are the candidate cache's host mirrors."""
    flags = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(808)
    m_s, n = 70_001, 18_000
    a = _random_soa(rng, m_s, n, 30, 300)
    for rule in (O.RULE_BEST, O.RULE_BLOCK):
        eng = M.PivotEngine(n, len(a["src"]), m_s, rule=RULES[rule], optimized=True, block_size=211, flags=flags)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        next_arc = 0
        for it in range(6):
            f, e, c, na = _oracle_scan(rule, True, a, m_s, 211, next_arc)
            assert eng.find_entering() == (f, e, c), it
            next_arc = na
            arcs = rng.choice(len(a["src"]), size=int(rng.choice([1, 5, 40])), replace=False).astype(np.int32)
            src = rng.integers(0, n, len(arcs)).astype(np.int32); tgt = rng.integers(0, n, len(arcs)).astype(np.int32)
            cost = rng.integers(-30, 31, len(arcs)).astype(np.int64)
            a["src"][arcs] = src; a["tgt"][arcs] = tgt; a["cost"][arcs] = cost
            eng.patch_arcs(arcs, src, tgt, cost)
            nodes = rng.choice(n, size=int(rng.choice([1, 200])), replace=False).astype(np.int32)
            a["pi"][nodes] += 2
            eng.update_potential(nodes, 2)
        assert np.array_equal(eng.download_pi(), a["pi"])


@pytest.mark.gpu
def test_rc_layout_shards_and_shard_group(monkeypatch):
    """Arc shards in the RC layout: every shard keeps the reduced costs of ITS arcs and the arc lists of all nodes restricted to them."""
    monkeypatch.setenv("MCF_HIP_RC", "1")
    rng = np.random.default_rng(5150)
    m_s, n, world = 120_003, 25_000, 3
    a = _random_soa(rng, m_s, n, 4, 12, extra=0)
    shards = []
    for r in range(world):
        e = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, optimized=True, shard=M.shard_range(m_s, r, world), flags=M.ENGINE_DISPATCH)
        e.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        assert e.stats()["rc_layout"] == 1
        shards.append(e)
    for it in range(10):
        f, arc, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        cands = [e.find_entering_local() for e in shards]
        got = [e.resolve(cands) for e in shards]
        assert all(g[0] == f and (not f or (g[1], g[2]) == (arc, c)) for g in got), (it, got, arc, c)
        arcs = rng.choice(m_s, size=3, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 3).astype(np.int8)
        a["state"][arcs] = vals
        nodes = rng.choice(n, size=int(rng.choice([1, 40, 5000])), replace=False).astype(np.int32)
        a["pi"][nodes] -= 2
        for e in shards:
            e.patch_state(arcs, vals)
            e.update_potential(nodes, -2)
    p = load("netgen_8_14a")
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BEST)
    st_o, tr_o = o.solve(trace_cap=1 << 22)
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True)
    ns.set_shard_group([0, 0]).record_trace(1 << 22)
    assert ns.solve() == st_o == O.OPTIMAL and np.array_equal(ns.trace(), tr_o)
    assert ns.get_metrics()["engine"]["rc_layout"] == 1


@pytest.mark.gpu
def test_candidate_cache_survives_the_wrap_of_its_epoch_counter(monkeypatch):
    """Every search has an epoch number (32 bits); MCF_HIP_CAND_EPOCH0 starts the counter 300 searches before it would wrap: the cache starts
    over there and the pivots stay the oracle's."""
    monkeypatch.setenv("MCF_HIP_CAND_EPOCH0", str(0xFFFFFF00 - 300))
    p = load("netgen_8_13a")
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST)
    assert st == st_o == O.OPTIMAL and len(tr_o) > 5000
    assert np.array_equal(ns.trace(), tr_o), int(np.argmax(ns.trace()[: len(tr_o)] != tr_o[: len(ns.trace())]))
    e = ns.get_metrics()["engine"]
    assert e["candidates"] == 1 and e["host_decided"] > e["resident_requests"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [pytest.param("rc", id="rc-layout"), pytest.param("rc-resident", id="rc-resident-lds"), pytest.param("rc-stream", id="rc-resident-stream"),
                                  pytest.param("rc-cand", id="rc-candidates-lds"), pytest.param("rc-cand-stream", id="rc-candidates-stream")])
def test_rc_layout_with_hub_nodes(mode, monkeypatch):
    """RC layout on a graph with hubs (two nodes carry a third of the arc ends): a hub's arc list is far longer than what rides in a scan's
    arguments, and in update_rc_kernel it is walked by a whole workgroup instead of one thread."""
    flags = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(919)
    m_s, n = 90_007, 12_000
    a = _random_soa(rng, m_s, n, 9, 40, extra=0)
    hub = rng.random(m_s)
    a["src"][hub < 0.2] = 0
    a["tgt"][(hub >= 0.2) & (hub < 0.33)] = 7
    eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=flags)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for it in range(12):
        f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        assert eng.find_entering() == (f, e, c), it
        k = int(rng.choice([1, 3, 40, 700, 3000]))
        nodes = rng.choice(np.arange(8, n), size=k, replace=False).astype(np.int32)
        if it % 2 == 0:
            nodes[0] = 0                      # the hub moves with the list
        if it % 3 == 0 and k > 1:
            nodes[1] = 7
        sigma = int(rng.integers(-6, 7))
        a["pi"][nodes] += sigma
        eng.shift_potential(nodes, a["pi"][nodes], sigma)
        arcs = rng.choice(m_s, size=2, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 2).astype(np.int8)
        a["state"][arcs] = vals
        eng.patch_state(arcs, vals)
    f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
    assert eng.find_entering() == (f, e, c)
    assert np.array_equal(eng.download_pi(), a["pi"]) and np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])
    assert eng.stats()["rc_layout"] == 1


@pytest.mark.gpu
def test_headline_workload_is_pivot_for_pivot_the_oracle():
    """BASELINE config 3 -- the bench's headline workload -- at full size, default engine (resident grid + candidate cache): all 190 580 entering
    arcs equal the CPU oracle's Best-Eligible pivots (about a minute of one core), and so do cost, flows and potentials."""
    g = M.netgen_like(13502460, 100_000, 300_000, 316, 316)
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(0, 64).record_trace(1 << 20)
    assert ns.solve() == M.SolverStatus.Optimal
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BEST)
    st_o, tr_o = o.solve(trace_cap=1 << 20)
    assert st_o == O.OPTIMAL
    tr = ns.trace()
    assert len(tr) == len(tr_o) == o.n_pivots
    assert np.array_equal(tr, tr_o), int(np.argmax(tr != tr_o))
    assert ns.get_total_cost() == o.total_cost
    assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
    e = ns.get_metrics()["engine"]
    assert e["candidates"] == 1 and e["host_decided"] > 5 * e["resident_requests"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [pytest.param(0, id="candidates"), pytest.param(M.ENGINE_NO_CANDIDATES, id="resident"), pytest.param(M.ENGINE_DISPATCH, id="dispatch")])
def test_overlapping_potential_lists_between_two_searches(mode):
    """Two mcf_engine_set_potential calls between two searches, the second repeating nodes of the first with newer values (a big list then
    a small one, two big ones, a small one then a big one): the later value wins in every engine mode."""
    rng = np.random.default_rng(4711)
    m_s, n = 150_001, 40_000
    a = _random_soa(rng, m_s, n, 5, 15)
    eng = M.PivotEngine(n, len(a["src"]), m_s, rule=M.PivotRule.BestEligible, optimized=True, flags=mode)
    eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
    for it, (k1, k2) in enumerate([(6000, 3), (5000, 7000), (4, 6000), (2, 2), (6000, 40), (9000, 9000)]):
        f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
        assert eng.find_entering() == (f, e, c), it
        first = rng.choice(n, size=k1, replace=False).astype(np.int32)
        a["pi"][first] -= 3
        eng.set_potential(first, a["pi"][first])
        second = np.concatenate([first[: min(k1, k2) // 2 + 1], rng.choice(n, size=k2, replace=False).astype(np.int32)])
        second = np.unique(second).astype(np.int32)
        a["pi"][second] += 5
        eng.set_potential(second, a["pi"][second])
    f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
    assert eng.find_entering() == (f, e, c)
    assert np.array_equal(eng.download_pi(), a["pi"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_solve_with_nodes_relabelled_in_thread_order(mode, monkeypatch):
    """mcf_ns_solve relabels the nodes in thread order every so often (renumber_nodes / mcf_engine_renumber_nodes) so that the host's walks run
    through memory front to back.  Arc ids never change, so the pivots, flows and (after the ids are restored) potentials are the reference's.
    Forced here to happen hundreds of times per solve, in every engine mode."""
    rc_layout = isinstance(mode, str)
    flags = _mode_flags(mode, monkeypatch)
    monkeypatch.setenv("MCF_NS_RENUMBER", "0.05")
    for name, cases in [("netgen_8_10a", [(O.SEM_CSHARP_OPT, O.RULE_BEST, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 4), (O.SEM_CSHARP, O.RULE_BLOCK, 4), (O.SEM_CSHARP_OPT, O.RULE_FIRST, 4)]),
                        ("AURV19V6", [(O.SEM_CSHARP_OPT, O.RULE_BEST, 4), (O.SEM_CSHARP_OPT, O.RULE_BLOCK, 0)])]:
        p = load(name)
        for sem, rule, vw in cases:
            o, st_o, tr_o, ns, st = _solve_both(p, sem, rule, flags=flags, vw=vw)
            assert st == st_o == O.OPTIMAL
            assert np.array_equal(ns.trace(), tr_o), (name, sem, rule)
            assert ns.get_total_cost() == o.total_cost and np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
            m = ns.get_metrics()
            # (an engine that shares its device with other solvers' grids declines: a relabelling stops and restarts the resident grid)
            shared = not rc_layout and bool(flags & M.ENGINE_SHARE_DEVICE)
            assert (m["engine"]["renumberings"] == 0) if shared else (m["engine"]["renumberings"] >= 5), (name, sem, rule, m["engine"]["renumberings"])
            v = ns.validate()
            assert v["valid"] == 1 and v["objective"] == v["dual_cost"] == o.total_cost


@pytest.mark.gpu
@pytest.mark.parametrize("bound", [False, True], ids=["mirror", "bound-potentials"])
@pytest.mark.parametrize("mode", MODES)
def test_raw_engine_follows_a_relabelling_of_its_nodes(mode, bound, monkeypatch):
    """mcf_engine_renumber_nodes between searches, with patches queued before it (they go out under the old ids) and after it."""
    flags = _mode_flags(mode, monkeypatch)
    rng = np.random.default_rng(5)
    for m_s, n in [(5003, 700), (400003, 100001)]:
        a = _random_soa(rng, m_s, n, 50, 1000, extra=0)
        eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, flags=flags)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        pi = a["pi"].copy()
        if bound:
            eng.bind_potentials(pi)
        for it in range(8):
            f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
            assert eng.find_entering() == (f, e, c), (m_s, it)
            arcs = rng.choice(m_s, 2, replace=False).astype(np.int32); vals = rng.integers(-1, 2, 2).astype(np.int8)
            a["state"][arcs] = vals
            eng.patch_state(arcs, vals)
            k = int(rng.choice([1, 30, 700]))
            nodes = rng.choice(n, size=min(k, n), replace=False).astype(np.int32)
            sigma = int(rng.integers(-40, 41))
            a["pi"][nodes] += sigma
            if bound:
                pi[nodes] += sigma
                eng.shift_potential(nodes, pi[nodes] if it % 3 else None, sigma)      # None: "the bound array holds them"
            else:
                eng.update_potential(nodes, sigma)
            if it % 2 == 1:
                new_of = rng.permutation(n).astype(np.int32)
                back = np.empty(n, np.int32); back[new_of] = np.arange(n, dtype=np.int32)
                a["src"], a["tgt"] = new_of[a["src"]], new_of[a["tgt"]]
                a["pi"] = a["pi"][back].copy()
                if bound:
                    pi[:] = pi[back]
                eng.renumber_nodes(new_of)
        assert np.array_equal(eng.download_pi(), a["pi"])
        assert eng.stats()["renumberings"] == 4
    with pytest.raises(M.McfError):
        eng.renumber_nodes(np.zeros(n, np.int32))          # not a permutation
    if not bound:
        with pytest.raises(M.McfError):
            eng.shift_potential(np.arange(3, dtype=np.int32), None, 1)      # no values and no bound array to read them from


@pytest.mark.gpu
def test_candidate_heap_is_swept_of_outdated_entries(monkeypatch):
    """candidate_cache.hip.h: the heap of touched arcs deletes lazily and is compacted when it outgrows a bound (2^18 entries; config 5 reaches it).
    With the bound at 64 entries a mid-size solve compacts hundreds of times -- and still takes the oracle's pivots."""
    monkeypatch.setenv("MCF_HIP_CAND_HEAP_COMPACT", "64")
    g = M.netgen_like(13502460, 20_000, 70_000, 140, 140)
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    o, st_o, tr_o, ns, st = _solve_both(p, O.SEM_CSHARP_OPT, O.RULE_BEST, flags=0)
    assert st == st_o == O.OPTIMAL and np.array_equal(ns.trace(), tr_o)
    e = ns.get_metrics()["engine"]
    assert e["candidates"] == 1 and e["heap_compactions"] > 20, e["heap_compactions"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_grid_that_left_on_its_idle_timeout_comes_back(mode, monkeypatch):
    """A resident grid that hears nothing for its idle time leaves by itself; the next search finds the exit record while it waits for its
    answer and starts the grid again (collect / cand_collect -> resident_restart).  With the idle time cut to 5 ms and pauses between the
    searches that happens before every other search -- with patches queued meanwhile, in every engine mode (the grid that is patched straight
    from the request gets its arrays written again from the host's mirrors and the waiting request put there again without patches)."""
    import time
    flags = _mode_flags(mode, monkeypatch)
    monkeypatch.setenv("MCF_HIP_IDLE_MS", "5")
    rng = np.random.default_rng(77)
    for m_s, n in [(60001, 17000), (9001, 1200)]:
        a = _random_soa(rng, m_s, n, 60, 500, extra=0)
        eng = M.PivotEngine(n, m_s, m_s, rule=M.PivotRule.BestEligible, flags=flags)
        eng.upload(a["src"], a["tgt"], a["cost"], a["state"], a["pi"])
        for it in range(10):
            f, e, c, _ = _oracle_scan(O.RULE_BEST, True, a, m_s, 1, 0)
            assert eng.find_entering() == (f, e, c), (m_s, it)
            if f:
                a["state"][e] = 0
                eng.patch_state([e], [0])
            nodes = rng.choice(n, size=int(rng.choice([1, 4, 300, 2000])), replace=False).astype(np.int32)
            sigma = int(rng.integers(-30, 31))
            a["pi"][nodes] += sigma
            eng.update_potential(nodes, sigma)
            if it % 2 == 0:
                time.sleep(0.03)          # the grid is gone when the next search is posted
        st = eng.stats()
        if st["resident"]:
            assert st["resident_launches"] >= 4, st["resident_launches"]
        assert np.array_equal(eng.download_pi(), a["pi"]) and np.array_equal(eng.download_state()[:m_s], a["state"][:m_s])
        assert eng.check_reduced_costs() == (0, -1)
