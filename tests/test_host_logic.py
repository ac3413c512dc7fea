"""CPU tests of the product's host side: the C ABI surface, the sequential pivot driver (replayed against the oracle
with entering arcs supplied by the test -- the library itself has no CPU search), generators, DIMACS I/O, errors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kat_data as K
import mincostflow_amd as M
from helpers import fixtures, load, problem_from_dict, validate_solution
from mincostflow_amd import _lib as L
from oracle import ns_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mcf_hip.h")).read()
    declared = set(re.findall(r"MCF_API\s+[\w\s\*]+?\b(mcf_\w+)\s*\(", header))
    assert len(declared) >= 55
    lib = C.CDLL(M.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mcf_hip.h but not exported"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    assert b"gfx950" in L.lib().mcf_version()


def test_no_cpu_search_path_without_device(have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    p = load("transport_2x3")
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
    with pytest.raises(M.McfError) as ei:
        ns.solve()
    assert ei.value.code == L.ERR_NO_DEVICE
    with pytest.raises(M.McfError) as ei:
        M.PivotEngine(10, 20, 20)
    assert ei.value.code == L.ERR_NO_DEVICE


def _replay(p, sem, rule, supply_type=O.GEQ):
    o = O.Oracle(p, sem, rule, supply_type=supply_type)
    ok = o.init()
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply).set_supply_type(supply_type)
    st = ns.begin()
    if not ok:
        assert st == M.SolverStatus.Infeasible
        return None, ns
    it, oa = ns.internal(), o.internal_arrays()
    A = o.all_arc_num
    assert it["search_arc_num"] == o.search_arc_num
    for k, ok_ in (("source", "src"), ("target", "tgt"), ("cost", "cost"), ("state", "state")):
        assert np.array_equal(it[k][:A], oa[ok_][:A]), k
    assert np.array_equal(it["pi"], oa["pi"])
    n = 0
    while True:
        f, e = o.find_entering()
        if not f:
            break
        r = o.apply_pivot(e)
        unb = ns.apply_pivot(e)
        assert unb == (r == O.UNBOUNDED)
        if unb:
            break
        lp = ns.last_pivot()
        assert len(lp["nodes"]) == o.last_subtree and lp["sigma"] == o.last_sigma
        assert len(set(lp["nodes"].tolist())) == len(lp["nodes"])
        n += 1
    st_o = o.finish() if o.status != O.UNBOUNDED and not unb_or(o) else O.UNBOUNDED
    st = ns.finish()
    return (o, st_o, n), (ns, st)


def unb_or(o):
    return o.status == O.UNBOUNDED


@pytest.mark.parametrize("name", ["netgen_8_08a", "netgen_8_10a", "transport_40x30", "circulation_100_0_10",
                                  "SimpleProblemIllustration2NonSparse", "assignment_50x50", "grid_5x5", "AURV19V6"])
def test_sequential_driver_replays_identically(name):
    """Tree surgery, flows, potentials and the per-pivot device patches of the product equal the oracle's when both are
    fed the same entering arcs (C# semantics, all rules)."""
    p = load(name)
    for sem, rule in [(O.SEM_CSHARP_OPT, O.RULE_BLOCK), (O.SEM_CSHARP, O.RULE_BEST), (O.SEM_CSHARP_OPT, O.RULE_FIRST)]:
        if p.m > 20000 and rule != O.RULE_BLOCK:
            continue
        (o, st_o, n), (ns, st) = _replay(p, sem, rule)
        assert st == st_o == O.OPTIMAL
        assert ns.get_total_cost() == o.total_cost
        assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
        it, oa = ns.internal(), o.internal_arrays()
        assert np.array_equal(it["state"][: o.all_arc_num], oa["state"][: o.all_arc_num])
        assert np.array_equal(it["pi"], oa["pi"])
        validate_solution(p, ns.flows(), ns.potentials())


@pytest.mark.parametrize("name", ["netgen_8_10a", "transport_40x30", "grid_5x5", "generated-6000"])
@pytest.mark.parametrize("renumber_every", [0.0, 0.1, 0.5, 4.0])
@pytest.mark.parametrize("smaller_side", [False, True])
def test_replay_with_relabelled_nodes_and_smaller_side_walks(name, renumber_every, smaller_side):
    """mcf_ns_replay: a recorded sequence of entering arcs through the host driver's own walk aids -- the smaller side of the tree moves, the
    nodes are relabelled in thread order every `renumber_every` * n walked nodes (then the walks go in runs of consecutive ids) -- ends with the
    oracle's flows, states and potentials (node ids restored, pi[root] back at 0).  No device involved."""
    if name == "generated-6000":       # big enough for walks of 512 nodes and more: after a relabelling those write runs of consecutive ids, not nodes
        g = M.netgen_like(99, 6000, 20000, 60, 60)
        p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    else:
        p = load(name)
    o = O.Oracle(p, O.SEM_CSHARP, O.RULE_BEST)
    assert o.init()
    arcs = []
    while True:
        f, e = o.find_entering()
        if not f:
            break
        assert o.apply_pivot(e) != O.UNBOUNDED
        arcs.append(e)
    assert o.finish() == O.OPTIMAL and len(arcs) > 20
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
    assert ns.begin() != M.SolverStatus.Infeasible
    ns.replay(np.array(arcs, np.int32), smaller_side=smaller_side, renumber_every=renumber_every)
    m = ns.get_metrics()
    assert m["iterations"] == len(arcs)
    assert ns.replay_relabellings == 0 if renumber_every == 0 else (ns.replay_relabellings >= 2 or renumber_every > 0.2), ns.replay_relabellings
    it, oa = ns.internal(), o.internal_arrays()
    assert np.array_equal(it["state"][: o.all_arc_num], oa["state"][: o.all_arc_num])
    assert np.array_equal(it["pi"], oa["pi"])
    assert ns.finish() == O.OPTIMAL
    assert ns.get_total_cost() == o.total_cost
    assert np.array_equal(ns.flows(), o.flow()) and np.array_equal(ns.potentials(), o.potential())
    with pytest.raises(M.McfError):
        ns.replay(np.array([arcs[0]] * 2 + [10 ** 9], np.int32))          # an arc that cannot enter


def test_device_share_arguments():
    """mcf_ns_set_device_share: 0 (the whole device) or 8 .. 256 workgroups; nothing else (no device needed to say so)."""
    p = load("grid_5x5")
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
    ns.set_device_share(0).set_device_share(8).set_device_share(256)
    for bad in (-1, 3, 257):
        with pytest.raises(M.McfError):
            ns.set_device_share(bad)


@pytest.mark.parametrize("kat", K.CSHARP_KATS, ids=[k[0] for k in K.CSHARP_KATS])
def test_csharp_kats_through_the_host_driver(kat):
    name, d, status, cost, flows = kat
    p = problem_from_dict(d)
    res = _replay(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK)
    (o, st_o, n), (ns, st) = res
    assert st == status
    if status == M.SolverStatus.Optimal:
        if cost is not None:
            assert ns.get_total_cost() == cost
        if flows is not None:
            assert ns.flows().tolist() == flows
        for a in range(p.m):
            assert ns.get_flow(a) == ns.flows()[a]
    else:
        with pytest.raises(M.McfError) as ei:       # "Solution not optimal" (NetworkSimplex.cs:418-421)
            ns.get_total_cost()
        assert ei.value.code == L.ERR_STATE


def test_leq_and_lower_bounds():
    d = [c for c in K.LEMON_TABLE if c[0] == 3][0]          # balanced supplies, lower bounds l2: cost 5970
    p = problem_from_dict(d[1])
    (o, st_o, n), (ns, st) = _replay(p, O.SEM_CSHARP, O.RULE_BEST, supply_type=O.LEQ)
    assert st == st_o == O.OPTIMAL and ns.get_total_cost() == o.total_cost == 5970
    assert np.all(ns.flows() >= p.lower)
    validate_solution(p, ns.flows(), ns.potentials(), supply_type=O.LEQ)
    # upper bounds are reduced in place by the transformation (D11, NetworkSimplex.cs:649,:519-527)
    e = int(np.nonzero(p.lower)[0][0])
    assert ns.get_arc_upper_bound(e) == p.upper[e] - p.lower[e]
    # D9: with genuine LEQ slack the C# feasibility check (NetworkSimplex.cs:689,:1272-1283) looks at the root links
    # that carry the slack and answers Infeasible where LEMON finds 5930 -- mirrored, not repaired
    d = [c for c in K.LEMON_TABLE if c[0] == 20][0]
    p = problem_from_dict(d[1])
    (o, st_o, n), (ns, st) = _replay(p, O.SEM_CSHARP, O.RULE_BEST, supply_type=O.LEQ)
    assert st == st_o == O.INFEASIBLE
    ol = O.Oracle(p, O.SEM_LEMON, O.RULE_BEST, supply_type=O.LEQ)
    assert ol.solve()[0] == O.OPTIMAL and ol.total_cost == 5930


def test_argument_errors_mirror_the_reference():
    ns = M.NetworkSimplex(3, [0, 1], [1, 2])
    for call in (lambda: ns.set_arc_cost(2, 1), lambda: ns.set_arc_bounds(-1, 0, 1), lambda: ns.set_node_supply(3, 1),
                 lambda: ns.set_pivot_rule(M.PivotRule.CandidateList)):     # NetworkSimplex.cs:155-158,:884
        with pytest.raises(M.McfError) as ei:
            call()
        assert ei.value.code == L.ERR_INVALID
    with pytest.raises(M.McfError):
        M.NetworkSimplex(2, [0, 5], [1, 1])
    with pytest.raises(M.McfError) as ei:
        ns.apply_pivot(0)           # before begin()
    assert ei.value.code == L.ERR_STATE
    assert ns.begin() == 0
    with pytest.raises(M.McfError):
        ns.apply_pivot(99)
    with pytest.raises(M.McfError) as ei:
        ns.begin()                  # single shot
    assert ei.value.code == L.ERR_STATE


def test_bounds_check_comes_first():
    ns = M.NetworkSimplex(2, [0], [1]).set_arc_bounds(0, 5, 3).set_node_supply(0, 4).set_node_supply(1, -4)
    assert ns.begin() == M.SolverStatus.Infeasible


def test_generators_are_deterministic_and_feasible():
    g1 = M.netgen_like(13502460, 2000, 9000, 40, 40)
    g2 = M.netgen_like(13502460, 2000, 9000, 40, 40)
    for f in ("source", "target", "lower", "upper", "cost", "supply"):
        assert np.array_equal(getattr(g1, f), getattr(g2, f))
    assert g1.node_count == 2000 and g1.arc_count == 9000
    assert g1.supply.sum() == 0 and g1.supply[:40].sum() == 40 * 1000 and np.all(g1.supply[:40] > 0)
    assert np.all(g1.supply[-40:] < 0) and np.all(g1.supply[40:-40] == 0)
    assert np.all(g1.cost >= 1) and np.all(g1.cost <= 10000) and np.all(g1.upper >= 1) and np.all(g1.lower == 0)
    assert np.all(np.diff(g1.source) >= 0), "arcs are grouped by tail node"
    assert np.all(g1.source != g1.target)
    p = O.Problem(g1.node_count, g1.arc_count, g1.source, g1.target, g1.lower, g1.upper, g1.cost, g1.supply)
    costs = set()
    for sem, rule in [(O.SEM_CSHARP_OPT, O.RULE_BLOCK), (O.SEM_LEMON, O.RULE_BLOCK), (O.SEM_CSHARP, O.RULE_BEST)]:
        o = O.Oracle(p, sem, rule)
        assert o.solve()[0] == O.OPTIMAL
        costs.add(o.total_cost)
    assert len(costs) == 1
    g3 = M.netgen_like(7, 2000, 9000, 40, 40)
    assert not np.array_equal(g1.cost, g3.cost)
    a = M.assignment(42, 30)
    assert a.node_count == 60 and a.arc_count == 900 and a.supply.sum() == 0 and set(a.upper.tolist()) == {1}
    pa = O.Problem(a.node_count, a.arc_count, a.source, a.target, a.lower, a.upper, a.cost, a.supply)
    o = O.Oracle(pa, O.SEM_CSHARP_OPT, O.RULE_BEST)
    assert o.solve()[0] == O.OPTIMAL and o.flow().sum() == 30
    with pytest.raises(M.McfError):
        M.netgen_like(1, 100, 50, 10, 10)      # fewer arcs than the skeleton needs


def test_dimacs_reader_matches_fixtures_and_roundtrips(tmp_path):
    for name, path, _ in fixtures():
        if name not in ("netgen_8_08a", "transport_2x3", "AURV19V6", "grid_5x5", "TablesAreCorrectSimple"):
            continue
        a, b = M.read_dimacs(path), load(path)
        assert (a.node_count, a.arc_count) == (b.n, b.m)
        for fa, fb in (("source", "src"), ("target", "tgt"), ("lower", "lower"), ("upper", "upper"), ("cost", "cost"), ("supply", "supply")):
            assert np.array_equal(getattr(a, fa), getattr(b, fb)), (name, fa)
        out = str(tmp_path / (name + ".min"))
        M.write_dimacs(a, out)
        c = M.read_dimacs(out)
        for f in ("source", "target", "lower", "upper", "cost", "supply"):
            assert np.array_equal(getattr(a, f), getattr(c, f))
    with pytest.raises(M.McfError) as ei:
        M.read_dimacs(str(tmp_path / "missing.min"))
    assert ei.value.code == L.ERR_IO
    bad = tmp_path / "bad.min"
    bad.write_text("p min 2 2\na 1 2 0 1 1\n")
    with pytest.raises(M.McfError):
        M.read_dimacs(str(bad))
    empty = tmp_path / "empty.min"
    empty.write_text("c nothing\np min 3 0\nn 1 0\n")
    e = M.read_dimacs(str(empty))
    assert e.arc_count == 0 and e.node_count == 3


def test_shard_ranges_partition_the_search_arcs():
    for m_s in (0, 1, 3, 4, 5, 1023, 40000, 400001):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                b, e = M.shard_range(m_s, r, world)
                assert b == prev and e >= b and b % 4 == 0
                prev = e
            assert prev == m_s
    with pytest.raises(M.McfError):
        M.shard_range(10, 2, 2)


# ------------------------------------------------------------------ block sizing of the plain BlockSearchPivot (NetworkSimplex.cs:1304-1337, :1400-1438)

def _cfg_dict(c):
    d = c.as_dict()
    return d


@pytest.mark.parametrize("name,path,want", fixtures(), ids=[f[0] for f in fixtures()])
def test_auto_configuration_matches_the_oracle_on_every_fixture(name, path, want):
    """mcf_block_config_auto (ProblemAnalyzer + OptimizationSelector) and mcf_block_initial_size against the oracle's own restatement:
    same flags, same initial block size, same 'the reference would use the cached finder' verdict."""
    p = load(path)
    c = M.auto_block_config(p.n, p.src, p.tgt)
    o = O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK, auto_config=True)
    if not o.init():
        pytest.skip("infeasible bounds")
    assert c.flags == o.config_flags
    b, dmin = C.c_int32(), C.c_int32()
    L.check(L.lib().mcf_block_initial_size(C.byref(c), o.search_arc_num, p.n, C.byref(b), C.byref(dmin)))
    assert b.value == o.initial_block_size == o.block_size


def test_block_config_defaults_and_small_blocks():
    c = M.block_config()
    assert _cfg_dict(c) == dict(flags=0, min_block_size=25, max_block_size=100, consecutive_hits_before_adapt=3, min_block_size_ratio=0.125,
                                block_size_growth_factor=1.2, block_size_shrink_factor=0.8, low_hit_rate_threshold=0.05, high_hit_rate_threshold=0.3)
    b, dmin = C.c_int32(), C.c_int32()
    # sqrt(40000) = 200; default config: max(200, max(25, 25)) = 200
    L.check(L.lib().mcf_block_initial_size(C.byref(c), 40000, 10000, C.byref(b), C.byref(dmin)))
    assert (b.value, dmin.value) == (200, 25)
    # SmallBlocksForDense: density 40000 / 1000 = 40 > 10 -> min(50, 200 / 4) = 50
    c = M.block_config(flags=M.OPT_SMALL_BLOCKS_FOR_DENSE)
    L.check(L.lib().mcf_block_initial_size(C.byref(c), 40000, 1000, C.byref(b), C.byref(dmin)))
    assert (b.value, dmin.value) == (50, 25)
    # density exactly 10 is not dense (NetworkSimplex.cs:1318 'density > 10')
    L.check(L.lib().mcf_block_initial_size(C.byref(c), 40000, 4000, C.byref(b), C.byref(dmin)))
    assert b.value == 200
    # tiny search range: the dynamic minimum wins
    L.check(L.lib().mcf_block_initial_size(C.byref(c), 100, 3, C.byref(b), C.byref(dmin)))
    assert (b.value, dmin.value) == (25, 25)


@pytest.mark.parametrize("name", ["netgen_8_10a", "transport_40x30", "circulation_100_0_10", "AURV19V6"])
@pytest.mark.parametrize("cfg", [dict(flags=1), dict(flags=3), dict(flags=1, block_size_shrink_factor=0.7, consecutive_hits_before_adapt=2, min_block_size=10),
                                 dict(flags=1, low_hit_rate_threshold=0.001, high_hit_rate_threshold=0.002, max_block_size=400, block_size_growth_factor=1.3)],
                         ids=["adaptive", "adaptive+small", "fast-shrink", "growing"])
def test_adaptive_block_size_follows_the_oracle(name, cfg):
    """mcf_block_adapt driven with the oracle's per-search arc counts reproduces the oracle's block size after every search of a whole
    solve (the growing configuration makes the 'found quickly' branch fire, which the defaults never reach)."""
    p = load(name)
    o = O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK, config=cfg)
    assert o.init()
    c = M.block_config(**cfg)
    b, dmin = C.c_int32(), C.c_int32()
    L.check(L.lib().mcf_block_initial_size(C.byref(c), o.search_arc_num, p.n, C.byref(b), C.byref(dmin)))
    assert b.value == o.block_size
    counters = (C.c_int32 * 2)(0, 0)
    sizes, before = set(), o.arcs_checked
    for it in range(200000):
        f, e = o.find_entering()
        checked = o.arcs_checked - before
        before = o.arcs_checked
        if not f:
            break
        L.check(L.lib().mcf_block_adapt(C.byref(c), dmin.value, checked, C.byref(b), counters))
        assert b.value == o.block_size, (it, checked)
        sizes.add(b.value)
        o.apply_pivot(e)
    assert len(sizes) > 1 or b.value == dmin.value          # the size did move (unless it started at the dynamic minimum)
    assert o.finish() == O.OPTIMAL


def test_auto_configuration_off_equals_the_default_config():
    p = load("netgen_8_10a")
    a = O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK); a.solve()
    b = O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK, config={}); b.solve()
    c = O.Oracle(p, O.SEM_CSHARP, O.RULE_BLOCK, auto_config=True); st, _ = c.solve()
    assert a.pivots == b.pivots and a.total_cost == b.total_cost == c.total_cost and st == O.OPTIMAL
    assert c.initial_block_size == c.block_size or c.config_flags & 1


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """include/mcf_hip.h compiled as plain C (it is a C ABI) and its struct sizes compared with the ctypes mirrors in _lib.py."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(mcf_engine_stats), sizeof(mcf_ns_metrics),'
                   ' sizeof(mcf_engine_desc), sizeof(mcf_block_config), sizeof(mcf_candidate), sizeof(mcf_validation));return 0;}\n'
                   % os.path.join(ROOT, "include", "mcf_hip.h"))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(t) for t in (L.EngineStats, L.NsMetrics, L.EngineDesc, L.BlockConfig, L.Candidate, L.Validation)]
    assert got == want


def test_new_entry_points_validate_their_arguments():
    """Argument and call-order errors of the round-2 entry points (no GPU involved): the reference's ArgumentException / InvalidOperationException cases."""
    lib = L.lib()
    p = load("transport_2x3")
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
    for call in (lambda: ns.set_sharding_host("no-slash", 0, 2), lambda: ns.set_sharding_host("/x", 2, 2), lambda: ns.set_shard_group([]),
                 lambda: ns.set_shard_group([0, -1]), lambda: ns.tree()):
        with pytest.raises(M.McfError) as ei:
            call()
        assert ei.value.code in (L.ERR_INVALID, L.ERR_STATE)
    assert lib.mcf_ns_set_optimization_config(ns._h, None) == L.ERR_INVALID            # ArgumentNullException, NetworkSimplex.cs:559
    assert ns.begin() == 0
    t = ns.tree()
    assert len(t["parent"]) == p.n + 1 and t["parent"][p.n] == -1 and (t["succ_num"][p.n] == p.n + 1)
    c = M.block_config()
    b, d = C.c_int32(), C.c_int32()
    assert lib.mcf_block_initial_size(None, 10, 10, C.byref(b), C.byref(d)) == L.ERR_INVALID
    assert lib.mcf_block_initial_size(C.byref(c), -1, 10, C.byref(b), C.byref(d)) == L.ERR_INVALID
    assert lib.mcf_block_adapt(C.byref(c), 25, 100, None, None) == L.ERR_INVALID
    with pytest.raises(M.McfError):
        M.auto_block_config(2, [0, 5], [1, 0])                                           # end point outside the node range
    # density of an empty search range over no nodes: C#'s 0.0 / 0 is NaN, 'NaN > 10' is false
    small = M.block_config(flags=M.OPT_SMALL_BLOCKS_FOR_DENSE)
    L.check(lib.mcf_block_initial_size(C.byref(small), 0, 0, C.byref(b), C.byref(d)))
    assert (b.value, d.value) == (25, 25)
    # the adaptive rule leaves the size alone without its flag
    counters = (C.c_int32 * 2)(0, 0)
    b.value = 77
    L.check(lib.mcf_block_adapt(C.byref(c), 25, 1000, C.byref(b), counters))
    assert b.value == 77 and list(counters) == [0, 0]
