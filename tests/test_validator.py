"""SolutionValidator: the numpy restatement against the bundled solutions and hand-checked cases on the CPU,
the device reductions against the restatement on the GPU (bit-exact: counts, first failing ids, objective, dual cost)."""
import numpy as np
import pytest

import kat_data as K
import mincostflow_amd as M
from helpers import INF, fixtures, load, problem_from_dict
from oracle import ns_oracle as O
from oracle import validator as V

KEYS = ("valid", "supply_type", "objective", "dual_cost", "errors", "first")


def _same(dev: dict, ref: dict):
    for k in KEYS:
        assert dev[k] == ref[k], (k, dev[k], ref[k])


# ------------------------------------------------------------------------------------------------ CPU: the restatement itself
@pytest.mark.parametrize("name,path,want", fixtures(), ids=[f[0] for f in fixtures()])
def test_restatement_accepts_every_bundled_optimum(name, path, want):
    """Solve with the oracle (C# semantics), validate like SolutionValidator.Validate(): no message, objective == dual == the .sol cost."""
    p = load(path)
    if p.m > 60000:
        pytest.skip("covered on the GPU; the CPU suite stays short")
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK)
    st, _ = o.solve()
    assert st == O.OPTIMAL
    r = V.validate(p.n, p.src, p.tgt, p.lower, p.upper, p.cost, p.supply, V.GEQ, o.flow(), o.potential(), o.total_cost)
    assert r["valid"] == 1 and r["objective"] == want and r["dual_cost"] == want, r


def test_restatement_on_a_hand_checked_broken_solution():
    # two arcs 0->1 (cost 3, bounds [1, 4]) and 1->2 (cost -2, bounds [0, 5]); supplies 2, 0, -2
    src, tgt = [0, 1], [1, 2]
    lower, upper, cost, supply = [1, 0], [4, 5], [3, -2], [2, 0, -2]
    # a valid optimum: push 2 units along the path; pi chosen so that both reduced costs are 0
    ok = V.validate(3, src, tgt, lower, upper, cost, supply, V.EQ, [2, 2], [0, 3, 1], 2)
    assert ok["valid"] == 1 and ok["objective"] == 2 and ok["dual_cost"] == 2
    # flow 0 on arc 0 (below its lower bound), 6 on arc 1 (above its upper bound), pi all zero
    r = V.validate(3, src, tgt, lower, upper, cost, supply, V.GEQ, [0, 6], [0, 0, 0], 7)
    # net flow: node0 = 0, node1 = 6, node2 = -6; GEQ needs net >= supply: node0 0>=2 fails, node1 ok, node2 -6>=-2 fails
    assert r["errors"]["conservation"] == 2 and r["first"]["conservation"] == 0
    assert r["errors"]["lower"] == 1 and r["first"]["lower"] == 0
    assert r["errors"]["upper"] == 1 and r["first"]["upper"] == 1
    # reduced costs 3 and -2: arc 0 needs flow == lower (0 != 1), arc 1 needs flow == upper (6 != 5)
    assert r["errors"]["slack_pos"] == 1 and r["errors"]["slack_neg"] == 1
    assert r["errors"]["node_dual"] == 0 and r["errors"]["node_slack"] == 0
    assert r["objective"] == -12 and r["errors"]["objective"] == 1
    # dual: lower*cost = 3; adjusted supply * pi = 0; arc 1: -(5 - 0) * 2 = -10  ->  -7
    assert r["dual_cost"] == -7 and r["errors"]["dual_cost"] == 1 and r["valid"] == 0
    # LEQ with a negative potential on node 1 whose balance is off
    r = V.validate(3, src, tgt, lower, upper, cost, supply, V.LEQ, [2, 1], [0, -1, 0], 4)
    assert r["errors"]["node_dual"] == 1 and r["first"]["node_dual"] == 1


def test_restatement_wraps_like_unchecked_long():
    big = 2 ** 62
    r = V.validate(2, [0], [1], [0], [INF], [4], [0, 0], V.GEQ, [big], [0, 0], 0)
    assert r["objective"] == 0          # 4 * 2^62 wraps to 0
    r = V.validate(2, [0], [1], [0], [INF], [3], [0, 0], V.GEQ, [big], [0, 0], 0)
    assert r["objective"] == -(2 ** 62)  # 3 * 2^62 = 2^63 + 2^62 -> -2^63 + 2^62


def test_validator_has_no_cpu_path():
    if M.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(M.McfError) as e:
        M.SolutionValidator(4, 4)
    assert e.value.code == -2


# ------------------------------------------------------------------------------------------------ GPU: device reductions
def _random_case(rng, n, m, with_lower, span):
    src = rng.integers(0, max(n, 1), m, dtype=np.int32)
    tgt = rng.integers(0, max(n, 1), m, dtype=np.int32)
    lower = rng.integers(0, 4, m).astype(np.int64) * (rng.random(m) < 0.3) if with_lower else np.zeros(m, np.int64)
    upper = lower + rng.integers(0, 6, m)
    upper[rng.random(m) < 0.05] = INF
    cost = rng.integers(-span, span + 1, m).astype(np.int64)
    # mostly zero flows like a basic solution, some at a bound, some out of bounds
    flow = np.where(rng.random(m) < 0.7, lower, np.minimum(upper, lower + 9))
    flow = np.where(rng.random(m) < 0.05, flow + rng.integers(-3, 4, m), flow).astype(np.int64)
    pi = rng.integers(-span, span + 1, n).astype(np.int64)
    pi[rng.random(n) < 0.5] = 0
    supply = np.zeros(n, np.int64)
    np.add.at(supply, src, flow); np.subtract.at(supply, tgt, flow)
    supply = np.where(rng.random(n) < 0.1, supply + rng.integers(-2, 3, n), supply).astype(np.int64)
    return src, tgt, lower.astype(np.int64), upper.astype(np.int64), cost, supply, flow, pi


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(1, 0), (2, 1), (7, 63), (64, 64), (65, 257), (300, 1025), (5000, 40001), (100_000, 1_000_003)])
def test_device_validator_matches_restatement_on_random_data(n, m):
    rng = np.random.default_rng(1000 + m)
    for with_lower in (False, True):
        for span in (50, 2 ** 40):                      # the second one makes the 64-bit products wrap
            case = _random_case(rng, n, m, with_lower, span)
            src, tgt, lower, upper, cost, supply, flow, pi = case
            v = M.SolutionValidator(n, m).upload_network(src, tgt, lower, upper, cost, supply).upload_solution(flow, pi)
            for st in (V.GEQ, V.LEQ, V.EQ):
                ref = V.validate(n, src, tgt, lower, upper, cost, supply, st, flow, pi, 12345)
                _same(v.run(st, 12345), ref)
            # reported cost equal to the computed one clears exactly that message
            ref = V.validate(n, src, tgt, lower, upper, cost, supply, V.GEQ, flow, pi, 0)
            dev = v.run(V.GEQ, ref["objective"])
            assert dev["errors"]["objective"] == 0 and dev["objective"] == ref["objective"]


@pytest.mark.gpu
def test_device_validator_keeps_network_and_solution_separately():
    rng = np.random.default_rng(5)
    src, tgt, lower, upper, cost, supply, flow, pi = _random_case(rng, 500, 3000, True, 100)
    v = M.SolutionValidator(500, 3000)
    with pytest.raises(M.McfError):
        v.run(V.GEQ, 0)                                  # nothing uploaded
    v.upload_network(src, tgt, lower, upper, cost, supply)
    with pytest.raises(M.McfError):
        v.run(V.GEQ, 0)                                  # no solution yet
    v.upload_solution(flow, pi)
    a = v.run(V.GEQ, 0)
    flow2 = flow.copy(); flow2[::7] += 1
    v.upload_solution(flow2, pi)
    _same(v.run(V.GEQ, 0), V.validate(500, src, tgt, lower, upper, cost, supply, V.GEQ, flow2, pi, 0))
    v.upload_solution(flow, pi)
    _same(v.run(V.GEQ, 0), {k: a[k] for k in KEYS})
    with pytest.raises(M.McfError):
        v.upload_network(np.full(3000, 500, np.int32), tgt, lower, upper, cost, supply)     # end point out of range
    with pytest.raises(M.McfError):
        v.run(7, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,path,want", fixtures(), ids=[f[0] for f in fixtures()])
def test_every_fixture_solution_validates_on_the_device(name, path, want):
    """Validate() after Solve() on every bundled instance: valid, objective == dual cost == the .sol cost, same as the restatement."""
    p = load(path)
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply).enable_optimized_pivot(True)
    assert ns.solve() == M.SolverStatus.Optimal
    dev = ns.validate()
    assert dev["valid"] == 1 and dev["objective"] == want and dev["dual_cost"] == want, dev
    upper = np.array([ns.get_arc_upper_bound(e) for e in range(p.m)], np.int64) if p.m <= 2000 else p.upper
    ref = V.validate(p.n, p.src, p.tgt, p.lower, upper, p.cost, p.supply, V.GEQ, ns.flows(), ns.potentials(), ns.get_total_cost())
    _same(dev, ref)


@pytest.mark.gpu
def test_validate_mirrors_the_reference_on_its_unit_test_instances():
    for name, d, status, cost, flows in K.CSHARP_KATS:
        p = problem_from_dict(d)
        ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
        st = ns.solve()
        assert st == status
        dev = ns.validate()
        if st != M.SolverStatus.Optimal:                 # SolutionValidator.cs:28-33
            assert dev["valid"] == 0 and dev["errors"]["status"] == 1 and sum(dev["errors"].values()) == 1
            continue
        upper = np.array([ns.get_arc_upper_bound(e) for e in range(p.m)], np.int64)
        ref = V.validate(p.n, p.src, p.tgt, p.lower, upper, p.cost, p.supply, V.GEQ, ns.flows(), ns.potentials(), ns.get_total_cost())
        _same(dev, ref)
        if cost is not None:
            assert dev["objective"] == cost
        if not np.any(p.lower):
            assert dev["valid"] == 1, (name, dev)


@pytest.mark.gpu
def test_validator_at_full_size():
    """Config 5's shape (1M nodes / 8M arcs): a solution-like random vector, totals against numpy; the run is timed for the record."""
    rng = np.random.default_rng(8)
    n, m = 1_000_000, 8_000_000
    src, tgt, lower, upper, cost, supply, flow, pi = _random_case(rng, n, m, False, 10_000)
    v = M.SolutionValidator(n, m).upload_network(src, tgt, lower, upper, cost, supply).upload_solution(flow, pi)
    dev = v.run(V.GEQ, 0)
    ref = V.validate(n, src, tgt, lower, upper, cost, supply, V.GEQ, flow, pi, 0)
    _same(dev, ref)
    assert dev["kernel_us"] > 0 and dev["algorithmic_bytes"] == 40 * (n + m)


# ------------------------------------------------------------------------------------------------ .sol files (SolutionLoader.cs)
def _sol_fixtures():
    import os
    return [(n, p, w) for n, p, w in fixtures() if os.path.exists(p[:-4] + ".sol")]


@pytest.mark.parametrize("name,path,want", _sol_fixtures(), ids=[f[0] for f in _sol_fixtures()])
def test_bundled_solutions_certify_the_oracle_potentials(name, path, want):
    """The reference's .sol files hold an independent optimal flow (Gurobi).  Complementary slackness holds between ANY optimal
    flow and ANY optimal dual, so their flows + the oracle's potentials must pass every check, with objective = dual = the s line."""
    p = load(path)
    if p.m > 60000:
        pytest.skip("covered on the GPU; the CPU suite stays short")
    g = M.read_dimacs(path)
    sol = M.read_solution(path[:-4] + ".sol", g)
    assert sol["cost"] == want and sol["pi"] is None
    o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK)
    assert o.solve()[0] == O.OPTIMAL
    r = V.validate(p.n, p.src, p.tgt, p.lower, p.upper, p.cost, p.supply, V.GEQ, sol["flow"], o.potential(), sol["cost"])
    assert r["valid"] == 1 and r["objective"] == want and r["dual_cost"] == want, r


def test_solution_files_round_trip(tmp_path):
    g = M.Problem(3, 4, np.array([0, 0, 0, 1], np.int32), np.array([1, 1, 2, 2], np.int32), np.array([0, 1, 0, 0], np.int64),
                  np.array([5, 4, 9, 9], np.int64), np.array([7, 2, 1, 1], np.int64), np.array([6, 0, -6], np.int64))
    path = str(tmp_path / "a.sol")
    M.write_solution(path, -12, [0, 3, 0, 4], [0, -2, 5])
    assert open(path).read().split("\n")[:3] == ["s -12", "f 1 3", "f 3 4"]            # SaveToFile: s line, non-zero flows by arc id
    r = M.read_solution(path, g)
    assert r["cost"] == -12 and list(r["flow"]) == [0, 3, 0, 4] and list(r["pi"]) == [0, -2, 5]
    # end-point format, parallel arcs 0 -> 1: lower bounds first (arc 1 has lower 1), then the cheaper arc (arc 1, cost 2) up to its capacity
    with open(path, "w") as f:
        f.write("c Solution file generated from Gurobi output\ns 40\nf 1 2 6\nf 2 3 4\n")
    r = M.read_solution(path, g)
    assert r["cost"] == 40 and list(r["flow"]) == [2, 4, 0, 4] and r["pi"] is None
    with open(path, "w") as f:
        f.write("f 3 1 2\n")
    with pytest.raises(M.McfError):
        M.read_solution(path, g)                                                        # 3 -> 1 is not an arc
    with open(path, "w") as f:
        f.write("f 1 2 1\n")
    r = M.read_solution(path, g)
    assert r["cost"] is None and list(r["flow"]) == [0, 1, 0, 0]
    with pytest.raises(M.McfError):
        M.read_solution(str(tmp_path / "missing.sol"), g)


@pytest.mark.gpu
@pytest.mark.parametrize("name,path,want", _sol_fixtures(), ids=[f[0] for f in _sol_fixtures()])
def test_bundled_solutions_certify_the_device_potentials(name, path, want):
    """Same certificate end to end on the device: Gurobi's flow from the .sol file + the potentials of the GPU solve through the
    device validator."""
    g = M.read_dimacs(path)
    sol = M.read_solution(path[:-4] + ".sol", g)
    ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(M.PivotRule.BestEligible if g.arc_count <= 20000 else M.PivotRule.BlockSearch)
    assert ns.enable_optimized_pivot(True).solve() == M.SolverStatus.Optimal
    v = M.SolutionValidator(g.node_count, g.arc_count).upload_network(g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    r = v.upload_solution(sol["flow"], ns.potentials()).run(V.GEQ, sol["cost"])
    assert r["valid"] == 1 and r["objective"] == want and r["dual_cost"] == want, r
