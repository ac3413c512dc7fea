"""Shared test helpers: fixture loading, oracle construction and the reference's validator checks."""
import glob
import os

import numpy as np

from oracle import ns_oracle as O
from oracle.dimacs import GOLDEN_DIMACS, read_min, read_sol_cost

INF = np.iinfo(np.int64).max


def problem_from_dict(d) -> O.Problem:
    return O.Problem(d["n"], d["m"], np.array(d["src"], np.int32), np.array(d["tgt"], np.int32),
                     np.array(d["lower"], np.int64), np.array(d["upper"], np.int64), np.array(d["cost"], np.int64),
                     np.array(d["supply"], np.int64))


def fixtures():
    """[(name, path, expected_cost or None)] for every bundled DIMACS instance."""
    from kat_data import EXTRA_COSTS
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIMACS, "*", "*.min"))):
        name = os.path.basename(path)[:-4]
        sol = path[:-4] + ".sol"
        want = read_sol_cost(sol) if os.path.exists(sol) else EXTRA_COSTS.get(name)
        out.append((name, path, want))
    return out


def load(name_or_path, lemon_caps=False) -> O.Problem:
    if os.path.exists(name_or_path):
        return read_min(name_or_path, lemon_caps)
    hits = glob.glob(os.path.join(GOLDEN_DIMACS, "*", name_or_path + ".min"))
    assert hits, name_or_path
    return read_min(hits[0], lemon_caps)


def validate_solution(p: O.Problem, flow, pi, supply_type=O.GEQ, check_dual=True):
    """The checks of src/MinCostFlow.Core/Lemon/Validation/SolutionValidator.cs:55-342 (mirroring LEMON's
    checkFlow/checkPotential/checkDualCost, lemon-1.3.1/test/min_cost_flow_test.cc:205-299), for the C# solver's
    sign convention: reduced cost = cost + pi[source] - pi[target]."""
    flow = np.asarray(flow, np.int64); pi = np.asarray(pi, np.int64)
    up = np.where(p.upper == INF, INF // 2, p.upper)
    # bounds (:102-123)
    assert np.all(flow >= p.lower), "flow below lower bound"
    assert np.all(flow <= up), "flow above upper bound"
    # conservation (:55-100)
    net = np.zeros(p.n, np.int64)
    np.add.at(net, p.src, flow); np.subtract.at(net, p.tgt, flow)
    if supply_type == O.GEQ:
        assert np.all(net >= p.supply), "conservation (GEQ) violated"
    else:
        assert np.all(net <= p.supply), "conservation (LEQ) violated"
    balanced = int(p.supply.sum()) == 0
    if balanced:
        assert np.array_equal(net, p.supply), "conservation (balanced) violated"
    # complementary slackness (:133-229)
    rc = p.cost + pi[p.src] - pi[p.tgt]
    assert np.all((rc <= 0) | (flow == p.lower)), "rc > 0 but flow above lower bound"
    assert np.all((rc >= 0) | (flow == up)), "rc < 0 but flow below upper bound"
    # node duals sign (:198-211): GEQ -> pi <= 0, slack only where pi == 0
    if supply_type == O.GEQ:
        assert np.all(pi <= 0), "positive potential under GEQ"
        assert np.all((net == p.supply) | (pi == 0))
    else:
        assert np.all(pi >= 0)
        assert np.all((net == p.supply) | (pi == 0))
    primal = int((flow.astype(object) * p.cost.astype(object)).sum())
    if check_dual:
        # dual objective == primal (:265-342); finite capacities only
        red = np.maximum(-rc, 0).astype(object)
        finite = p.upper != INF
        assert np.all((rc >= 0) | finite), "negative reduced cost on an uncapacitated arc"
        dual = -int((pi.astype(object) * p.supply.astype(object)).sum())
        dual -= int((red[finite] * up[finite].astype(object)).sum())
        dual += int((np.maximum(rc, 0).astype(object) * p.lower.astype(object)).sum())
        assert dual == primal, (dual, primal)
    return primal
