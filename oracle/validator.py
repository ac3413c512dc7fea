"""CPU restatement of the reference's SolutionValidator -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(mincostflow_amd) never does.

Follows src/MinCostFlow.Core/Lemon/Validation/SolutionValidator.cs of the reference:
  ValidateFlowConservation        :55-100
  ValidateCapacityConstraints     :102-125
  ValidateComplementarySlackness  :135-228
  ValidateObjectiveValue          :230-264
  ValidateDualCost                :270-340
All arithmetic is C# `long` in an unchecked context, i.e. it wraps; numpy int64 wraps the same way.
Instead of the reference's list of message strings the result holds, per check, how many messages the
reference would add and the lowest arc / node id among them (the reference walks ids upwards, so that is
the id in its first message of the kind).

Pinning: the checks are exercised on every bundled fixture solution (all must pass, objective = the `s`
line of the .sol file) and on the reference's own unit-test instances (tests/kat_data.py).
"""
from __future__ import annotations

import numpy as np

KINDS = ("conservation", "lower", "upper", "slack_pos", "slack_neg", "node_dual", "node_slack", "objective", "dual_cost", "status")
GEQ, LEQ, EQ = 0, 1, 2          # SupplyType.Geq, SupplyType.Leq, "anything else" (SolutionValidator.cs:79-84)


def _first(mask: np.ndarray) -> int:
    idx = np.flatnonzero(mask)
    return int(idx[0]) if idx.size else -1


def validate(node_count, source, target, lower, upper, cost, supply, supply_type, flow, pi, reported_cost) -> dict:
    src = np.asarray(source, np.int64)
    tgt = np.asarray(target, np.int64)
    lower = np.asarray(lower, np.int64)
    upper = np.asarray(upper, np.int64)
    cost = np.asarray(cost, np.int64)
    supply = np.asarray(supply, np.int64)
    flow = np.asarray(flow, np.int64)
    pi = np.asarray(pi, np.int64)
    n = int(node_count)
    errors = dict.fromkeys(KINDS, 0)
    first = dict.fromkeys(KINDS, -1)

    def record(kind, mask):
        errors[kind] = int(np.count_nonzero(mask))
        first[kind] = _first(mask)

    with np.errstate(over="ignore"):
        # :62-72 net flow per node
        net = np.zeros(n, np.int64)
        np.add.at(net, src, flow)
        np.subtract.at(net, tgt, flow)
        # :75-99
        if supply_type == GEQ:
            ok = net >= supply
        elif supply_type == LEQ:
            ok = net <= supply
        else:
            ok = net == supply
        record("conservation", ~ok)
        # :104-124
        record("lower", flow < lower)
        record("upper", flow > upper)
        # :146-177
        rc = cost + pi[src] - pi[tgt]
        record("slack_pos", (rc > 0) & (flow != lower))
        record("slack_neg", (rc < 0) & (flow != upper))
        # :193-227
        if supply_type == GEQ:
            record("node_dual", pi > 0)
            record("node_slack", (pi < 0) & (net != supply))
        elif supply_type == LEQ:
            record("node_dual", pi < 0)
            record("node_slack", (pi > 0) & (net != supply))
        # :232-263
        objective = int(np.sum(flow * cost, dtype=np.int64))
        if objective != int(np.int64(reported_cost)):
            errors["objective"], first["objective"] = 1, 0
        # :276-339
        adjusted = supply.copy()
        nz = lower != 0
        dual = np.int64(np.sum(lower[nz] * cost[nz], dtype=np.int64))
        np.subtract.at(adjusted, src[nz], lower[nz])
        np.add.at(adjusted, tgt[nz], lower[nz])
        dual = dual - np.sum(adjusted * pi, dtype=np.int64)
        neg = rc < 0
        dual = dual - np.sum((upper[neg] - lower[neg]) * (-rc[neg]), dtype=np.int64)
        dual = int(np.int64(dual))
        if dual != int(np.int64(reported_cost)):
            errors["dual_cost"], first["dual_cost"] = 1, 0
    return {"valid": int(all(v == 0 for v in errors.values())), "supply_type": int(supply_type), "objective": objective,
            "dual_cost": dual, "errors": errors, "first": first}
