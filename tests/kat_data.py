"""Known-answer data harvested from the reference's own tests (inputs and expected outputs only).

Sources (relative to the reference repository):
  * src/MinCostFlow.Tests/Lemon/NetworkSimplexTests.cs:29-249   (C# unit tests with exact flows)
  * src/MinCostFlow.Tests/Lemon/OptimizationTests.cs:14-69,:75-120,:124-145,:177-193
  * lemon-1.3.1/test/min_cost_flow_test.cc:40-113 (networks), :329-422 (21-case table)
  * README.md:42-53 (the 4-node example; infeasible as written)
"""
import numpy as np

INF = np.iinfo(np.int64).max


def _p(n, arcs, supply, cost, lower=None, upper=None):
    m = len(arcs)
    return dict(n=n, m=m, src=[a[0] for a in arcs], tgt=[a[1] for a in arcs],
                lower=list(lower) if lower is not None else [0] * m,
                upper=list(upper) if upper is not None else [INF] * m,
                cost=list(cost), supply=list(supply))


# ---- C# unit tests: (name, problem, expected status, expected cost, expected flows or None)
CSHARP_KATS = [
    ("SimpleTransportationProblem", _p(4, [(0, 2), (0, 3), (1, 2), (1, 3)], [10, 15, -12, -13], [3, 5, 4, 2]),
     1, 64, [10, 0, 2, 13]),                                              # NetworkSimplexTests.cs:29-79
    ("MinimumCostCirculation", _p(3, [(0, 1), (1, 2), (2, 0)], [0, 0, 0], [2, 3, -6], upper=[10, 10, 10]),
     1, -10, [10, 10, 10]),                                               # :82-127
    ("ExcessSupply_FeasibleWithGEQ", _p(3, [(0, 1), (0, 2), (1, 2)], [10, 0, -10], [3, 1, 2]),
     1, 10, [0, 10, 0]),                                                  # :130-180
    ("LowerBounds_Respected", _p(2, [(0, 1)], [10, -10], [1], lower=[5], upper=[15]),
     1, 10, [10]),                                                        # :183-206
    ("ComplementarySlackness", _p(3, [(0, 1), (1, 2)], [10, 0, -10], [1, 1], upper=[20, 20]),
     1, 20, [10, 10]),                                                    # :209-249
    ("OptimizedPivot_ProducesSameResults",                                # OptimizationTests.cs:14-69,:124-145
     _p(5, [(0, 1), (0, 2), (1, 3), (2, 3), (2, 4), (3, 4)], [50, 20, -10, -30, -30], [10, 20, 30, 15, 25, 35],
        upper=[100] * 6), 1, None, None),
    ("AllOptimizations_Grid",                                             # OptimizationTests.cs:75-120,:177-193
     _p(8, [(0, 1), (1, 2), (2, 3), (0, 4), (1, 5), (2, 6), (3, 7), (4, 5), (5, 6), (6, 7)],
        [100, 0, 0, 0, 0, 0, 0, -100], [(i + 1) * 10 for i in range(10)], upper=[100] * 10), 1, None, None),
    # README.md:42-53: supply 15 but the max s-t flow is 5 + 8 = 13 -> Infeasible; with 13 the optimum is 71
    ("README_literal", _p(4, [(0, 1), (0, 2), (1, 3), (2, 3)], [15, 0, 0, -15], [2, 3, 1, 4], upper=[10, 8, 5, 10]),
     2, None, None),
    ("README_supply13", _p(4, [(0, 1), (0, 2), (1, 3), (2, 3)], [13, 0, 0, -13], [2, 3, 1, 4], upper=[10, 8, 5, 10]),
     1, 71, [5, 8, 5, 8]),
]

# ---- LEMON min_cost_flow_test.cc networks (1-based labels in the file -> 0-based here)
_GR_ARCS = [(1, 2), (1, 3), (1, 4), (2, 8), (3, 5), (4, 6), (4, 7), (4, 8), (5, 7), (5, 11), (6, 3), (6, 9), (6, 10),
            (7, 1), (8, 12), (9, 12), (10, 12), (10, 2), (10, 7), (11, 10), (12, 11)]
_GR = [(a - 1, b - 1) for a, b in _GR_ARCS]
_C = [70, 150, 80, 80, 140, 60, 80, 110, 60, 120, 0, 140, 90, 30, 60, 50, 70, 100, 60, 20, 30]
_U = [11, 3, 15, 12, 5, 10, 2, 3, 14, 12, 3, 4, 8, 5, 16, 6, 13, 7, 10, 14, 10]
_L1 = [0] * 21
_L2 = [8, 1, 2, 0, 3, 1, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 5, 0, 0, 6, 0]
_L3 = [8, 0, 2, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, -5, 3, 0, 2, 0, -3, -20, -10]
_S1 = [20, -4, 0, 0, 9, -6, 0, 0, 3, -2, 0, -20]
_S2 = [27, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -27]          # stSupply(v=1, w=12, 27)
_S3 = [0] * 12
_S4 = [30, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -30]
_S5 = [20, -8, 0, 0, 6, -5, 0, 0, 0, -7, -10, -30]
_S6 = [30, -3, 0, 0, 11, -6, 0, 3, 0, -2, 0, -20]
_CC = [1] * 21
_CU = [INF] * 21

_NEG1 = [(0, 1), (0, 2), (1, 3), (2, 3), (2, 1), (4, 2), (4, 5), (5, 6), (6, 4)]
_NEG1_C = [100, 30, 20, 80, 50, 10, 80, 30, -120]
_NEG1_L1 = [0] * 9
_NEG1_L2 = [0, 0, 0, 0, 0, 0, 1000, -1000, 0]
_NEG1_S = [100, 0, 0, -100, 0, 0, 0]
_NEG2 = [(0, 1)]

GEQ, LEQ = 0, 1
# (id, problem, supply_type, expected status, expected cost)   statuses: 1 optimal, 2 infeasible, 3 unbounded
LEMON_TABLE = [
    (1, _p(12, _GR, _S1, _C, _L1, _U), GEQ, 1, 5240),
    (2, _p(12, _GR, _S2, _C, _L1, _U), GEQ, 1, 7620),
    (3, _p(12, _GR, _S1, _C, _L2, _U), GEQ, 1, 5970),
    (4, _p(12, _GR, _S2, _C, _L2, _U), GEQ, 1, 8010),
    (5, _p(12, _GR, _S1, _CC, _L1, _CU), GEQ, 1, 74),
    (6, _p(12, _GR, _S2, _CC, _L2, _CU), GEQ, 1, 94),
    (7, _p(12, _GR, _S3, _CC, _L1, _CU), GEQ, 1, 0),
    (8, _p(12, _GR, _S3, _CC, _L2, _U), GEQ, 2, 0),
    (9, _p(12, _GR, _S4, _C, _L3, _U), GEQ, 1, 6360),
    (10, _p(12, _GR, _S5, _C, _L1, _U), GEQ, 1, 3530),
    (11, _p(12, _GR, _S5, _C, _L2, _U), GEQ, 1, 4540),
    (12, _p(12, _GR, _S6, _C, _L2, _U), GEQ, 2, 0),
    (13, _p(7, _NEG1, _NEG1_S, _NEG1_C, _NEG1_L1, [INF] * 9), GEQ, 3, 0),
    (14, _p(7, _NEG1, _NEG1_S, _NEG1_C, _NEG1_L1, [5000] * 9), GEQ, 1, -40000),
    (15, _p(7, _NEG1, _NEG1_S, _NEG1_C, _NEG1_L2, [INF] * 9), GEQ, 3, 0),
    (16, _p(2, _NEG2, [100, -300], [-1], [0], [INF]), GEQ, 1, -300),      # NetworkSimplex: full negative cost support
    (18, _p(2, _NEG2, [100, -300], [-1], [0], [1000]), GEQ, 1, -300),
    (19, _p(12, _GR, _S6, _C, _L1, _U), LEQ, 1, 5080),
    (20, _p(12, _GR, _S6, _C, _L2, _U), LEQ, 1, 5930),
    (21, _p(12, _GR, _S5, _C, _L2, _U), LEQ, 2, 0),
]

# optimal costs of the bundled fixtures that ship without a .sol file
# (src/MinCostFlow.Benchmarks/benchmarks/performance_comparison.csv:118,:126)
EXTRA_COSTS = {"netgen_8_10a": 369269289, "netgen_8_14a": 1772056888}
