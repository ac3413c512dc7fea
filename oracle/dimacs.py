"""DIMACS .min / .sol readers for the tests  --  TEST INFRASTRUCTURE ONLY.

Format (src/MinCostFlow.Problems/Loaders/DimacsReader.cs:36-147, lemon/dimacs.h:129-186):
  p min N M / n id supply / a u v low cap cost   (1-based ids)
.sol (Loaders/SolutionLoader.cs): s cost / f u v flow / p node potential
"""
from __future__ import annotations

import gzip
import os

import numpy as np

from .ns_oracle import INF_CAP, Problem


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def read_min(path: str, lemon_caps: bool = False) -> Problem:
    """lemon_caps=True applies lemon/dimacs.h:178-181 (cap < low => infinite); the C# reader keeps
    the value (DimacsReader.cs:84-91) and Solve() then reports Infeasible (NetworkSimplex.cs:624-634)."""
    n = m = 0
    supply = None
    src, tgt, low, cap, cost = [], [], [], [], []
    with _open(path) as f:
        for line in f:
            if not line or line[0] in "c\n":
                continue
            t = line.split()
            if not t:
                continue
            if t[0] == "p":
                n, m = int(t[2]), int(t[3])
                supply = np.zeros(n, np.int64)
            elif t[0] == "n":
                supply[int(t[1]) - 1] = int(t[2])
            elif t[0] == "a":
                src.append(int(t[1]) - 1); tgt.append(int(t[2]) - 1)
                lo, up = int(t[3]), int(t[4])
                if lemon_caps and up < lo:
                    up = INF_CAP
                low.append(lo); cap.append(up); cost.append(int(t[5]))
    assert len(src) == m, f"{path}: header says {m} arcs, found {len(src)}"
    return Problem(n, m, np.array(src, np.int32), np.array(tgt, np.int32), np.array(low, np.int64),
                   np.array(cap, np.int64), np.array(cost, np.int64), supply)


def read_sol_cost(path: str):
    with _open(path) as f:
        for line in f:
            if line.startswith("s "):
                return int(line.split()[1])
    return None


def read_sol(path: str):
    """Returns (cost, flows[(u, v, flow)], potentials{node: pot}) with 0-based nodes."""
    cost, flows, pots = None, [], {}
    with _open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "s":
                cost = int(t[1])
            elif t[0] == "f":
                flows.append((int(t[1]) - 1, int(t[2]) - 1, int(t[3])))
            elif t[0] == "p" and len(t) == 3:
                pots[int(t[1]) - 1] = int(t[2])
    return cost, flows, pots


def write_min(path: str, p: Problem, comment: str = ""):
    with open(path, "w") as f:
        if comment:
            for ln in comment.splitlines():
                f.write(f"c {ln}\n")
        f.write(f"p min {p.n} {p.m}\n")
        for v in range(p.n):
            if p.supply[v] != 0:
                f.write(f"n {v + 1} {int(p.supply[v])}\n")
        for e in range(p.m):
            f.write(f"a {int(p.src[e]) + 1} {int(p.tgt[e]) + 1} {int(p.lower[e])} {int(p.upper[e])} {int(p.cost[e])}\n")


GOLDEN_DIMACS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "tests", "golden", "dimacs")
