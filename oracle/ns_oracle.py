"""ctypes binding of oracle/libns_oracle.so  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of oracle/ns_oracle.c).  Nothing in mincostflow_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libns_oracle.so")

SEM_LEMON, SEM_CSHARP, SEM_CSHARP_OPT = 0, 1, 2
RULE_FIRST, RULE_BEST, RULE_BLOCK = 0, 1, 2          # Types/PivotRule.cs
GEQ, LEQ = 0, 1                                       # Types/SupplyType.cs
NOT_SOLVED, OPTIMAL, INFEASIBLE, UNBOUNDED = 0, 1, 2, 3   # Types/SolverStatus.cs
INF_CAP = np.iinfo(np.int64).max

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ns_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libns_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.nso_create.restype = C.c_void_p
    L.nso_create.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i64p, _i64p, _i64p, _i64p,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.nso_destroy.argtypes = [C.c_void_p]
    L.nso_init.argtypes = [C.c_void_p]
    L.nso_initial_pivots.argtypes = [C.c_void_p]
    L.nso_find_entering.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.nso_apply_pivot.argtypes = [C.c_void_p, C.c_int32]
    L.nso_finish.argtypes = [C.c_void_p]
    L.nso_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.nso_run_pivots.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.nso_set_config.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    L.nso_set_auto_config.argtypes = [C.c_void_p, C.c_int]
    L.nso_set_vector_width.argtypes = [C.c_void_p, C.c_int]
    L.nso_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.nso_phase_us.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    for f in ("nso_status", "nso_search_arc_num", "nso_all_arc_num", "nso_block_size", "nso_next_arc",
              "nso_last_subtree", "nso_initial_block_size", "nso_config_flags", "nso_would_cache"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int
    for f in ("nso_pivots", "nso_init_pivot_count", "nso_art_cost", "nso_total_cost", "nso_last_sigma", "nso_arcs_checked"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int64
    L.nso_get_flow.argtypes = [C.c_void_p, _i64p]
    L.nso_get_potential.argtypes = [C.c_void_p, _i64p]
    L.nso_get_arc_id.argtypes = [C.c_void_p, _i32p]
    for f, t in (("nso_src", C.c_int32), ("nso_tgt", C.c_int32), ("nso_cost", C.c_int64),
                 ("nso_state", C.c_int8), ("nso_pi", C.c_int64), ("nso_flow_internal", C.c_int64),
                 ("nso_thread", C.c_int32), ("nso_parent", C.c_int32)):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.POINTER(t)
    L.nso_scan_best.argtypes = [C.c_int, _i8p, _i64p, _i32p, _i32p, _i64p,
                                C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.nso_scan_first.argtypes = [C.c_int, _i8p, _i64p, _i32p, _i32p, _i64p,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.nso_scan_block.argtypes = [C.c_int, _i8p, _i64p, _i32p, _i32p, _i64p, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    _lib = L
    return L


@dataclass
class Problem:
    """A min-cost-flow instance in the caller's (DIMACS / C#) numbering, 0-based."""
    n: int
    m: int
    src: np.ndarray      # int32[m]
    tgt: np.ndarray      # int32[m]
    lower: np.ndarray    # int64[m]
    upper: np.ndarray    # int64[m]; INF_CAP = no bound
    cost: np.ndarray     # int64[m]
    supply: np.ndarray   # int64[n]

    def __post_init__(self):
        self.src = np.ascontiguousarray(self.src, np.int32)
        self.tgt = np.ascontiguousarray(self.tgt, np.int32)
        self.lower = np.ascontiguousarray(self.lower, np.int64)
        self.upper = np.ascontiguousarray(self.upper, np.int64)
        self.cost = np.ascontiguousarray(self.cost, np.int64)
        self.supply = np.ascontiguousarray(self.supply, np.int64)
        assert self.src.shape == (self.m,) and self.supply.shape == (self.n,)


class Oracle:
    """One network-simplex run of the CPU restatement."""

    def __init__(self, p: Problem, semantics=SEM_CSHARP, rule=RULE_BLOCK, supply_type=GEQ,
                 arc_mixing=True, block_size=0, auto_config=False, config=None, vector_width=4):
        """vector_width: Vector<long>.Count of the machine the reference runs on (BlockSearchPivotOptimized.cs:74, :115): 4 on x64,
        2 on NEON, 0 = Vector.IsHardwareAccelerated is false.  Only SEM_CSHARP_OPT + RULE_BLOCK reads it.
        auto_config: the reference's SetAutoConfiguration (its default is ON; the oracle's default is OFF so that the
        plain rule runs with `new OptimizationConfig()`).  config: dict of OptimizationConfig fields (SetOptimizationConfig)."""
        self.p = p
        self.L = lib()
        self.h = self.L.nso_create(p.n, p.m, p.src, p.tgt, p.lower, p.upper, p.cost, p.supply,
                                   semantics, rule, supply_type, int(arc_mixing), block_size)
        self.semantics, self.rule = semantics, rule
        if not self.L.nso_set_vector_width(self.h, int(vector_width)):
            raise ValueError(f"vector_width {vector_width} is not one of 0, 2, 4, 8")
        if config is not None:
            c = dict(flags=0, min_block_size=25, max_block_size=100, consecutive_hits_before_adapt=3, min_block_size_ratio=0.125,
                     block_size_growth_factor=1.2, block_size_shrink_factor=0.8, low_hit_rate_threshold=0.05, high_hit_rate_threshold=0.3)
            c.update(config)
            self.L.nso_set_config(self.h, c["flags"], c["min_block_size"], c["max_block_size"], c["consecutive_hits_before_adapt"],
                                  c["min_block_size_ratio"], c["block_size_growth_factor"], c["block_size_shrink_factor"],
                                  c["low_hit_rate_threshold"], c["high_hit_rate_threshold"])
        if auto_config:
            self.L.nso_set_auto_config(self.h, 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nso_destroy(self.h)
            self.h = None

    # stepwise
    def init(self) -> bool:
        return bool(self.L.nso_init(self.h))

    def initial_pivots(self) -> bool:
        return bool(self.L.nso_initial_pivots(self.h))

    def find_entering(self):
        e = C.c_int32(-1)
        found = self.L.nso_find_entering(self.h, C.byref(e))
        return (True, e.value) if found else (False, -1)

    def apply_pivot(self, arc: int) -> int:
        return self.L.nso_apply_pivot(self.h, arc)

    def finish(self) -> int:
        return self.L.nso_finish(self.h)

    def run_pivots(self, max_pivots: int):
        """(ended, pivots_done) -- at most max_pivots iterations of the main loop."""
        n = C.c_int64(0)
        ended = self.L.nso_run_pivots(self.h, max_pivots, C.byref(n))
        return bool(ended), n.value

    # whole solve; returns (status, trace[int32]) -- trace = entering arcs of the main loop
    def solve(self, trace_cap: int = 0):
        n = C.c_int64(0)
        tr = np.empty(max(trace_cap, 1), np.int32)
        st = self.L.nso_solve(self.h, tr.ctypes.data if trace_cap else None, trace_cap, C.byref(n))
        self.n_pivots = n.value
        return st, tr[: min(n.value, trace_cap)]

    # results
    @property
    def status(self): return self.L.nso_status(self.h)
    @property
    def total_cost(self): return self.L.nso_total_cost(self.h)
    @property
    def search_arc_num(self): return self.L.nso_search_arc_num(self.h)
    @property
    def all_arc_num(self): return self.L.nso_all_arc_num(self.h)
    @property
    def block_size(self): return self.L.nso_block_size(self.h)
    @property
    def initial_block_size(self): return self.L.nso_initial_block_size(self.h)
    @property
    def config_flags(self): return self.L.nso_config_flags(self.h)
    @property
    def would_cache(self): return bool(self.L.nso_would_cache(self.h))
    @property
    def arcs_checked(self): return self.L.nso_arcs_checked(self.h)
    @property
    def next_arc(self): return self.L.nso_next_arc(self.h)
    @property
    def pivots(self): return self.L.nso_pivots(self.h)
    @property
    def art_cost(self): return self.L.nso_art_cost(self.h)
    @property
    def last_subtree(self): return self.L.nso_last_subtree(self.h)
    @property
    def last_sigma(self): return self.L.nso_last_sigma(self.h)

    def enable_timing(self, on=True):
        self.L.nso_enable_timing(self.h, int(on))

    def phase_us(self):
        """(pivot search, tree update, potential update) in microseconds: the reference's SolverMetrics buckets."""
        out = (C.c_double * 3)()
        self.L.nso_phase_us(self.h, out)
        return tuple(out)

    def flow(self):
        out = np.empty(self.p.m, np.int64); self.L.nso_get_flow(self.h, out); return out

    def potential(self):
        out = np.empty(self.p.n, np.int64); self.L.nso_get_potential(self.h, out); return out

    def arc_id(self):
        out = np.empty(self.p.m, np.int32); self.L.nso_get_arc_id(self.h, out); return out

    def _view(self, fn, count, dtype):
        ptr = getattr(self.L, fn)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)

    def internal_arrays(self):
        """Copies of the SoA the engine must hold: src, tgt, cost, state over all arcs; pi over n+1."""
        a = self.p.m + 2 * self.p.n
        return dict(src=self._view("nso_src", a, np.int32), tgt=self._view("nso_tgt", a, np.int32),
                    cost=self._view("nso_cost", a, np.int64), state=self._view("nso_state", a, np.int8),
                    pi=self._view("nso_pi", self.p.n + 1, np.int64))

    def thread(self): return self._view("nso_thread", self.p.n + 1, np.int32)
    def parent(self): return self._view("nso_parent", self.p.n + 1, np.int32)


# raw scans on bare arrays --------------------------------------------------------------------

def _prep(state, cost, src, tgt, pi):
    return (np.ascontiguousarray(state, np.int8), np.ascontiguousarray(cost, np.int64),
            np.ascontiguousarray(src, np.int32), np.ascontiguousarray(tgt, np.int32),
            np.ascontiguousarray(pi, np.int64))


def scan_best(m_s, state, cost, src, tgt, pi):
    a, c = C.c_int32(-1), C.c_int64(0)
    f = lib().nso_scan_best(m_s, *_prep(state, cost, src, tgt, pi), C.byref(a), C.byref(c))
    return bool(f), a.value, c.value


def scan_first(m_s, state, cost, src, tgt, pi, next_arc):
    a, c, na = C.c_int32(-1), C.c_int64(0), C.c_int32(next_arc)
    f = lib().nso_scan_first(m_s, *_prep(state, cost, src, tgt, pi), C.byref(na), C.byref(a), C.byref(c))
    return bool(f), a.value, c.value, na.value


def scan_block(m_s, state, cost, src, tgt, pi, block_size, optimized, next_arc, vector_width=4):
    a, c, na = C.c_int32(-1), C.c_int64(0), C.c_int32(next_arc)
    f = lib().nso_scan_block(m_s, *_prep(state, cost, src, tgt, pi), block_size, int(optimized), int(vector_width),
                             C.byref(na), C.byref(a), C.byref(c))
    return bool(f), a.value, c.value, na.value
