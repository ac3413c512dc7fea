"""Python mirror of the reference's solver surface over the C ABI.

`NetworkSimplex` keeps the method names and argument meaning of the reference class
(src/MinCostFlow.Core/Lemon/Algorithms/NetworkSimplex.cs:153-587): set_arc_bounds / set_arc_cost /
set_node_supply / set_supply_type / set_pivot_rule / enable_optimized_pivot / solve / get_flow /
get_potential / get_total_cost / status / get_metrics.  Everything is forwarded to libmcf_hip.so; the
entering-arc search and the potential update run on the MI355X, nothing is computed in Python.

`PivotEngine` is the bare device seam (`IFindEnteringArc`, NetworkSimplex.cs:1286-1289) for hosts that
own their spanning tree.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from ._lib import McfError  # noqa: F401  (re-export)


class PivotRule:                      # Types/PivotRule.cs
    FirstEligible, BestEligible, BlockSearch, CandidateList, AlteringList = 0, 1, 2, 3, 4


class SupplyType:                     # Types/SupplyType.cs
    Geq, Leq = 0, 1


class SolverStatus:                   # Types/SolverStatus.cs
    NotSolved, Optimal, Infeasible, Unbounded, Unbalanced = 0, 1, 2, 3, 4


def _vector_width_abi(v: int) -> int:
    """Python side: 0 = not hardware accelerated, 2 / 4 / 8 = Vector<long>.Count (the oracle's convention).  C ABI: MCF_VECTOR_NONE = -1, 0 = default (4)."""
    if v not in (0, 2, 4, 8):
        raise ValueError(f"vector_width {v} is not one of 0, 2, 4, 8")
    return L.VECTOR_NONE if v == 0 else v


def _i32(a): return np.ascontiguousarray(a, np.int32)
def _i64(a): return np.ascontiguousarray(a, np.int64)
def _i8(a): return np.ascontiguousarray(a, np.int8)


@dataclass
class Problem:
    """Flat min-cost-flow instance, 0-based; upper == INF_CAP means unbounded."""
    node_count: int
    arc_count: int
    source: np.ndarray
    target: np.ndarray
    lower: np.ndarray
    upper: np.ndarray
    cost: np.ndarray
    supply: np.ndarray

    @staticmethod
    def _take(ps: L.ProblemStruct) -> "Problem":
        n, m = ps.node_count, ps.arc_count
        def arr(ptr, k, dt):
            return np.ctypeslib.as_array(ptr, shape=(max(k, 1),))[:k].astype(dt, copy=True)
        p = Problem(n, m, arr(ps.source, m, np.int32), arr(ps.target, m, np.int32), arr(ps.lower, m, np.int64),
                    arr(ps.upper, m, np.int64), arr(ps.cost, m, np.int64), arr(ps.supply, n, np.int64))
        L.lib().mcf_problem_free(C.byref(ps))
        return p


def netgen_like(seed: int, nodes: int, arcs: int, n_src: int, n_snk: int, min_cost=1, max_cost=10000,
                min_cap=1, max_cap=1000) -> Problem:
    """Build-owned NETGEN-like generator; mirrors the headers of the bundled netgen_8_*.min files."""
    ps = L.ProblemStruct()
    L.check(L.lib().mcf_gen_netgen_like(C.byref(ps), seed, nodes, arcs, n_src, n_snk, min_cost, max_cost, min_cap, max_cap))
    return Problem._take(ps)


def assignment(seed: int, n: int, min_cost=1, max_cost=100) -> Problem:
    ps = L.ProblemStruct()
    L.check(L.lib().mcf_gen_assignment(C.byref(ps), seed, n, min_cost, max_cost))
    return Problem._take(ps)


def read_dimacs(path: str) -> Problem:
    ps = L.ProblemStruct()
    L.check(L.lib().mcf_dimacs_read(C.byref(ps), path.encode()))
    return Problem._take(ps)


def _borrow(p: Problem):
    """A mcf_problem view of p's arrays (they must stay alive while the struct is in use)."""
    keep = (_i32(p.source), _i32(p.target), _i64(p.lower), _i64(p.upper), _i64(p.cost), _i64(p.supply))
    src, tgt, lo, up, co, su = keep
    ps = L.ProblemStruct(p.node_count, p.arc_count, src.ctypes.data_as(C.POINTER(C.c_int32)),
                         tgt.ctypes.data_as(C.POINTER(C.c_int32)), lo.ctypes.data_as(C.POINTER(C.c_int64)),
                         up.ctypes.data_as(C.POINTER(C.c_int64)), co.ctypes.data_as(C.POINTER(C.c_int64)),
                         su.ctypes.data_as(C.POINTER(C.c_int64)))
    return ps, keep


def write_dimacs(p: Problem, path: str) -> None:
    ps, _keep = _borrow(p)
    L.check(L.lib().mcf_dimacs_write(C.byref(ps), path.encode()))


def write_solution(path: str, cost: int, flow, pi=None) -> None:
    """SolutionLoader.SaveToFile (Loaders/SolutionLoader.cs:186-210)."""
    flow = _i64(flow)
    pi = None if pi is None else _i64(pi)
    L.check(L.lib().mcf_solution_write(path.encode(), int(cost), flow.shape[0], flow.ctypes.data_as(C.c_void_p),
                                       0 if pi is None else pi.shape[0], None if pi is None else pi.ctypes.data_as(C.c_void_p)))


def read_solution(path: str, p: Problem) -> dict:
    """SolutionLoader.LoadFromFile (:59-176) mapped onto p's arcs: {'cost' (None when the file has no s line), 'flow', 'pi' (None without p lines)}."""
    ps, _keep = _borrow(p)
    flow, pi = np.zeros(max(p.arc_count, 1), np.int64), np.zeros(max(p.node_count, 1), np.int64)
    cost, has_cost, has_pi = C.c_int64(), C.c_int32(), C.c_int32()
    L.check(L.lib().mcf_solution_read(path.encode(), C.byref(ps), C.byref(cost), C.byref(has_cost), flow.ctypes.data_as(C.c_void_p),
                                      pi.ctypes.data_as(C.c_void_p), C.byref(has_pi)))
    return {"cost": cost.value if has_cost.value else None, "flow": flow[: p.arc_count], "pi": pi[: p.node_count] if has_pi.value else None}


class NetworkSimplex:
    """Primal network simplex; same call sequence as the reference class."""

    def __init__(self, node_count: int, source, target):
        self._src, self._tgt = _i32(source), _i32(target)
        self.node_count, self.arc_count = int(node_count), int(self._src.shape[0])
        self._h = C.c_void_p()
        L.check(L.lib().mcf_ns_create(C.byref(self._h), self.node_count, self.arc_count, self._src, self._tgt))
        self._trace = None

    @classmethod
    def from_problem(cls, p: Problem) -> "NetworkSimplex":
        ns = cls(p.node_count, p.source, p.target)
        ns.set_problem(p.lower, p.upper, p.cost, p.supply)
        return ns

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            L.lib().mcf_ns_destroy(h)
            self._h = None

    # --- NetworkSimplex.cs:153-210
    def set_arc_bounds(self, arc: int, lower: int, upper: int):
        L.check(L.lib().mcf_ns_set_arc_bounds(self._h, arc, lower, upper)); return self

    def set_arc_cost(self, arc: int, cost: int):
        L.check(L.lib().mcf_ns_set_arc_cost(self._h, arc, cost)); return self

    def set_node_supply(self, node: int, supply: int):
        L.check(L.lib().mcf_ns_set_node_supply(self._h, node, supply)); return self

    def set_problem(self, lower=None, upper=None, cost=None, supply=None):
        keep = [None if a is None else _i64(a) for a in (lower, upper, cost, supply)]
        ptr = [None if a is None else a.ctypes.data for a in keep]
        L.check(L.lib().mcf_ns_set_problem(self._h, *ptr)); return self

    def set_supply_type(self, t: int):
        L.check(L.lib().mcf_ns_set_supply_type(self._h, t)); return self

    def set_pivot_rule(self, rule: int):
        L.check(L.lib().mcf_ns_set_pivot_rule(self._h, rule)); return self

    def enable_optimized_pivot(self, enable: bool = True):     # NetworkSimplex.cs:532-535
        L.check(L.lib().mcf_ns_enable_optimized_pivot(self._h, int(enable))); return self

    def set_vector_width(self, vector_width: int):
        """Vector<long>.Count of the machine whose EnableOptimizedPivot(true) Block Search is reproduced: 4 (x64, the default), 2, 8, or
        0 = Vector.IsHardwareAccelerated is false (BlockSearchPivotOptimized.cs:74, :115; the reference reads the property, it has no setter)."""
        L.check(L.lib().mcf_ns_set_vector_width(self._h, _vector_width_abi(vector_width))); return self

    # --- NetworkSimplex.cs:549-570
    def set_optimization_config(self, config):                 # SetOptimizationConfig: switches auto-configuration off
        L.check(L.lib().mcf_ns_set_optimization_config(self._h, C.byref(config))); return self

    def enable_optimizations(self, flags: int):
        L.check(L.lib().mcf_ns_enable_optimizations(self._h, flags)); return self

    def set_auto_configuration(self, enable: bool = True):     # the reference's default is on
        L.check(L.lib().mcf_ns_set_auto_configuration(self._h, int(enable))); return self

    # --- device options (no counterpart in the reference)
    def set_device(self, device=0, int_width=0, block_size=0, engine_flags=0):
        L.check(L.lib().mcf_ns_set_device(self._h, device, int_width, block_size, engine_flags)); return self

    def set_device_share(self, resident_workgroups: int):
        """mcf_ns_set_device_share: this solver's resident grid gets that many workgroups (256 / K for K solvers in flight on one device)."""
        L.check(L.lib().mcf_ns_set_device_share(self._h, resident_workgroups)); return self

    def set_sharding(self, nccl_id: np.ndarray, rank: int, world: int):
        L.check(L.lib().mcf_ns_set_sharding(self._h, np.ascontiguousarray(nccl_id, np.uint8), rank, world)); return self

    def set_sharding_host(self, exchange_name: str, rank: int, world: int):
        """Arc shards over `world` ranks of one node, candidates exchanged through shared memory (mcf_exchange_*)."""
        L.check(L.lib().mcf_ns_set_sharding_host(self._h, exchange_name.encode(), rank, world)); return self

    def set_shard_group(self, devices):
        """Arc shards inside this process: one engine per entry of `devices` (entries may repeat), reduced by the host thread."""
        d = _i32(devices)
        L.check(L.lib().mcf_ns_set_shard_group(self._h, d.shape[0], d)); return self

    def set_pivot_limit(self, max_pivots: int):
        L.check(L.lib().mcf_ns_set_pivot_limit(self._h, max_pivots)); return self

    def record_trace(self, capacity: int):
        self._trace = np.zeros(max(capacity, 1), np.int32)
        L.check(L.lib().mcf_ns_set_trace(self._h, self._trace.ctypes.data, capacity)); return self

    def trace(self) -> np.ndarray:
        n = C.c_int64()
        L.check(L.lib().mcf_ns_get_trace_length(self._h, C.byref(n)))
        return self._trace[: n.value].copy()

    def prepare(self):
        """Standard form, start basis, engine creation and upload of the SoA arrays into HBM (solve() does it if needed)."""
        L.check(L.lib().mcf_ns_prepare(self._h)); return self

    # --- NetworkSimplex.cs:215-470
    def solve(self) -> int:
        st = C.c_int32()
        L.check(L.lib().mcf_ns_solve(self._h, C.byref(st)))
        return st.value

    @property
    def status(self) -> int:
        st = C.c_int32()
        L.check(L.lib().mcf_ns_status(self._h, C.byref(st)))
        return st.value

    def get_flow(self, arc: int) -> int:
        v = C.c_int64(); L.check(L.lib().mcf_ns_get_flow(self._h, arc, C.byref(v))); return v.value

    def get_potential(self, node: int) -> int:
        v = C.c_int64(); L.check(L.lib().mcf_ns_get_potential(self._h, node, C.byref(v))); return v.value

    def get_total_cost(self) -> int:
        v = C.c_int64(); L.check(L.lib().mcf_ns_get_total_cost(self._h, C.byref(v))); return v.value

    def get_arc_upper_bound(self, arc: int) -> int:
        v = C.c_int64(); L.check(L.lib().mcf_ns_get_arc_upper_bound(self._h, arc, C.byref(v))); return v.value

    def flows(self) -> np.ndarray:
        out = np.empty(max(self.arc_count, 1), np.int64); L.check(L.lib().mcf_ns_get_flows(self._h, out)); return out[: self.arc_count]

    def potentials(self) -> np.ndarray:
        out = np.empty(max(self.node_count, 1), np.int64); L.check(L.lib().mcf_ns_get_potentials(self._h, out)); return out[: self.node_count]

    def get_metrics(self) -> dict:
        m = L.NsMetrics(); L.check(L.lib().mcf_ns_get_metrics(self._h, C.byref(m))); return m.as_dict()

    def check_reduced_costs(self) -> int:
        m = C.c_int64(); L.check(L.lib().mcf_ns_check_reduced_costs(self._h, C.byref(m))); return m.value

    def validate(self) -> dict:                                  # SolutionValidator(graph, solver).Validate()
        v = L.Validation(); L.check(L.lib().mcf_ns_validate(self._h, C.byref(v))); return v.as_dict()

    # --- the sequential half on its own (what a C# host keeps); never searches for an entering arc
    def begin(self) -> int:
        st = C.c_int32(); L.check(L.lib().mcf_ns_begin(self._h, C.byref(st))); return st.value

    def apply_pivot(self, entering_arc: int) -> bool:
        unb = C.c_int32(); L.check(L.lib().mcf_ns_apply_pivot(self._h, entering_arc, C.byref(unb))); return bool(unb.value)

    def finish(self) -> int:
        st = C.c_int32(); L.check(L.lib().mcf_ns_finish(self._h, C.byref(st))); return st.value

    def replay(self, arcs, smaller_side=True, renumber_every=0.0):
        """mcf_ns_replay: the given entering arcs applied back to back (no engine): the sequential half alone, timed in get_metrics()."""
        arcs = _i32(arcs)
        L.check(L.lib().mcf_ns_replay(self._h, arcs, arcs.shape[0], int(smaller_side), float(renumber_every)))
        m = L.NsMetrics(); L.check(L.lib().mcf_ns_get_metrics(self._h, C.byref(m)))
        self.replay_relabellings = int(m.reserved)          # how often the nodes were relabelled (mcf_ns_metrics.reserved after a replay)
        return self

    def internal(self) -> dict:
        ms, cap = C.c_int32(), C.c_int32()
        ps, pt = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        pc, ppi, pst = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int8)()
        L.check(L.lib().mcf_ns_internal(self._h, C.byref(ms), C.byref(cap), C.byref(ps), C.byref(pt), C.byref(pc), C.byref(pst), C.byref(ppi)))
        a, n1 = cap.value, self.node_count + 1
        cp = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
        return dict(search_arc_num=ms.value, arc_capacity=a, source=cp(ps, a), target=cp(pt, a), cost=cp(pc, a),
                    state=cp(pst, a), pi=cp(ppi, n1))

    def tree(self) -> dict:
        """Copies of Parent, Pred, SuccNum, PredDir (node_count + 1 entries) and the internal flow / upper arrays (arc_capacity entries)."""
        par, pred, succ = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        pdir, flow, upper = C.POINTER(C.c_int8)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
        L.check(L.lib().mcf_ns_tree(self._h, C.byref(par), C.byref(pred), C.byref(succ), C.byref(pdir), C.byref(flow), C.byref(upper)))
        n1, a = self.node_count + 1, self.arc_count + 2 * self.node_count
        cp = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
        return dict(parent=cp(par, n1), pred_arc=cp(pred, n1), succ_num=cp(succ, n1), pred_dir=cp(pdir, n1), flow=cp(flow, a), upper=cp(upper, a))

    def last_pivot(self) -> dict:
        ns, nn, sigma = C.c_int32(), C.c_int32(), C.c_int64()
        arcs, states = (C.c_int32 * 2)(), (C.c_int8 * 2)()
        nodes = C.POINTER(C.c_int32)()
        L.check(L.lib().mcf_ns_last_pivot(self._h, C.byref(ns), arcs, states, C.byref(nn), C.byref(nodes), C.byref(sigma)))
        nd = np.ctypeslib.as_array(nodes, shape=(nn.value,)).copy() if nn.value else np.zeros(0, np.int32)
        return dict(state_arcs=np.array(arcs[: ns.value], np.int32), state_values=np.array(states[: ns.value], np.int8),
                    nodes=nd, sigma=sigma.value)


def block_config(**kw) -> "L.BlockConfig":
    """new OptimizationConfig { ... } (OptimizationTypes.cs:24-38): the defaults, with the given fields replaced."""
    c = L.BlockConfig()
    L.lib().mcf_block_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def auto_block_config(node_count: int, source, target) -> "L.BlockConfig":
    """What the reference's auto-configuration picks for this graph (ProblemAnalyzer + OptimizationSelector)."""
    c = L.BlockConfig()
    src, tgt = _i32(source), _i32(target)
    L.check(L.lib().mcf_block_config_auto(C.byref(c), node_count, src.shape[0], src, tgt))
    return c


class SolutionValidator:
    """The reference's SolutionValidator checks as device reductions (mcf_validator_* of include/mcf_hip.h)."""

    def __init__(self, node_count: int, arc_count: int, device: int = 0):
        self.node_count, self.arc_count = int(node_count), int(arc_count)
        self._h = C.c_void_p()
        L.check(L.lib().mcf_validator_create(C.byref(self._h), device, self.node_count, self.arc_count))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            L.lib().mcf_validator_destroy(h)
            self._h = None

    def _ptrs(self, arrays, kinds, lengths):
        keep, out = [], []
        for a, dt, k in zip(arrays, kinds, lengths):
            if a is None:
                out.append(None)
                continue
            a = np.ascontiguousarray(a, dt)
            if a.shape != (k,):
                raise ValueError(f"expected an array of {k} entries, got {a.shape}")
            keep.append(a)
            out.append(a.ctypes.data_as(C.c_void_p))
        return keep, out

    def upload_network(self, source, target, lower, upper, cost, supply):
        m, n = self.arc_count, self.node_count
        keep, p = self._ptrs((source, target, lower, upper, cost, supply), (np.int32, np.int32, np.int64, np.int64, np.int64, np.int64), (m, m, m, m, m, n))
        L.check(L.lib().mcf_validator_upload(self._h, *p, None, None))
        return self

    def upload_solution(self, flow, pi):
        keep, p = self._ptrs((flow, pi), (np.int64, np.int64), (self.arc_count, self.node_count))
        L.check(L.lib().mcf_validator_upload(self._h, None, None, None, None, None, None, *p))
        return self

    def run(self, supply_type: int, reported_cost: int) -> dict:
        v = L.Validation(); L.check(L.lib().mcf_validator_run(self._h, int(supply_type), int(reported_cost), C.byref(v))); return v.as_dict()


class PivotEngine:
    """Device-resident SoA + pivot rules (mcf_engine_* of include/mcf_hip.h)."""

    def __init__(self, node_count: int, arc_capacity: int, search_arc_num: int, rule=PivotRule.BlockSearch,
                 optimized=True, int_width=64, block_size=0, device=0, shard=(0, 0), scan_workgroups=0, flags=0, resident_workgroups=0, vector_width=4):
        d = L.EngineDesc(node_count, arc_capacity, search_arc_num, int_width, rule, L.SEM_OPTIMIZED if optimized else L.SEM_PLAIN,
                         block_size, device, shard[0], shard[1], scan_workgroups, flags, resident_workgroups, _vector_width_abi(vector_width))
        self._h = C.c_void_p()
        L.check(L.lib().mcf_engine_create(C.byref(self._h), C.byref(d)))
        self.node_count, self.arc_capacity, self.search_arc_num = node_count, arc_capacity, search_arc_num

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            L.lib().mcf_engine_destroy(h)
            self._h = None

    def upload(self, source, target, cost, state, pi):
        L.check(L.lib().mcf_engine_upload(self._h, _i32(source), _i32(target), _i64(cost), _i8(state), _i64(pi)))

    def patch_state(self, arcs, states):
        arcs, states = _i32(arcs), _i8(states)
        L.check(L.lib().mcf_engine_patch_state(self._h, len(arcs), arcs, states))

    def update_potential(self, nodes, sigma: int):
        nodes = _i32(nodes)
        L.check(L.lib().mcf_engine_update_potential(self._h, len(nodes), nodes, sigma))

    def set_potential(self, nodes, values):
        nodes, values = _i32(nodes), _i64(values)
        L.check(L.lib().mcf_engine_set_potential(self._h, len(nodes), nodes, values))

    def append_potential(self, nodes, values):
        nodes, values = _i32(nodes), _i64(values)
        L.check(L.lib().mcf_engine_append_potential(self._h, nodes.shape[0], nodes, values))

    def shift_potential(self, nodes, values, sigma: int):
        """mcf_engine_shift_potential; `values` may be None when the potentials are bound (bind_potentials): the bound array holds them."""
        nodes = _i32(nodes)
        L.check(L.lib().mcf_engine_shift_potential(self._h, nodes.shape[0], nodes, None if values is None else _i64(values), sigma))

    def shift_potential_runs(self, first, length, sigma: int):
        """mcf_engine_shift_potential_runs: nodes first[r] .. first[r] + length[r] - 1 for every r moved by sigma (bound potentials only)."""
        first, length = _i32(first), _i32(length)
        L.check(L.lib().mcf_engine_shift_potential_runs(self._h, first.shape[0], first, length, sigma))

    def bind_potentials(self, pi):
        """mcf_engine_bind_potentials: `pi` (int64[node_count], C-contiguous) is read in place from now on; the caller keeps it alive and current."""
        if pi is None:
            self._bound_pi = None
            L.check(L.lib().mcf_engine_bind_potentials(self._h, None))
            return
        assert pi.dtype == np.int64 and pi.flags["C_CONTIGUOUS"]
        self._bound_pi = pi
        L.check(L.lib().mcf_engine_bind_potentials(self._h, pi.ctypes.data))

    def reload_threshold(self) -> int:
        v = C.c_int32(0)
        L.check(L.lib().mcf_engine_reload_threshold(self._h, C.byref(v)))
        return v.value

    def reload_potentials(self, changed_nodes: int):
        L.check(L.lib().mcf_engine_reload_potentials(self._h, changed_nodes))

    def renumber_nodes(self, new_of):
        """mcf_engine_renumber_nodes: new_of[old id] = new id (a permutation); a bound potential array must already be in the new order."""
        new_of = _i32(new_of)
        assert new_of.shape == (self.node_count,)
        L.check(L.lib().mcf_engine_renumber_nodes(self._h, new_of))

    def check_reduced_costs(self):
        """(mismatching arcs, lowest such arc or -1): the per-arc reduced costs of the RC layout against cost + pi[source] - pi[target] on the device."""
        m, f = C.c_int64(), C.c_int32()
        L.check(L.lib().mcf_engine_check_reduced_costs(self._h, C.byref(m), C.byref(f)))
        return m.value, f.value

    def patch_arcs(self, arcs, source, target, cost):
        arcs = _i32(arcs)
        L.check(L.lib().mcf_engine_patch_arcs(self._h, len(arcs), arcs, _i32(source), _i32(target), _i64(cost)))

    def find_entering(self):
        f, a, c = C.c_int32(), C.c_int32(), C.c_int64()
        L.check(L.lib().mcf_engine_find_entering(self._h, C.byref(f), C.byref(a), C.byref(c)))
        return bool(f.value), a.value, c.value

    def search_begin(self):
        L.check(L.lib().mcf_engine_search_begin(self._h))

    def search_end(self):
        f, a, c = C.c_int32(), C.c_int32(), C.c_int64()
        L.check(L.lib().mcf_engine_search_end(self._h, C.byref(f), C.byref(a), C.byref(c)))
        return bool(f.value), a.value, c.value

    def search_end_local(self) -> L.Candidate:
        c = L.Candidate()
        L.check(L.lib().mcf_engine_search_end_local(self._h, C.byref(c)))
        return c

    def find_entering_local(self) -> L.Candidate:
        c = L.Candidate()
        L.check(L.lib().mcf_engine_find_entering_local(self._h, C.byref(c)))
        return c

    def resolve(self, cands):
        arr = (L.Candidate * len(cands))(*cands)
        f, a, c = C.c_int32(), C.c_int32(), C.c_int64()
        L.check(L.lib().mcf_engine_resolve(self._h, len(cands), arr, C.byref(f), C.byref(a), C.byref(c)))
        return bool(f.value), a.value, c.value

    @property
    def next_arc(self) -> int:
        v = C.c_int32(); L.check(L.lib().mcf_engine_get_next_arc(self._h, C.byref(v))); return v.value

    @next_arc.setter
    def next_arc(self, v: int):
        L.check(L.lib().mcf_engine_set_next_arc(self._h, v))

    @property
    def block_size(self) -> int:
        v = C.c_int32(); L.check(L.lib().mcf_engine_get_block_size(self._h, C.byref(v))); return v.value

    def set_block_config(self, config, graph_node_count: int):
        L.check(L.lib().mcf_engine_set_block_config(self._h, C.byref(config), graph_node_count))

    def download_pi(self) -> np.ndarray:
        out = np.empty(self.node_count, np.int64); L.check(L.lib().mcf_engine_download_pi(self._h, out)); return out

    def download_state(self) -> np.ndarray:
        out = np.zeros(self.arc_capacity, np.int8); L.check(L.lib().mcf_engine_download_state(self._h, out)); return out

    def stats(self) -> dict:
        s = L.EngineStats(); L.check(L.lib().mcf_engine_get_stats(self._h, C.byref(s))); return s.as_dict()

    def park(self):
        L.check(L.lib().mcf_engine_park(self._h))

    def reset_stats(self):
        L.check(L.lib().mcf_engine_reset_stats(self._h))

    def bench_scan(self, reps=20, cold=False, flush_bytes=512 << 20):
        avg, mn = C.c_double(), C.c_double()
        L.check(L.lib().mcf_engine_bench_scan(self._h, reps, int(cold), flush_bytes, C.byref(avg), C.byref(mn)))
        return avg.value, mn.value


class HostExchange:
    """mcf_exchange_*: all-gather of the 16-byte candidates between the ranks of one node through POSIX shared memory."""

    def __init__(self, name: str, rank: int, world: int):
        self.world = world
        self._h = C.c_void_p()
        L.check(L.lib().mcf_exchange_open(C.byref(self._h), name.encode(), rank, world))

    def all_gather(self, mine: "L.Candidate"):
        out = (L.Candidate * self.world)()
        L.check(L.lib().mcf_exchange_all_gather(self._h, C.byref(mine), out))
        return list(out)

    def close(self):
        if self._h:
            L.lib().mcf_exchange_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


def _bench_update(self, count, reps=20):
    """mcf_engine_bench_update: (avg ns, min ns, algorithmic bytes) of the potential-update kernel over `count` distinct nodes."""
    avg, mn, nb = C.c_double(), C.c_double(), C.c_int64()
    L.check(L.lib().mcf_engine_bench_update(self._h, count, reps, C.byref(avg), C.byref(mn), C.byref(nb)))
    return avg.value, mn.value, nb.value


PivotEngine.bench_update = _bench_update


def _bench_search(self, reps=1000):
    avg, mn = C.c_double(), C.c_double()
    L.check(L.lib().mcf_engine_bench_search(self._h, reps, C.byref(avg), C.byref(mn)))
    return avg.value, mn.value


PivotEngine.bench_search = _bench_search


def shard_range(search_arc_num: int, rank: int, world: int):
    b, e = C.c_int32(), C.c_int32()
    L.check(L.lib().mcf_shard_range(search_arc_num, rank, world, C.byref(b), C.byref(e)))
    return b.value, e.value


def resolve_candidates(rule: int, optimized: bool, search_arc_num: int, block_size: int, next_arc: int, cands, vector_width=4):
    """Engine-free MINLOC over per-shard candidates; returns (found, arc, reduced_cost, new_next_arc)."""
    arr = (L.Candidate * len(cands))(*cands)
    na, f, a, c = C.c_int32(next_arc), C.c_int32(), C.c_int32(), C.c_int64()
    L.check(L.lib().mcf_resolve_candidates(rule, L.SEM_OPTIMIZED if optimized else L.SEM_PLAIN, _vector_width_abi(vector_width), search_arc_num, block_size,
                                           C.byref(na), len(cands), arr, C.byref(f), C.byref(a), C.byref(c)))
    return bool(f.value), a.value, c.value, na.value
