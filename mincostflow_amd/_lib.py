"""ctypes binding of libmcf_hip.so (the C ABI declared in include/mcf_hip.h).

There is no Python or CPU fallback: if the shared library is missing this module raises, and every
device operation fails with McfError when no MI355X is usable.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcf_hip.so")

# enums of include/mcf_hip.h
RULE_FIRST_ELIGIBLE, RULE_BEST_ELIGIBLE, RULE_BLOCK_SEARCH = 0, 1, 2
SEM_PLAIN, SEM_OPTIMIZED = 1, 2
VECTOR_DEFAULT, VECTOR_NONE = 0, -1          # mcf_engine_desc.vector_width: 0 = 4 (x64), -1 = Vector.IsHardwareAccelerated is false
SUPPLY_GEQ, SUPPLY_LEQ = 0, 1
NOT_SOLVED, OPTIMAL, INFEASIBLE, UNBOUNDED, UNBALANCED = 0, 1, 2, 3, 4
STATE_UPPER, STATE_TREE, STATE_LOWER = -1, 0, 1
INF_CAP = np.iinfo(np.int64).max
ENGINE_SAMPLE_KERNEL_TIME, ENGINE_TIME_EVERY_KERNEL, ENGINE_NO_INLINE_UPDATE, ENGINE_RESIDENT, ENGINE_DISPATCH, ENGINE_CANDIDATES = 1, 2, 4, 8, 16, 32
ENGINE_SHARE_DEVICE = 64
ENGINE_NO_CANDIDATES = 128
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OVERFLOW, ERR_TIMEOUT, ERR_STATE, ERR_IO, ERR_COMM = -1, -2, -3, -4, -5, -6, -7, -8


class McfError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mcf_hip error {code}: {message}")
        self.code = code


class EngineDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("node_count", "arc_capacity", "search_arc_num", "int_width", "rule", "semantics",
                                         "block_size", "device", "shard_begin", "shard_end", "scan_workgroups", "flags", "resident_workgroups", "vector_width")]


class Candidate(C.Structure):
    _fields_ = [("reduced_cost", C.c_int64), ("pos", C.c_uint32), ("arc", C.c_int32),
                ("range_cost", C.c_int64), ("range_pos", C.c_uint32), ("range_arc", C.c_int32)]

    def __init__(self, reduced_cost=0, pos=0xFFFFFFFF, arc=-1, range_cost=0, range_pos=0xFFFFFFFF, range_arc=-1):
        super().__init__(reduced_cost, pos, arc, range_cost, range_pos, range_arc)


class EngineStats(C.Structure):
    _fields_ = [("searches", C.c_int64), ("scan_launches", C.c_int64), ("update_launches", C.c_int64),
                ("inline_updates", C.c_int64), ("potential_nodes", C.c_int64), ("arcs_scanned", C.c_int64),
                ("timed_scans", C.c_int64), ("timed_scan_ns", C.c_double), ("host_wait_ns", C.c_double),
                ("host_launch_ns", C.c_double), ("scan_workgroups", C.c_int32), ("scan_threads", C.c_int32),
                ("bytes_per_scan", C.c_int64), ("resident", C.c_int64), ("resident_launches", C.c_int64),
                ("resident_requests", C.c_int64), ("resident_scan_ns", C.c_double), ("resident_kernel_ns", C.c_double), ("candidates", C.c_int64),
                ("host_decided", C.c_int64), ("arcs_checked", C.c_int64), ("initial_block_size", C.c_int32), ("current_block_size", C.c_int32),
                ("comm_ranks", C.c_int32), ("reserved", C.c_int32), ("async_refreshes", C.c_int64), ("scan_bytes_read", C.c_int64), ("rc_layout", C.c_int64), ("rc_recomputes", C.c_int64), ("renumberings", C.c_int64), ("heap_compactions", C.c_int64), ("rc_reloads_in_grid", C.c_int64), ("shift_grid", C.c_int64), ("shift_lists", C.c_int64), ("mirror_uploads", C.c_int64),
                ("phase_shift_ns", C.c_double), ("phase_values_ns", C.c_double), ("phase_scan_ns", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


OPT_NONE, OPT_ADAPTIVE_BLOCK_SIZE, OPT_SMALL_BLOCKS_FOR_DENSE, OPT_REDUCED_COST_CACHING = 0, 1, 2, 4


class BlockConfig(C.Structure):
    """mcf_block_config: the OptimizationConfig fields the plain BlockSearchPivot reads (OptimizationTypes.cs:24-38)."""
    _fields_ = [("flags", C.c_int32), ("min_block_size", C.c_int32), ("max_block_size", C.c_int32), ("consecutive_hits_before_adapt", C.c_int32),
                ("min_block_size_ratio", C.c_double), ("block_size_growth_factor", C.c_double), ("block_size_shrink_factor", C.c_double),
                ("low_hit_rate_threshold", C.c_double), ("high_hit_rate_threshold", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class NsMetrics(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("total_solve_us", C.c_double), ("pivot_search_us", C.c_double),
                ("tree_update_us", C.c_double), ("potential_update_us", C.c_double), ("setup_us", C.c_double),
                ("loop_us", C.c_double), ("search_arc_num", C.c_int32),
                ("block_size", C.c_int32), ("int_width", C.c_int32), ("reserved", C.c_int32),
                ("degenerate_pivots", C.c_int64), ("potential_nodes", C.c_int64), ("engine", EngineStats),
                ("initial_block_size", C.c_int32), ("final_block_size", C.c_int32), ("total_arcs_checked", C.c_int64),
                ("average_arcs_checked_per_pivot", C.c_double), ("baseline_iterations", C.c_int32), ("config_flags", C.c_int32),
                ("iteration_ratio", C.c_double), ("reference_selects_cached_pivot", C.c_int32), ("reserved2", C.c_int32)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("engine", "reserved", "reserved2")}
        d["engine"] = self.engine.as_dict()
        return d


VALIDATION_KINDS = ("conservation", "lower", "upper", "slack_pos", "slack_neg", "node_dual", "node_slack", "objective", "dual_cost", "status")


class Validation(C.Structure):
    _fields_ = [("valid", C.c_int32), ("supply_type", C.c_int32), ("objective", C.c_int64), ("dual_cost", C.c_int64),
                ("errors", C.c_int64 * len(VALIDATION_KINDS)), ("first", C.c_int64 * len(VALIDATION_KINDS)),
                ("kernel_us", C.c_double), ("algorithmic_bytes", C.c_int64)]

    def as_dict(self):
        return {"valid": int(self.valid), "supply_type": int(self.supply_type), "objective": int(self.objective), "dual_cost": int(self.dual_cost),
                "errors": dict(zip(VALIDATION_KINDS, map(int, self.errors))), "first": dict(zip(VALIDATION_KINDS, map(int, self.first))),
                "kernel_us": float(self.kernel_us), "algorithmic_bytes": int(self.algorithmic_bytes)}


class ProblemStruct(C.Structure):
    _fields_ = [("node_count", C.c_int32), ("arc_count", C.c_int32), ("source", C.POINTER(C.c_int32)),
                ("target", C.POINTER(C.c_int32)), ("lower", C.POINTER(C.c_int64)), ("upper", C.POINTER(C.c_int64)),
                ("cost", C.POINTER(C.c_int64)), ("supply", C.POINTER(C.c_int64))]


_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


class _i64p_or_null(_i64p):
    """int64 array or None (NULL) -- for the arguments the header marks as optional."""
    @classmethod
    def from_param(cls, obj):
        return None if obj is None else _i64p.from_param(obj)


_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_P = C.POINTER

# every symbol include/mcf_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mcf_last_error": (C.c_char_p, []),
    "mcf_version": (C.c_char_p, []),
    "mcf_device_count": (C.c_int, []),
    "mcf_device_compute_units": (C.c_int, [C.c_int32]),
    "mcf_engine_create": (C.c_int, [_P(C.c_void_p), _P(EngineDesc)]),
    "mcf_engine_destroy": (None, [C.c_void_p]),
    "mcf_engine_upload": (C.c_int, [C.c_void_p, _i32p, _i32p, _i64p, _i8p, _i64p]),
    "mcf_engine_patch_state": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i8p]),
    "mcf_engine_update_potential": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_int64]),
    "mcf_engine_set_potential": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i64p]),
    "mcf_engine_append_potential": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i64p]),
    "mcf_engine_bind_potentials": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcf_engine_shift_potential": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i64p_or_null, C.c_int64]),
    "mcf_engine_shift_potential_runs": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, C.c_int64]),
    "mcf_engine_reload_threshold": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mcf_engine_reload_potentials": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_engine_patch_arcs": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, _i32p, _i64p]),
    "mcf_engine_can_renumber": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_engine_check_reduced_costs": (C.c_int, [C.c_void_p, _P(C.c_int64), _P(C.c_int32)]),
    "mcf_ns_check_reduced_costs": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "mcf_engine_renumber_nodes": (C.c_int, [C.c_void_p, _i32p]),
    "mcf_engine_find_entering": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]),
    "mcf_engine_search_begin": (C.c_int, [C.c_void_p]),
    "mcf_engine_search_end": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]),
    "mcf_engine_find_entering_local": (C.c_int, [C.c_void_p, _P(Candidate)]),
    "mcf_engine_search_end_local": (C.c_int, [C.c_void_p, _P(Candidate)]),
    "mcf_exchange_open": (C.c_int, [_P(C.c_void_p), C.c_char_p, C.c_int32, C.c_int32]),
    "mcf_exchange_close": (None, [C.c_void_p]),
    "mcf_exchange_all_gather": (C.c_int, [C.c_void_p, _P(Candidate), _P(Candidate)]),
    "mcf_engine_resolve": (C.c_int, [C.c_void_p, C.c_int32, _P(Candidate), _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]),
    "mcf_resolve_candidates": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32), C.c_int32, _P(Candidate),
                                        _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]),
    "mcf_shard_range": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "mcf_engine_park": (C.c_int, [C.c_void_p]),
    "mcf_engine_get_next_arc": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_engine_set_next_arc": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_engine_get_block_size": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_block_config_default": (None, [_P(BlockConfig)]),
    "mcf_block_config_auto": (C.c_int, [_P(BlockConfig), C.c_int32, C.c_int32, _i32p, _i32p]),
    "mcf_block_initial_size": (C.c_int, [_P(BlockConfig), C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "mcf_block_adapt": (C.c_int, [_P(BlockConfig), C.c_int32, C.c_int64, _P(C.c_int32), _P(C.c_int32)]),
    "mcf_engine_set_block_config": (C.c_int, [C.c_void_p, _P(BlockConfig), C.c_int32]),
    "mcf_engine_download_pi": (C.c_int, [C.c_void_p, _i64p]),
    "mcf_engine_download_state": (C.c_int, [C.c_void_p, _i8p]),
    "mcf_engine_get_stats": (C.c_int, [C.c_void_p, _P(EngineStats)]),
    "mcf_engine_reset_stats": (C.c_int, [C.c_void_p]),
    "mcf_engine_bench_scan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, _P(C.c_double), _P(C.c_double)]),
    "mcf_engine_bench_search": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_double), _P(C.c_double)]),
    "mcf_engine_bench_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_int64)]),
    "mcf_comm_unique_id": (C.c_int, [_u8p]),
    "mcf_engine_comm_init": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32]),
    "mcf_engine_find_entering_sharded": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]),
    "mcf_ns_create": (C.c_int, [_P(C.c_void_p), C.c_int32, C.c_int32, _i32p, _i32p]),
    "mcf_ns_destroy": (None, [C.c_void_p]),
    "mcf_ns_set_arc_bounds": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
    "mcf_ns_set_arc_cost": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "mcf_ns_set_node_supply": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "mcf_ns_set_problem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcf_ns_set_supply_type": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_set_pivot_rule": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_enable_optimized_pivot": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_set_vector_width": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_set_optimization_config": (C.c_int, [C.c_void_p, _P(BlockConfig)]),
    "mcf_ns_enable_optimizations": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_set_auto_configuration": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_set_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mcf_ns_set_device_share": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcf_ns_set_sharding": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32]),
    "mcf_ns_set_sharding_host": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]),
    "mcf_ns_set_shard_group": (C.c_int, [C.c_void_p, C.c_int32, _i32p]),
    "mcf_ns_prepare": (C.c_int, [C.c_void_p]),
    "mcf_ns_solve": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_ns_status": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_ns_get_flow": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int64)]),
    "mcf_ns_get_potential": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int64)]),
    "mcf_ns_get_total_cost": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "mcf_ns_get_flows": (C.c_int, [C.c_void_p, _i64p]),
    "mcf_ns_get_potentials": (C.c_int, [C.c_void_p, _i64p]),
    "mcf_ns_get_arc_upper_bound": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int64)]),
    "mcf_ns_get_metrics": (C.c_int, [C.c_void_p, _P(NsMetrics)]),
    "mcf_ns_set_pivot_limit": (C.c_int, [C.c_void_p, C.c_int64]),
    "mcf_ns_set_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mcf_ns_get_trace_length": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "mcf_ns_begin": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_ns_apply_pivot": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "mcf_ns_finish": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "mcf_ns_replay": (C.c_int, [C.c_void_p, _i32p, C.c_int64, C.c_int32, C.c_double]),
    "mcf_ns_internal": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(_P(C.c_int32)), _P(_P(C.c_int32)),
                                  _P(_P(C.c_int64)), _P(_P(C.c_int8)), _P(_P(C.c_int64))]),
    "mcf_ns_tree": (C.c_int, [C.c_void_p, _P(_P(C.c_int32)), _P(_P(C.c_int32)), _P(_P(C.c_int32)), _P(_P(C.c_int8)), _P(_P(C.c_int64)), _P(_P(C.c_int64))]),
    "mcf_ns_last_pivot": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_int8), _P(C.c_int32),
                                    _P(_P(C.c_int32)), _P(C.c_int64)]),
    "mcf_validator_create": (C.c_int, [_P(C.c_void_p), C.c_int32, C.c_int32, C.c_int32]),
    "mcf_validator_destroy": (None, [C.c_void_p]),
    "mcf_validator_upload": (C.c_int, [C.c_void_p] + [C.c_void_p] * 8),
    "mcf_validator_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, _P(Validation)]),
    "mcf_ns_validate": (C.c_int, [C.c_void_p, _P(Validation)]),
    "mcf_problem_free": (None, [_P(ProblemStruct)]),
    "mcf_gen_netgen_like": (C.c_int, [_P(ProblemStruct), C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "mcf_gen_assignment": (C.c_int, [_P(ProblemStruct), C.c_uint64, C.c_int32, C.c_int64, C.c_int64]),
    "mcf_dimacs_read": (C.c_int, [_P(ProblemStruct), C.c_char_p]),
    "mcf_dimacs_write": (C.c_int, [_P(ProblemStruct), C.c_char_p]),
    "mcf_solution_write": (C.c_int, [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mcf_solution_read": (C.c_int, [C.c_char_p, _P(ProblemStruct), _P(C.c_int64), _P(C.c_int32), C.c_void_p, C.c_void_p, _P(C.c_int32)]),
}

_lib = None


def lib():
    """The loaded library.  Raises if libmcf_hip.so has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C mincostflow_amd/csrc` "
                              "(there is no Python/CPU implementation to fall back to)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise McfError(rc, lib().mcf_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().mcf_device_count()


def comm_unique_id() -> np.ndarray:
    """128-byte ncclUniqueId (rank 0 creates it, the caller broadcasts it)."""
    out = np.zeros(128, np.uint8)
    check(lib().mcf_comm_unique_id(out))
    return out
