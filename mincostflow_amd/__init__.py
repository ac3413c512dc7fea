"""mincostflow_amd -- MI355X-native entering-arc / potential-update engine behind the reference's
NetworkSimplex seam.  See DESIGN.md; the C ABI is include/mcf_hip.h."""
from ._lib import (ENGINE_NO_CANDIDATES, ENGINE_CANDIDATES, ENGINE_DISPATCH, ENGINE_SHARE_DEVICE, ENGINE_NO_INLINE_UPDATE, ENGINE_RESIDENT, ENGINE_SAMPLE_KERNEL_TIME, ENGINE_TIME_EVERY_KERNEL, INF_CAP, LIB_PATH,
                   McfError, comm_unique_id, device_count)
from ._lib import OPT_ADAPTIVE_BLOCK_SIZE, OPT_NONE, OPT_REDUCED_COST_CACHING, OPT_SMALL_BLOCKS_FOR_DENSE, BlockConfig
from .network_simplex import (auto_block_config, block_config, HostExchange, NetworkSimplex, PivotEngine, PivotRule, Problem, SolutionValidator, SolverStatus, SupplyType, assignment,
                              netgen_like, read_dimacs, read_solution, resolve_candidates, shard_range, write_dimacs, write_solution)

__all__ = ["HostExchange", "BlockConfig", "block_config", "auto_block_config", "OPT_NONE", "OPT_ADAPTIVE_BLOCK_SIZE", "OPT_SMALL_BLOCKS_FOR_DENSE", "OPT_REDUCED_COST_CACHING", "NetworkSimplex", "PivotEngine", "SolutionValidator", "PivotRule", "Problem", "SolverStatus", "SupplyType", "assignment",
           "netgen_like", "read_dimacs", "write_dimacs", "read_solution", "write_solution", "shard_range", "resolve_candidates", "McfError", "device_count", "comm_unique_id", "INF_CAP", "LIB_PATH",
           "ENGINE_SAMPLE_KERNEL_TIME", "ENGINE_TIME_EVERY_KERNEL", "ENGINE_NO_INLINE_UPDATE", "ENGINE_RESIDENT", "ENGINE_DISPATCH", "ENGINE_CANDIDATES", "ENGINE_NO_CANDIDATES", "ENGINE_SHARE_DEVICE"]
