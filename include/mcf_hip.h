/*
 * mcf_hip.h -- C ABI of libmcf_hip.so, the MI355X (gfx950) network-simplex pivot engine.
 *
 * Drop-in boundary.  All file:line citations are relative to the reference
 * repository pdegenhardt/MinCostFlow; "NS.cs" = src/MinCostFlow.Core/Lemon/Algorithms/NetworkSimplex.cs,
 * "BSPO.cs" = src/MinCostFlow.Core/Lemon/Algorithms/Internal/BlockSearchPivotOptimized.cs.
 *
 * Two layers are exported:
 *
 *  (1) mcf_engine_*  -- the device side of the seam `private interface IFindEnteringArc
 *      { bool FindEnteringArc(); }` (NS.cs:1286-1289) plus UpdatePotentials (NS.cs:1185-1209).  The
 *      SoA arc arrays (Source, Target, _cost, State) and the node potentials _pi live in HBM; the
 *      host keeps the sequential pivot loop and the spanning tree.  This is what a C# host binds
 *      with [DllImport("mcf_hip")] in place of OptimizedPivotWrapper (NS.cs:1671-1724); the stub is
 *      in INTEGRATION.md.
 *
 *  (2) mcf_ns_*      -- a C++ restatement of the host side of NetworkSimplex (the public setters /
 *      Solve() / getters of NS.cs:153-527 and the tree maintenance NS.cs:925-1183) that drives
 *      the engine, so the path can be exercised end to end without a .NET toolchain.  Names and
 *      argument meaning mirror the reference class.  There is NO CPU entering-arc search in this
 *      library: mcf_ns_solve() fails with MCF_ERR_NO_DEVICE when no HIP device is usable.
 *
 *  (3) mcf_validator_* / mcf_ns_validate -- the reference's SolutionValidator (Validation/SolutionValidator.cs)
 *      as two device reductions; like the search it has no CPU path in this library.
 *
 * Conventions: every function returns 0 (MCF_OK) or a negative mcf_status; the message is available
 * from mcf_last_error() (thread-local).  Nothing throws across the ABI.  Host arrays are borrowed
 * for the duration of the call only.  One engine / one solver = one host thread at a time (the
 * reference solver is single-threaded: docs/platform-architecture.md:126-130).
 */
#ifndef MCF_HIP_H
#define MCF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCF_API __attribute__((visibility("default")))

typedef enum mcf_status {
    MCF_OK = 0,
    MCF_ERR_INVALID = -1,     /* bad argument (ArgumentException in the reference) */
    MCF_ERR_NO_DEVICE = -2,   /* no usable HIP device / HIP runtime failure at start-up */
    MCF_ERR_HIP = -3,         /* a HIP call failed later on */
    MCF_ERR_OVERFLOW = -4,    /* a value does not fit the engine's int_width (use 64) */
    MCF_ERR_TIMEOUT = -5,     /* the device did not answer */
    MCF_ERR_STATE = -6,       /* call out of order (InvalidOperationException in the reference) */
    MCF_ERR_IO = -7,          /* file / parse error */
    MCF_ERR_COMM = -8         /* RCCL failure */
} mcf_status;

/* Types/PivotRule.cs:7-41 (values kept) */
typedef enum mcf_pivot_rule { MCF_RULE_FIRST_ELIGIBLE = 0, MCF_RULE_BEST_ELIGIBLE = 1, MCF_RULE_BLOCK_SEARCH = 2 } mcf_pivot_rule;
/* Which of the reference's two implementations of a rule is reproduced:
 * PLAIN = the nested classes of NS.cs:1292-1668; OPTIMIZED = BSPO.cs (EnableOptimizedPivot(true)). */
typedef enum mcf_semantics { MCF_SEM_PLAIN = 1, MCF_SEM_OPTIMIZED = 2 } mcf_semantics;
/* OPTIMIZED Block Search depends on a property of the MACHINE the reference runs on (difference D14):
 * BSPO.cs:74 sends ranges of at least 2 * Vector<long>.Count arcs through ProcessArcRangeSIMD when Vector.IsHardwareAccelerated, and a
 * block-boundary hit in there returns into a scalar loop whose counter is 0 and never reaches 0 again (BSPO.cs:84-99, :143-148): the scan
 * runs on to the end of the range, the entering arc is the best of the whole range and _nextArc becomes the range's end.  Only a hit in
 * the tail behind the last full group of Vector<long>.Count arcs stops at the block boundary.  vector_width = Vector<long>.Count:
 * 4 on x64 (AVX2; also .NET 8 on AVX-512 hardware), 2 on Arm NEON, 8 with 512-bit Vector<T> enabled; MCF_VECTOR_NONE = not hardware
 * accelerated (the scalar loop alone: stop at the first block boundary behind an eligible arc).  0 = MCF_VECTOR_DEFAULT = 4. */
#define MCF_VECTOR_DEFAULT 0
#define MCF_VECTOR_NONE (-1)
/* Types/SupplyType.cs */
typedef enum mcf_supply_type { MCF_SUPPLY_GEQ = 0, MCF_SUPPLY_LEQ = 1 } mcf_supply_type;
/* Types/SolverStatus.cs:7-34 (values kept) */
typedef enum mcf_solver_status { MCF_NOT_SOLVED = 0, MCF_OPTIMAL = 1, MCF_INFEASIBLE = 2, MCF_UNBOUNDED = 3, MCF_UNBALANCED = 4 } mcf_solver_status;

/* SpanningTree.cs:53-71 */
#define MCF_STATE_UPPER (-1)
#define MCF_STATE_TREE 0
#define MCF_STATE_LOWER 1

/* "no upper bound" for mcf_ns_set_arc_bounds (the reference's default capacity, NS.cs:127,616) */
#define MCF_INF_CAP INT64_MAX

MCF_API const char *mcf_last_error(void);
MCF_API const char *mcf_version(void);
/* number of visible HIP devices (0 when there is none or the runtime cannot start); never fails */
MCF_API int mcf_device_count(void);
/* compute units of a device (0 when it cannot be asked): a resident grid is one workgroup per CU */
MCF_API int mcf_device_compute_units(int32_t device);

/* ------------------------------------------------------------------------------------------------
 * (1) Engine: entering-arc search + potential update on the device
 * ---------------------------------------------------------------------------------------------- */

typedef struct mcf_engine mcf_engine;

typedef struct mcf_engine_desc {
    int32_t node_count;       /* nodes INCLUDING the artificial root: NS.cs:137 (_nodeCount + 1) */
    int32_t arc_capacity;     /* arcs the host arrays hold: NS.cs:130 (_arcCount + 2 * _nodeCount) */
    int32_t search_arc_num;   /* arcs the pivot rules scan, [0, search_arc_num): NS.cs:722 */
    int32_t int_width;        /* 32 or 64: width of cost / potential ON THE DEVICE (the ABI is always int64) */
    int32_t rule;             /* mcf_pivot_rule */
    int32_t semantics;        /* mcf_semantics */
    int32_t block_size;       /* Block Search: 0 = the reference's default for the semantics
                                 (BSPO.cs:27-28 / NS.cs:1304-1336 with the default OptimizationConfig) */
    int32_t device;           /* HIP device ordinal */
    /* arc shard owned by this engine, [shard_begin, shard_end) within [0, search_arc_num); 0,0 = all.
     * A sharded engine answers for its shard only (mcf_engine_find_entering_local); its candidate cache (MCF_ENGINE_CANDIDATES) covers the
     * shard's arcs, so a holder asks its device only when ITS range cannot be decided on the host. */
    int32_t shard_begin, shard_end;
    int32_t scan_workgroups;  /* 0 = auto; otherwise the grid of the scan kernel */
    int32_t flags;            /* MCF_ENGINE_* */
    int32_t resident_workgroups;  /* 0 = auto (one workgroup per CU, at most 256); otherwise a cap on the resident grid: engines that share
                                     a device (arc shards rehearsed on one GPU) must all be co-resident to answer.  Workgroups are dealt
                                     to the 8 XCDs round-robin, so give k engines 8 * (32 / k) each, not 256 / k */
    int32_t vector_width;         /* MCF_SEM_OPTIMIZED + Block Search: Vector<long>.Count of the reference's host -- 2, 4, 8,
                                     MCF_VECTOR_NONE, or 0 = 4 (x64).  Ignored by every other rule / flavour */
} mcf_engine_desc;

#define MCF_ENGINE_SAMPLE_KERNEL_TIME 1   /* time every 16th scan dispatch with HIP events */
#define MCF_ENGINE_TIME_EVERY_KERNEL 2    /* time every scan dispatch (micro-benchmarks) */
#define MCF_ENGINE_NO_INLINE_UPDATE 4     /* always apply patches with the separate update kernel */
#define MCF_ENGINE_RESIDENT 8             /* (default behaviour, kept for explicitness) serve searches from ONE resident scan grid fed
                                             through a mailbox in BAR-mapped VRAM instead of one dispatch per search */
#define MCF_ENGINE_CANDIDATES 32          /* (default behaviour, kept for explicitness) Best Eligible, resident mode (register-resident arcs, or the grid of the
                                             per-arc reduced-cost layout that large instances get), sparse graph (2 m_s <= 24 n): every device search also returns a candidate list that is complete below a threshold;
                                             the following searches are answered on the host from that list plus a heap of the arcs the pivots
                                             touched, whenever that provably is the scan's answer, and the list is refreshed ahead of need
                                             (identical pivot sequence, a device round trip for about one search in seven).  DESIGN.md 3.4 */
#define MCF_ENGINE_NO_CANDIDATES 128      /* every search is a device search */
#define MCF_ENGINE_SHARE_DEVICE 64        /* several engines use this device at once (independent solves): keep the resident grid's register
                                             and LDS footprint small so that their grids are co-resident on every CU -- the grid then gathers
                                             the potentials for every request instead of keeping them in registers (~0.5 us per search) */
#define MCF_ENGINE_DISPATCH 16            /* one scan dispatch per search.  Also what an engine falls back to when it exchanges over RCCL
                                             (mcf_engine_comm_init), when kernel timing flags are set, when the platform has no host-writable
                                             VRAM, or while the device's resident slots (GPU_MAX_HW_QUEUES per process, 4) are all taken.
                                             The environment variable MCF_HIP_RESIDENT=0/1 overrides the choice. */

/* The part of OptimizationConfig (Algorithms/OptimizationTypes.cs:9-38) that the plain BlockSearchPivot consumes
 * (NS.cs:1304-1337 initial size, :1400-1438 adaptive size).  The OPTIMIZED flavour ignores it (BSPO.cs:27-28), and so
 * do the other rules.  Flag values are the reference's OptimizationFlags. */
#define MCF_OPT_NONE 0
#define MCF_OPT_ADAPTIVE_BLOCK_SIZE 1       /* OptimizationFlags.AdaptiveBlockSize */
#define MCF_OPT_SMALL_BLOCKS_FOR_DENSE 2    /* OptimizationFlags.SmallBlocksForDense */
#define MCF_OPT_REDUCED_COST_CACHING 4      /* OptimizationFlags.ReducedCostCaching: recorded, never acted on */
typedef struct mcf_block_config {
    int32_t flags;                          /* MCF_OPT_* */
    int32_t min_block_size;                 /* MinBlockSize                (default 25)    */
    int32_t max_block_size;                 /* MaxBlockSize                (default 100)   */
    int32_t consecutive_hits_before_adapt;  /* ConsecutiveHitsBeforeAdapt  (default 3)     */
    double min_block_size_ratio;            /* MinBlockSizeRatio           (default 0.125) */
    double block_size_growth_factor;        /* BlockSizeGrowthFactor       (default 1.2)   */
    double block_size_shrink_factor;        /* BlockSizeShrinkFactor       (default 0.8)   */
    double low_hit_rate_threshold;          /* LowHitRateThreshold         (default 0.05)  */
    double high_hit_rate_threshold;         /* HighHitRateThreshold        (default 0.3)   */
} mcf_block_config;
/* new OptimizationConfig() (OptimizationTypes.cs:24-38) */
MCF_API void mcf_block_config_default(mcf_block_config *c);
/* OptimizationSelector.SelectConfiguration(ProblemAnalyzer.Analyze(graph)) (Analysis/OptimizationSelector.cs:14-95,
 * Lemon/ProblemAnalyzer.cs:21-106) restricted to the fields above: what `new NetworkSimplex(g).Solve()` configures
 * itself with by default (NS.cs:90, :237-250).  Needs the original graph only (node count, arc end points). */
MCF_API int mcf_block_config_auto(mcf_block_config *c, int32_t node_count, int32_t arc_count, const int32_t *source, const int32_t *target);
/* The two pieces of BlockSearchPivot that act on the configuration, as pure functions (the engine calls them; a host that keeps
 * its own rule state may too).  mcf_block_initial_size: the constructor, NS.cs:1304-1336 -- *block_size = _blockSize,
 * *dynamic_min = _dynamicMinBlockSize; graph_node_count = the reference's _nodeCount (no artificial root).
 * mcf_block_adapt: NS.cs:1400-1438 after a successful search that examined arcs_checked arcs; counters[0] = _consecutiveLowHits,
 * counters[1] = _consecutiveHighHits. */
MCF_API int mcf_block_initial_size(const mcf_block_config *c, int32_t search_arc_num, int32_t graph_node_count, int32_t *block_size, int32_t *dynamic_min);
MCF_API int mcf_block_adapt(const mcf_block_config *c, int32_t dynamic_min, int64_t arcs_checked, int32_t *block_size, int32_t counters[2]);

/* replaces the constructor of OptimizedPivotWrapper (NS.cs:1677-1697) */
MCF_API int mcf_engine_create(mcf_engine **out, const mcf_engine_desc *desc);
MCF_API void mcf_engine_destroy(mcf_engine *e);

/* Copies the five arrays the reference pins per call (NS.cs:1701-1705): Source, Target (int32[arc_capacity]),
 * _cost (int64[arc_capacity]), State (int8[arc_capacity]), _pi (int64[node_count]). */
MCF_API int mcf_engine_upload(mcf_engine *e, const int32_t *source, const int32_t *target,
                              const int64_t *cost, const int8_t *state, const int64_t *pi);

/* State changes of ChangeFlow (NS.cs:1030-1039): at most two per pivot.  Queued; ordered before the next search. */
MCF_API int mcf_engine_patch_state(mcf_engine *e, int32_t count, const int32_t *arcs, const int8_t *states);

/* UpdatePotentials (NS.cs:1185-1209): pi[nodes[i]] += sigma.  The caller walks the thread list (it owns the tree);
 * nodes must be distinct.  Queued; ordered before the next search. */
MCF_API int mcf_engine_update_potential(mcf_engine *e, int32_t count, const int32_t *nodes, int64_t sigma);

/* Same as update_potential for callers that already hold the new values (the C++ host driver does: it keeps _pi itself):
 * pi[nodes[i]] = values[i]. */
MCF_API int mcf_engine_set_potential(mcf_engine *e, int32_t count, const int32_t *nodes, const int64_t *values);
/* Same, for a list that arrives in pieces (the subtree walk hands over what it has every few thousand nodes): may be called several
 * times between two searches; the pieces of one pivot must not repeat a node (UpdatePotentials visits every node once, NS.cs:1196-1208).
 * In resident mode the complete lines start travelling to the device at once and the grid applies them while the host is still
 * walking; the next search finishes the list. */
MCF_API int mcf_engine_append_potential(mcf_engine *e, int32_t count, const int32_t *nodes, const int64_t *values);
/* append_potential with the caller's promise that every values[i] is its node's previous potential + sigma -- which is what
 * UpdatePotentials produces: one sigma per pivot (NS.cs:1187-1190).  Layouts that keep reduced costs per arc (large sparse instances)
 * then shift those by sigma directly, for short lists inside the next search's dispatch.
 * `values` may be NULL when the potentials are bound (mcf_engine_bind_potentials): the bound array already holds them, and the host
 * driver's walk over a big subtree then writes node ids only. */
MCF_API int mcf_engine_shift_potential(mcf_engine *e, int32_t count, const int32_t *nodes, const int64_t *values, int64_t sigma);
/* The same list as RUNS of consecutive node ids: nodes first[r] .. first[r] + length[r] - 1 for every r, all moved by sigma; bound
 * potentials only (the runs carry no values).  For hosts whose node ids follow the thread order (mcf_engine_renumber_nodes): the subtree
 * of UpdatePotentials is then a few hundred runs instead of tens of thousands of nodes, the walk writes one pair per run, and the
 * register-resident candidate grid takes the pairs as they are.  May be called several times per pivot like append_potential (the runs of
 * one pivot must not overlap); engines without a use for runs expand them into the node list. */
MCF_API int mcf_engine_shift_potential_runs(mcf_engine *e, int32_t n_runs, const int32_t *first, const int32_t *length, int64_t sigma);

/* Optional: the caller keeps _pi anyway (the host solver does: sigma needs _pi[_vIn] and _pi[_uIn], NS.cs:1187-1190) -- bind that array
 * (int64[node_count], must outlive the binding; NULL unbinds) and the engine reads potentials from it instead of keeping a copy of its
 * own up to date.  Contract: when a search begins the array holds the values announced by set / append / shift_potential since the last
 * search, and it does not change while a search is in flight.  mcf_engine_update_potential (+= sigma) is not available while bound. */
MCF_API int mcf_engine_bind_potentials(mcf_engine *e, const int64_t *pi);
/* Bound potentials, layouts that keep reduced costs per arc: when a pivot moves a large part of the tree, naming the nodes costs more than
 * handing the whole array over.  mcf_engine_reload_threshold: *min_nodes = the list length from which a reload is the cheaper call for this
 * engine (0: the engine only takes lists -- no binding, potentials kept in registers / LDS, 32-bit potentials).  mcf_engine_reload_potentials:
 * "the bound array is current, `changed_nodes` of its entries differ from what you last heard" (the count only feeds the statistics); it
 * replaces every set / append / shift_potential call since the last search, is ordered before the next search like them, and the array must
 * stay unchanged until that search has been answered.  The engine copies the array to the device by itself (the binding registered it with HIP)
 * and computes every reduced cost of its shard again (NS.cs:1185-1209 moves the same values; what reaches the device is the result). */
MCF_API int mcf_engine_reload_threshold(mcf_engine *e, int32_t *min_nodes);
MCF_API int mcf_engine_reload_potentials(mcf_engine *e, int32_t changed_nodes);

/* Rewrites (source, target, cost) of arcs, e.g. artificial arcs re-pointed by a warm start. Synchronous. */
MCF_API int mcf_engine_patch_arcs(mcf_engine *e, int32_t count, const int32_t *arcs, const int32_t *source,
                                  const int32_t *target, const int64_t *cost);

/* Node ids are the host's business: the engine only ever sees them as end points of arcs and as indices into the potentials.  A host
 * that relabels its nodes (the C++ driver does, in thread order, so that its subtree walks run through memory front to back) tells the
 * engine with new_of[old id] = new id, a permutation of [0, node_count).  Between two searches; a bound potential array
 * (mcf_engine_bind_potentials) must already be in the new order, and replaces every potential change announced since the last search.
 * Arc ids, states and reduced costs are untouched, so no search result changes.  mcf_engine_can_renumber: 0 for the one layout that
 * orders its arcs by node id (MCF_HIP_BUCKET_NODES). */
/* Test aid, layouts that keep reduced costs per arc: how many of this engine's stored arcs carry a reduced cost that differs from
 * cost + pi[source] - pi[target] as the device holds them (0 for the other layouts); *first_arc (optional) = the lowest such arc. */
MCF_API int mcf_engine_check_reduced_costs(mcf_engine *e, int64_t *mismatches, int32_t *first_arc);
MCF_API int mcf_engine_can_renumber(mcf_engine *e, int32_t *yes);
MCF_API int mcf_engine_renumber_nodes(mcf_engine *e, const int32_t *new_of);

/* IFindEnteringArc.FindEnteringArc (NS.cs:1286-1289, :1699-1723): blocking.  *found = 0/1; *arc = entering arc;
 * *reduced_cost = state*(cost + pi[source] - pi[target]) of that arc.  Advances the rule's internal next_arc exactly
 * as the selected reference implementation does. */
MCF_API int mcf_engine_find_entering(mcf_engine *e, int32_t *found, int32_t *arc, int64_t *reduced_cost);
/* The same search in two halves: _begin posts it and returns, _end waits for the answer.  Between the two the host may do whatever
 * the search does not depend on (the reference's ChangeFlow arithmetic and UpdateTreeStructure of the pivot just made: the device
 * only needs the State[] writes and the potentials); patches queued in between belong to the NEXT search.  find_entering = both. */
MCF_API int mcf_engine_search_begin(mcf_engine *e);
MCF_API int mcf_engine_search_end(mcf_engine *e, int32_t *found, int32_t *arc, int64_t *reduced_cost);

/* Sharded search: the local candidate of this engine's shard, as an exchangeable 32-byte record. */
typedef struct mcf_candidate {
    int64_t reduced_cost;   /* 0 when none */
    uint32_t pos;           /* scan position (rule dependent ordering key); 0xFFFFFFFF when none */
    int32_t arc;            /* -1 when none */
    /* OPTIMIZED Block Search only (0 / 0xFFFFFFFF / -1 otherwise): the shard's best arc of the first of the reference's two ranges
     * ([next_arc, m_s), then [0, next_arc)) in which the shard has an eligible arc at all -- what enters when the reference's scan
     * runs to the end of the range (vector_width above) */
    int64_t range_cost;
    uint32_t range_pos;
    int32_t range_arc;
} mcf_candidate;
MCF_API int mcf_engine_find_entering_local(mcf_engine *e, mcf_candidate *out);
/* second half of a search posted with mcf_engine_search_begin on a sharded engine (the first half is the same call for every engine) */
MCF_API int mcf_engine_search_end_local(mcf_engine *e, mcf_candidate *out);
/* Picks the global winner among `count` candidates (one per shard) with the rule's exact tie-breaking and advances
 * next_arc on THIS engine (every rank calls it with the same gathered array).  *found / *arc as above. */
MCF_API int mcf_engine_resolve(mcf_engine *e, int32_t count, const mcf_candidate *all, int32_t *found,
                               int32_t *arc, int64_t *reduced_cost);
/* The same MINLOC + next_arc bookkeeping without an engine (stateless; *next_arc is read and advanced): what every rank
 * does with the all-gathered records.  block_size 0 = the reference default for the semantics. */
MCF_API int mcf_resolve_candidates(int32_t rule, int32_t semantics, int32_t vector_width, int32_t search_arc_num, int32_t block_size, int32_t *next_arc,
                                   int32_t count, const mcf_candidate *all, int32_t *found, int32_t *arc, int64_t *reduced_cost);
/* contiguous shard of [0, search_arc_num) for rank r of R, aligned to 4 arcs */
MCF_API int mcf_shard_range(int32_t search_arc_num, int32_t rank, int32_t world, int32_t *begin, int32_t *end);

/* Stops a running resident grid (it restarts with the next search).  Call before device-wide synchronisation. */
MCF_API int mcf_engine_park(mcf_engine *e);
MCF_API int mcf_engine_get_next_arc(mcf_engine *e, int32_t *next_arc);
MCF_API int mcf_engine_set_next_arc(mcf_engine *e, int32_t next_arc);
MCF_API int mcf_engine_get_block_size(mcf_engine *e, int32_t *block_size);
/* Block Search, PLAIN semantics: size the blocks like `new BlockSearchPivot(ns)` with this configuration (NS.cs:1304-1337;
 * graph_node_count = the reference's _nodeCount, i.e. WITHOUT the artificial root) and, with MCF_OPT_ADAPTIVE_BLOCK_SIZE, adapt
 * the size after every successful search like NS.cs:1400-1438.  The size travels with every search request.  Call before the
 * first search; a block_size given at creation (desc.block_size > 0) stays the initial size. */
MCF_API int mcf_engine_set_block_config(mcf_engine *e, const mcf_block_config *c, int32_t graph_node_count);

/* parity checks */
MCF_API int mcf_engine_download_pi(mcf_engine *e, int64_t *pi_out /* [node_count] */);
MCF_API int mcf_engine_download_state(mcf_engine *e, int8_t *state_out /* [arc_capacity] */);

typedef struct mcf_engine_stats {
    int64_t searches;             /* find_entering calls */
    int64_t scan_launches;        /* scan kernel dispatches */
    int64_t update_launches;      /* separate update kernel dispatches */
    int64_t inline_updates;       /* searches that carried their patches inside the scan dispatch */
    int64_t potential_nodes;      /* sum of update_potential counts */
    int64_t arcs_scanned;         /* sum over scan dispatches of arcs read */
    int64_t timed_scans;          /* scan dispatches timed with HIP events */
    double timed_scan_ns;         /* sum of their device durations, ns */
    double host_wait_ns;          /* host time spent waiting for device answers */
    double host_launch_ns;        /* host time spent inside launch calls */
    int32_t scan_workgroups, scan_threads;
    int64_t bytes_per_scan;       /* algorithmic bytes of one scan: SURVEY.md 8d */
    int64_t resident;             /* 1 when the engine runs in resident mode */
    int64_t resident_launches;    /* dispatches of the resident grid */
    int64_t resident_requests;    /* searches it served */
    double resident_scan_ns;      /* device clock: request seen -> record published, workgroup 0, summed over requests */
    double resident_kernel_ns;    /* HIP-event residency time of those dispatches */
    int64_t candidates;           /* 1 when the candidate cache is active */
    int64_t host_decided;         /* searches answered from the candidate list without a device request */
    int64_t arcs_checked;         /* arcs the REFERENCE's loop would have examined for the same searches: what it adds to
                                     SolverMetrics.TotalArcsChecked (NS.cs:286-290).  Only the plain BlockSearchPivot counts
                                     (NS.cs:1349-1350, :1371-1372); every other finder leaves it at 0, and so does this */
    int32_t initial_block_size, current_block_size;
    int32_t comm_ranks;           /* ranks of the RCCL communicator as ncclCommCount reports them (0: no communicator) */
    int32_t reserved;
    int64_t async_refreshes;      /* candidate lists requested ahead of need (the host kept answering while the device searched) */
    int64_t scan_bytes_read;      /* bytes one scan of THIS engine's layout has to read: bytes_per_scan for the gathering layouts, 9 per arc
                                     (state + the arc's reduced cost) in the RC layout, where the gathers moved into the potential update */
    int64_t rc_layout;            /* 1: reduced costs are kept per arc (large sparse instances) */
    int64_t rc_recomputes;        /* RC layout: potential lists naming more than a sixteenth of the nodes, after which every reduced cost of the
                                     shard was computed again instead of shifting the listed nodes' arcs one by one */
    int64_t renumberings;         /* mcf_engine_renumber_nodes calls */
    int64_t heap_compactions;     /* candidate cache: times the heap of touched arcs was swept of its outdated entries */
    int64_t rc_reloads_in_grid;   /* mcf_engine_reload_potentials carried out by the resident RC grid itself (the workgroups copy the bound array,
                                     meet at a grid-wide barrier and compute their arcs' reduced costs again) instead of stopping the grid */
    int64_t shift_grid;           /* 1: the candidate cache's grid is the one that is patched straight from the request (register-resident arcs
                                     and potentials, 64-bit, at most 131072 nodes): a pivot's one big subtree travels as node ids + sigma */
    int64_t shift_lists;          /* requests that carried such a list */
    int64_t mirror_uploads;       /* times the potentials / states in device memory were written again from the host's mirrors (that grid only
                                     reads them when it starts; the host writes them when it has left) */
    double phase_shift_ns, phase_values_ns, phase_scan_ns;   /* that grid's requests on workgroup 0's clock: fetching shift lines + setting bits /
                                     fetching and applying value entries and state writes / evaluating, reducing, publishing */
} mcf_engine_stats;
MCF_API int mcf_engine_get_stats(mcf_engine *e, mcf_engine_stats *out);
MCF_API int mcf_engine_reset_stats(mcf_engine *e);

/* Scan-only micro-benchmark on the engine's resident arrays: `reps` back-to-back scan dispatches, each timed with HIP
 * events on the engine's stream; cold != 0 streams `flush_bytes` through the caches before every repetition.
 * Results are not consumed (next_arc untouched).  avg/min in ns. */
MCF_API int mcf_engine_bench_scan(mcf_engine *e, int32_t reps, int32_t cold, int64_t flush_bytes,
                                  double *avg_ns, double *min_ns);

/* Potential-update micro-benchmark (20 bytes per node of a list; 12 with 32-bit potentials): the update kernel that dispatch
 * mode runs for long lists -- update_kernel, or update_rc_kernel where reduced costs are kept per arc (then the list's nodes' arc lists are
 * walked and shifted too: + 8 bytes per node for its list bounds and 20 per arc-list entry) -- over `count` distinct nodes, `reps` launches
 * each timed with HIP events.  *bytes = algorithmic bytes of one launch.  The engine's arrays are unchanged when the call returns. */
MCF_API int mcf_engine_bench_update(mcf_engine *e, int32_t count, int32_t reps, double *avg_ns, double *min_ns, int64_t *bytes);

/* Whole-search micro-benchmark: host wall time from posting / launching a search to its merged answer, `reps` times back to back without
 * patches (the candidate cache is bypassed by nothing here: an engine with the cache answers from it).  avg/min in ns. */
MCF_API int mcf_engine_bench_search(mcf_engine *e, int32_t reps, double *avg_ns, double *min_ns);

/* RCCL exchange for sharded engines: one ncclAllGather per pivot.  A resident engine whose grid leaves at least 8 CUs alone
 * (desc.resident_workgroups) KEEPS its grid and its candidate cache: the collective runs on a stream of its own and moves 64-byte records
 * {number, candidate, number} that lie in pinned host memory -- the host writes its own, polls for the others', nothing is copied or
 * synchronised per pivot.  Any other engine serves its searches with one dispatch each from then on; called without a posted search it
 * folds the scan's records on the device into the collective's send buffer (rounds 1-2), after mcf_engine_search_begin it exchanges like a
 * resident one.  id_out/id: the 128-byte ncclUniqueId, created on rank 0 and broadcast by the caller (torch.distributed). */
MCF_API int mcf_comm_unique_id(uint8_t id_out[128]);
MCF_API int mcf_engine_comm_init(mcf_engine *e, const uint8_t id[128], int32_t rank, int32_t world);
/* find_entering_local + all-gather + resolve in one call */
MCF_API int mcf_engine_find_entering_sharded(mcf_engine *e, int32_t *found, int32_t *arc, int64_t *reduced_cost);

/* Host exchange for sharded engines ("per-GPU result slots reduced by the host"): the ranks of ONE node put their
 * 32-byte candidates into POSIX shared memory and read each other's -- an all-gather without a collective library, a fraction of a
 * microsecond per pivot where one ncclAllGather of 32 bytes costs tens.  name: the same on every rank, "/something" (shm_open);
 * all ranks must have returned from mcf_exchange_open (any barrier of the caller's) before the first mcf_exchange_all_gather. */
typedef struct mcf_exchange mcf_exchange;
MCF_API int mcf_exchange_open(mcf_exchange **out, const char *name, int32_t rank, int32_t world);
MCF_API void mcf_exchange_close(mcf_exchange *x);
MCF_API int mcf_exchange_all_gather(mcf_exchange *x, const mcf_candidate *mine, mcf_candidate *all /* [world] */);

/* ------------------------------------------------------------------------------------------------
 * (2) Host driver: NetworkSimplex restated (mirror of the public surface of NS.cs)
 * ---------------------------------------------------------------------------------------------- */

typedef struct mcf_ns mcf_ns;

/* NetworkSimplex(IGraph graph) (NS.cs:119-148): the graph is given as flat arc lists, ids = positions */
MCF_API int mcf_ns_create(mcf_ns **out, int32_t node_count, int32_t arc_count, const int32_t *source, const int32_t *target);
MCF_API void mcf_ns_destroy(mcf_ns *s);
MCF_API int mcf_ns_set_arc_bounds(mcf_ns *s, int32_t arc, int64_t lower, int64_t upper);   /* NS.cs:153-164 */
MCF_API int mcf_ns_set_arc_cost(mcf_ns *s, int32_t arc, int64_t cost);                     /* NS.cs:169-178 */
MCF_API int mcf_ns_set_node_supply(mcf_ns *s, int32_t node, int64_t supply);               /* NS.cs:183-192 */
/* bulk forms of the three setters (NULL = keep defaults: lower 0, upper INF, cost 0, supply 0: NS.cs:614-621) */
MCF_API int mcf_ns_set_problem(mcf_ns *s, const int64_t *lower, const int64_t *upper, const int64_t *cost, const int64_t *supply);
MCF_API int mcf_ns_set_supply_type(mcf_ns *s, int32_t type);                               /* NS.cs:197-201 */
MCF_API int mcf_ns_set_pivot_rule(mcf_ns *s, int32_t rule);                                /* NS.cs:206-210 */
MCF_API int mcf_ns_enable_optimized_pivot(mcf_ns *s, int32_t enable);                      /* NS.cs:532-535 */
/* Vector<long>.Count of the machine whose EnableOptimizedPivot(true) Block Search is reproduced (mcf_engine_desc.vector_width; the
 * reference has no setter, it reads the property: BSPO.cs:74, :115).  Default 4 = x64. */
MCF_API int mcf_ns_set_vector_width(mcf_ns *s, int32_t vector_width);
/* SetOptimizationConfig (NS.cs:557-561; switches auto-configuration off), EnableOptimizations (NS.cs:549-552),
 * SetAutoConfiguration (NS.cs:567-570; the reference's default is ON, and so is this library's) */
MCF_API int mcf_ns_set_optimization_config(mcf_ns *s, const mcf_block_config *config);
MCF_API int mcf_ns_enable_optimizations(mcf_ns *s, int32_t flags);
MCF_API int mcf_ns_set_auto_configuration(mcf_ns *s, int32_t enable);
/* device-side options that have no counterpart in the reference */
MCF_API int mcf_ns_set_device(mcf_ns *s, int32_t device, int32_t int_width /* 32, 64, 0 = narrowest that is safe */,
                              int32_t block_size /* 0 = reference default */, int32_t engine_flags);
/* Several independent solvers of one process on one device (one host thread each): every solver's resident grid gets
 * `resident_workgroups` workgroups -- 256 / K for K solvers -- so that the grids sit on compute units of their own (a workgroup of
 * these grids fills a CU) instead of competing for the same ones; 0 = the whole device (default).  An instance whose arcs no longer
 * fit the registers of so few workgroups keeps reduced costs per arc (the layout of the large instances); the pivots are the same
 * either way.  A process runs at most GPU_MAX_HW_QUEUES (HIP's variable, default 4; read when HIP starts) resident grids per device:
 * set it to K or more before the first HIP call, further solvers serve their searches with one dispatch each. */
MCF_API int mcf_ns_set_device_share(mcf_ns *s, int32_t resident_workgroups);
/* shard the arc scan over `world` ranks exchanging over RCCL (every rank runs the same host loop) */
MCF_API int mcf_ns_set_sharding(mcf_ns *s, const uint8_t nccl_id[128], int32_t rank, int32_t world);

/* the same sharding with the candidates exchanged through shared memory (mcf_exchange_*) instead of RCCL: every rank runs the same
 * host loop, its engine keeps its arc shard in a resident grid; exchange_name as for mcf_exchange_open */
MCF_API int mcf_ns_set_sharding_host(mcf_ns *s, const char *exchange_name, int32_t rank, int32_t world);
/* the same sharding inside ONE process: `shards` engines, one per entry of devices[] (entries may repeat: arc shards rehearsed on fewer
 * GPUs), driven by this solver's single host thread, which posts every search to all of them and reduces their answers itself */
MCF_API int mcf_ns_set_shard_group(mcf_ns *s, int32_t shards, const int32_t *devices);

/* Optional: everything Solve() does before its pivot loop (CheckBounds, TransformToStandardForm, Initialize, creating the
 * engine and copying the SoA arrays into HBM).  mcf_ns_solve() calls it when the caller has not. */
MCF_API int mcf_ns_prepare(mcf_ns *s);
/* Solve() (NS.cs:215-411).  *status receives a mcf_solver_status. */
MCF_API int mcf_ns_solve(mcf_ns *s, int32_t *status);
MCF_API int mcf_ns_status(mcf_ns *s, int32_t *status);                                     /* NS.cs:470 */
MCF_API int mcf_ns_get_flow(mcf_ns *s, int32_t arc, int64_t *flow);                        /* NS.cs:416-429 */
MCF_API int mcf_ns_get_potential(mcf_ns *s, int32_t node, int64_t *potential);             /* NS.cs:434-447 */
MCF_API int mcf_ns_get_total_cost(mcf_ns *s, int64_t *cost);                               /* NS.cs:452-465 */
MCF_API int mcf_ns_get_flows(mcf_ns *s, int64_t *flow_out /* [arc_count] */);
MCF_API int mcf_ns_get_potentials(mcf_ns *s, int64_t *pi_out /* [node_count] */);
MCF_API int mcf_ns_get_arc_upper_bound(mcf_ns *s, int32_t arc, int64_t *upper);            /* NS.cs:519-527 (reduced after Solve, D11) */

/* SolverMetrics (OptimizationTypes.cs:43-70), same three phase buckets */
typedef struct mcf_ns_metrics {
    int64_t iterations;
    double total_solve_us, pivot_search_us, tree_update_us, potential_update_us;
    double setup_us;              /* mcf_ns_prepare: standard form, start basis, engine creation, upload */
    double loop_us;               /* the pivot loop alone (inputs already resident in HBM) */
    int32_t search_arc_num, block_size, int_width, reserved;
    int64_t degenerate_pivots;
    int64_t potential_nodes;      /* nodes whose potential the host walked over: the moved subtree's, or -- inside mcf_ns_solve with 64-bit engines --
                                     the REST of the tree's when that is the smaller side (reduced costs only see differences, so moving
                                     everything else by -sigma is the same search; the common offset is taken out before Solve() returns) */
    mcf_engine_stats engine;
    /* the remaining SolverMetrics fields (OptimizationTypes.cs:53-59), filled like NS.cs:262-270, :344-357 */
    int32_t initial_block_size, final_block_size;      /* Block Search, plain flavour; 0 otherwise (NS.cs:263-270, :348-355) */
    int64_t total_arcs_checked;                        /* see mcf_engine_stats.arcs_checked */
    double average_arcs_checked_per_pivot;
    int32_t baseline_iterations;                       /* (int)(sqrt(m_s) * n * 0.5), NS.cs:276 */
    int32_t config_flags;                              /* MCF_OPT_* in force during the solve (auto-configured or set) */
    double iteration_ratio;
    int32_t reference_selects_cached_pivot;            /* 1: with this configuration the reference would run CachedBlockSearchPivot
                                                          (NS.cs:855-883), which this library does not reproduce (SURVEY.md 8a row a8:
                                                          documented failure, stale cache entries); the plain Block Search ran instead */
    int32_t reserved2;
} mcf_ns_metrics;
MCF_API int mcf_ns_get_metrics(mcf_ns *s, mcf_ns_metrics *out);                            /* NS.cs:584-587 */
/* measurement aid: Solve() stops after max_pivots pivots and reports MCF_NOT_SOLVED (0 = no limit) */
MCF_API int mcf_ns_set_pivot_limit(mcf_ns *s, int64_t max_pivots);
/* optional pivot trace: entering arc of every pivot of the next Solve() (for parity tests) */
MCF_API int mcf_ns_set_trace(mcf_ns *s, int32_t *trace, int64_t capacity);
MCF_API int mcf_ns_get_trace_length(mcf_ns *s, int64_t *length);

/* Stepwise host side, the part a C# host keeps (NS.cs:253, :319-340, :360-388).  None of these searches for an
 * entering arc; they let the sequential part be driven (and tested) with entering arcs supplied by the caller. */
MCF_API int mcf_ns_begin(mcf_ns *s, int32_t *status);            /* CheckBounds + TransformToStandardForm + Initialize */
MCF_API int mcf_ns_apply_pivot(mcf_ns *s, int32_t entering_arc, int32_t *unbounded);   /* FindJoinNode .. UpdatePotentials */
MCF_API int mcf_ns_finish(mcf_ns *s, int32_t *status);           /* CheckFeasibility + lower-bound restore */
/* Measurement / test aid: `count` pivots with the given entering arcs applied back to back, no engine involved -- the sequential half
 * alone, with the walk aids mcf_ns_solve uses (smaller_side != 0: shift whichever side of the tree is smaller; renumber_every > 0: relabel
 * the nodes in thread order whenever the walks since the last relabelling covered that many times the node count).  Fills the metrics'
 * tree_update_us / potential_update_us / loop_us (setup_us = time spent relabelling).  Node ids are the caller's again when it returns. */
MCF_API int mcf_ns_replay(mcf_ns *s, const int32_t *arcs, int64_t count, int32_t smaller_side, double renumber_every);
/* views of the internal SoA after mcf_ns_begin (valid until destroy); sizes: arc_capacity / node_count + 1 */
MCF_API int mcf_ns_internal(mcf_ns *s, int32_t *search_arc_num, int32_t *arc_capacity, const int32_t **source,
                            const int32_t **target, const int64_t **cost, const int8_t **state, const int64_t **pi);
/* views of the spanning tree after mcf_ns_begin / mcf_ns_apply_pivot (valid until the next pivot; node arrays: node_count + 1 entries, the last
 * one is the artificial root; arc arrays: arc_capacity entries): Parent, Pred, SuccNum, PredDir of SpanningTree.cs:10-35 and _flow / _upper.
 * For tools that study the cycle search (NS.cs:925-1010) on real trees; any pointer may be NULL. */
MCF_API int mcf_ns_tree(mcf_ns *s, const int32_t **parent, const int32_t **pred_arc, const int32_t **succ_num, const int8_t **pred_dir,
                        const int64_t **flow, const int64_t **upper);
/* test aid: mcf_engine_check_reduced_costs summed over this solver's engines (after Solve()) */
MCF_API int mcf_ns_check_reduced_costs(mcf_ns *s, int64_t *mismatches);
/* what the last mcf_ns_apply_pivot changed: the engine calls a host would make */
MCF_API int mcf_ns_last_pivot(mcf_ns *s, int32_t *n_state, int32_t arcs[2], int8_t states[2], int32_t *n_nodes,
                              const int32_t **nodes, int64_t *sigma);

/* ------------------------------------------------------------------------------------------------
 * (3) Solution validator on the device: the checks of
 *     src/MinCostFlow.Core/Lemon/Validation/SolutionValidator.cs as reductions over the arcs and the nodes.
 *     Same arithmetic as the reference (C# long, unchecked: it wraps).  Instead of message strings the result
 *     carries, per check, the number of messages the reference would add and the lowest arc / node id among them.
 * ---------------------------------------------------------------------------------------------- */
typedef enum mcf_validation_kind {
    MCF_VAL_CONSERVATION = 0,   /* node: net flow vs supply under the supply type        SolutionValidator.cs:75-99   */
    MCF_VAL_LOWER = 1,          /* arc : flow < lower                                    :111-115 */
    MCF_VAL_UPPER = 2,          /* arc : flow > upper                                    :117-121 */
    MCF_VAL_SLACK_POS = 3,      /* arc : reduced cost > 0 but flow != lower              :164-168 */
    MCF_VAL_SLACK_NEG = 4,      /* arc : reduced cost < 0 but flow != upper              :170-174 */
    MCF_VAL_NODE_DUAL = 5,      /* node: sign of pi against the supply type              :203-207, :218-222 */
    MCF_VAL_NODE_SLACK = 6,     /* node: pi != 0 but net flow != supply                  :208-213, :223-228 */
    MCF_VAL_OBJECTIVE = 7,      /* sum flow*cost != reported cost                        :257-262 */
    MCF_VAL_DUAL_COST = 8,      /* dual cost != reported cost                            :333-339 */
    MCF_VAL_STATUS = 9,         /* solver status is not Optimal (nothing else is checked) :28-33 */
    MCF_VAL_KINDS = 10
} mcf_validation_kind;
/* a third supply type for the validator only: the reference's `_ =>` equality branch (SolutionValidator.cs:83) */
#define MCF_SUPPLY_EQ 2

typedef struct mcf_validation {
    int32_t valid;                      /* ValidationResult.IsValid: no message at all */
    int32_t supply_type;
    int64_t objective;                  /* calculatedCost (:232-255) */
    int64_t dual_cost;                  /* ValidationResult.DualCost (:270-331) */
    int64_t errors[MCF_VAL_KINDS];
    int64_t first[MCF_VAL_KINDS];       /* lowest failing arc / node id, -1 when the check passes */
    double kernel_us;                   /* HIP events around one run: memset + validate_arcs + validate_nodes + validate_fold */
    int64_t algorithmic_bytes;          /* 40 B per arc + 40 B per node, see DESIGN.md */
} mcf_validation;

typedef struct mcf_validator mcf_validator;
MCF_API int mcf_validator_create(mcf_validator **out, int32_t device, int32_t node_count, int32_t arc_count);
MCF_API void mcf_validator_destroy(mcf_validator *v);
/* Host arrays, copied.  The network (first six arrays) and the solution (flow, pi) may be uploaded separately:
 * pass NULL for everything that stays as it is on the device. */
MCF_API int mcf_validator_upload(mcf_validator *v, const int32_t *source, const int32_t *target, const int64_t *lower,
                                 const int64_t *upper, const int64_t *cost, const int64_t *supply, const int64_t *flow,
                                 const int64_t *pi);
MCF_API int mcf_validator_run(mcf_validator *v, int32_t supply_type, int64_t reported_cost, mcf_validation *out);
/* SolutionValidator(graph, solver).Validate() for a solver of this library: reads what the reference's validator
 * reads through the solver's getters (GetFlow, GetPotential, GetArcLowerBound = the original lower bound,
 * GetArcUpperBound = the bound as mutated by Solve(), NS.cs:519-527, difference D11) */
MCF_API int mcf_ns_validate(mcf_ns *s, mcf_validation *out);

/* ------------------------------------------------------------------------------------------------
 * (4) Problem sources (build-owned; the reference ships NETGEN outputs but no generator: SURVEY.md F6, 8d)
 * ---------------------------------------------------------------------------------------------- */

typedef struct mcf_problem {
    int32_t node_count, arc_count;
    int32_t *source, *target;          /* [arc_count], 0-based */
    int64_t *lower, *upper, *cost;     /* [arc_count] */
    int64_t *supply;                   /* [node_count] */
} mcf_problem;
MCF_API void mcf_problem_free(mcf_problem *p);
/* NETGEN-like transshipment network: SplitMix64(seed); n_src sources / n_snk sinks, total supply 1000*n_src,
 * skeleton chains source -> ... -> sink (cost = max_cost, capacity >= the chain's supply) make it feasible;
 * remaining arcs uniform; arcs are emitted grouped by tail node like NETGEN's output files. */
MCF_API int mcf_gen_netgen_like(mcf_problem *out, uint64_t seed, int32_t nodes, int32_t arcs, int32_t n_src,
                                int32_t n_snk, int64_t min_cost, int64_t max_cost, int64_t min_cap, int64_t max_cap);
/* complete bipartite assignment n x n, cost U[min_cost,max_cost], supply +1/-1, bounds [0,1]
 * (shape of src/MinCostFlow.Problems/Generators/ProblemGenerator.cs:194-232) */
MCF_API int mcf_gen_assignment(mcf_problem *out, uint64_t seed, int32_t n, int64_t min_cost, int64_t max_cost);
/* DIMACS min-cost-flow reader / writer (Loaders/DimacsReader.cs:36-147; lemon/dimacs.h:129-186) */
MCF_API int mcf_dimacs_read(mcf_problem *out, const char *path);
MCF_API int mcf_dimacs_write(const mcf_problem *p, const char *path);
/* .sol files (Loaders/SolutionLoader.cs:59-176 reader, :186-210 writer): "s COST", "f ARC FLOW" (0-based arc id, what the
 * reference writes) or "f SRC DST FLOW" (1-based end points, what its bundled Gurobi solutions hold), "p NODE POTENTIAL".
 * The reader fills flow[arc_count] (and pi[node_count] when given); end-point flows are split over parallel arcs by
 * increasing cost.  pi may be NULL in both. */
MCF_API int mcf_solution_write(const char *path, int64_t cost, int32_t arc_count, const int64_t *flow, int32_t node_count, const int64_t *pi);
MCF_API int mcf_solution_read(const char *path, const mcf_problem *p, int64_t *cost, int32_t *has_cost, int64_t *flow, int64_t *pi, int32_t *has_pi);

#ifdef __cplusplus
}
#endif
#endif /* MCF_HIP_H */
