/*
 * oracle/ns_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's primal network simplex, used ONLY as
 * the checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing under mincostflow_amd/ may call into this file.
 *
 * What pins this restatement (tests/test_oracle_golden.py):
 *   - the optimal COSTS of the reference's 37 .sol fixtures and 2 csv rows, in every mode and rule;
 *   - the exact FLOWS of the C# unit tests (NetworkSimplexTests.cs, OptimizationTests.cs) and
 *     LEMON's 21-case status / cost table (min_cost_flow_test.cc);
 *   - the README example (Infeasible as written, 71 with supply 13).
 * What nothing the reference holds pins: the PIVOT ORDER.  The C# reference has no toolchain in the
 * image and the vendored LEMON cannot be compiled from its own sources (lemon/config.h is generated
 * by its CMake build), so there is no oracle/_ref.  Pivot-for-pivot parity of the HIP path is parity
 * with this file's reading of the C# source; LEMON-mode internals that do not change the optimum
 * (arc mixing order, heuristic initial pivots) are pinned by costs and statuses only.
 *
 * Three semantics modes (differences D1-D13):
 *   NSO_SEM_LEMON      lemon-1.3.1/lemon/network_simplex.h (run(), arc mixing, EQ/LEQ/GEQ
 *                      init, heuristic initial pivots, final potential shift)
 *   NSO_SEM_CSHARP     src/MinCostFlow.Core/Lemon/Algorithms/NetworkSimplex.cs, plain pivot
 *                      rules, auto-configuration off (fixed block size)
 *   NSO_SEM_CSHARP_OPT same host driver, pivot rules of
 *                      .../Algorithms/Internal/BlockSearchPivotOptimized.cs
 *                      (what EnableOptimizedPivot(true) selects), with the vector width V of the
 *                      reference's host (nso_set_vector_width; default 4 = x64): for V > 0 the Block
 *                      Search there scans to the end of the range after a hit in its "SIMD" part
 *                      (BSPO.cs:84-99 falls through with cnt == 0), V = 0 is the scalar-only path
 *
 * Every function cites the reference lines it follows.  "NS.cs" below means
 * src/MinCostFlow.Core/Lemon/Algorithms/NetworkSimplex.cs, "ns.h" means
 * lemon-1.3.1/lemon/network_simplex.h, "BSPO.cs" means
 * src/MinCostFlow.Core/Lemon/Algorithms/Internal/BlockSearchPivotOptimized.cs.
 */
#define _POSIX_C_SOURCE 199309L   /* clock_gettime under -std=c11 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#define NSO_API __attribute__((visibility("default")))

enum { NSO_SEM_LEMON = 0, NSO_SEM_CSHARP = 1, NSO_SEM_CSHARP_OPT = 2 };
enum { NSO_RULE_FIRST = 0, NSO_RULE_BEST = 1, NSO_RULE_BLOCK = 2 };   /* Types/PivotRule.cs:7-41 */
enum { NSO_GEQ = 0, NSO_LEQ = 1 };                                    /* Types/SupplyType.cs */
enum { NSO_NOT_SOLVED = 0, NSO_OPTIMAL = 1, NSO_INFEASIBLE = 2, NSO_UNBOUNDED = 3 }; /* Types/SolverStatus.cs:7-34 */

enum { ST_UPPER = -1, ST_TREE = 0, ST_LOWER = 1, DIR_DOWN = -1, DIR_UP = 1 };  /* SpanningTree.cs:53-71, ns.h:179-189 */

typedef struct ns_oracle {
    int n, m, sem, rule, stype, arc_mixing;
    /* the caller's problem, original numbering */
    int32_t *osrc, *otgt;
    int64_t *olower, *oupper, *ocost, *osupply;
    /* original -> internal numbering */
    int32_t *arc_id, *node_id;
    /* internal SoA, arcs: m + 2n, nodes: n + 1 */
    int32_t *src, *tgt;
    int64_t *lower, *cap, *cost, *flow;
    int8_t *state;
    int64_t *supply, *pi;
    int32_t *parent, *pred, *thread, *rev_thread, *succ_num, *last_succ, *dirty;
    int8_t *pred_dir;
    int root, search_arc_num, all_arc_num, feas_lo, feas_hi;
    int64_t sum_supply, art_cost, vmax, vinf;
    /* current pivot */
    int in_arc, join, u_in, v_in, u_out, v_out;
    int64_t delta;
    /* pivot-rule state */
    int next_arc, block_size, block_size_req;
    int vector_width;                /* BSPO.cs:74,115: Vector<long>.Count of the reference's host; 0 = not hardware accelerated */
    /* OptimizationConfig (OptimizationTypes.cs:24-38), consumed by the plain C# BlockSearchPivot only */
    int cfg_flags, cfg_min_block, cfg_max_block, cfg_consec;
    double cfg_ratio, cfg_grow, cfg_shrink, cfg_low, cfg_high;
    int auto_config;                 /* NS.cs:90 _useAutoConfiguration (the oracle's default is OFF; tests switch it on) */
    int dyn_min_block, low_hits, high_hits, initial_block_size, would_cache;
    int64_t arcs_checked;            /* SolverMetrics.TotalArcsChecked, NS.cs:286-290 */
    int timing;                      /* the reference's three stopwatches (NS.cs:95-98, :285-289, :330-339), off unless asked for */
    double t_search, t_tree, t_pot;  /* ns */
    int status, initialized;
    int64_t pivots, init_pivots, max_iter;
    int64_t last_sigma;
    int last_subtree;
} ns_oracle;

/* ------------------------------------------------------------------ lifecycle */

NSO_API void nso_destroy(ns_oracle *o)
{
    if (!o) return;
    free(o->osrc); free(o->otgt); free(o->olower); free(o->oupper); free(o->ocost); free(o->osupply);
    free(o->arc_id); free(o->node_id); free(o->src); free(o->tgt); free(o->lower); free(o->cap);
    free(o->cost); free(o->flow); free(o->state); free(o->supply); free(o->pi); free(o->parent);
    free(o->pred); free(o->thread); free(o->rev_thread); free(o->succ_num); free(o->last_succ);
    free(o->dirty); free(o->pred_dir); free(o);
}

static void *dupmem(const void *p, size_t bytes)
{
    void *q = malloc(bytes ? bytes : 1);
    if (p && bytes) memcpy(q, p, bytes);
    return q;
}

/* upper[i] == INT64_MAX means "no upper bound" in the caller's problem. */
NSO_API ns_oracle *nso_create(int n, int m, const int32_t *src, const int32_t *tgt,
                              const int64_t *lower, const int64_t *upper, const int64_t *cost,
                              const int64_t *supply, int semantics, int rule, int supply_type,
                              int arc_mixing, int block_size)
{
    ns_oracle *o = (ns_oracle *)calloc(1, sizeof(*o));
    o->n = n; o->m = m; o->sem = semantics; o->rule = rule; o->stype = supply_type;
    o->arc_mixing = arc_mixing; o->block_size_req = block_size;
    o->osrc = dupmem(src, sizeof(int32_t) * m); o->otgt = dupmem(tgt, sizeof(int32_t) * m);
    o->olower = dupmem(lower, sizeof(int64_t) * m); o->oupper = dupmem(upper, sizeof(int64_t) * m);
    o->ocost = dupmem(cost, sizeof(int64_t) * m); o->osupply = dupmem(supply, sizeof(int64_t) * n);
    size_t A = (size_t)m + 2 * (size_t)n + 1, N = (size_t)n + 1;   /* NS.cs:130, ns.h:911-912 */
    o->arc_id = calloc(m + 1, sizeof(int32_t)); o->node_id = calloc(N, sizeof(int32_t));
    o->src = calloc(A, sizeof(int32_t)); o->tgt = calloc(A, sizeof(int32_t));
    o->lower = calloc(A, sizeof(int64_t)); o->cap = calloc(A, sizeof(int64_t));
    o->cost = calloc(A, sizeof(int64_t)); o->flow = calloc(A, sizeof(int64_t));
    o->state = calloc(A, sizeof(int8_t));
    o->supply = calloc(N, sizeof(int64_t)); o->pi = calloc(N, sizeof(int64_t));
    o->parent = calloc(N, sizeof(int32_t)); o->pred = calloc(N, sizeof(int32_t));
    o->thread = calloc(N, sizeof(int32_t)); o->rev_thread = calloc(N, sizeof(int32_t));
    o->succ_num = calloc(N, sizeof(int32_t)); o->last_succ = calloc(N, sizeof(int32_t));
    o->dirty = calloc(N + 1, sizeof(int32_t)); o->pred_dir = calloc(N, sizeof(int8_t));
    o->vmax = INT64_MAX;                                     /* NS.cs:126, ns.h:652 */
    o->vinf = (semantics == NSO_SEM_LEMON) ? INT64_MAX       /* ns.h:653-654 (integer: MAX) */
                                           : INT64_MAX / 2;  /* NS.cs:127 */
    o->status = NSO_NOT_SOLVED;
    /* new OptimizationConfig(): OptimizationTypes.cs:26-37 */
    o->cfg_flags = 0; o->cfg_max_block = 100; o->cfg_min_block = 25; o->cfg_grow = 1.2; o->cfg_shrink = 0.8;
    o->cfg_low = 0.05; o->cfg_high = 0.3; o->cfg_consec = 3; o->cfg_ratio = 0.125;
    o->vector_width = 4;             /* x64 with AVX2 (and .NET 8 on AVX-512 hardware): Vector<long>.Count == 4 */
    return o;
}

/* SetOptimizationConfig (NS.cs:557-561): replaces the configuration and switches auto-configuration off */
NSO_API void nso_set_config(ns_oracle *o, int flags, int min_block, int max_block, int consec, double ratio,
                            double grow, double shrink, double low, double high)
{
    o->cfg_flags = flags; o->cfg_min_block = min_block; o->cfg_max_block = max_block; o->cfg_consec = consec;
    o->cfg_ratio = ratio; o->cfg_grow = grow; o->cfg_shrink = shrink; o->cfg_low = low; o->cfg_high = high;
    o->auto_config = 0;
}
/* the reference has no such setter: Vector<long>.Count is a property of the machine it runs on (BSPO.cs:74, :115) */
NSO_API int nso_set_vector_width(ns_oracle *o, int v)
{
    if (v != 0 && v != 2 && v != 4 && v != 8) return 0;
    o->vector_width = v;
    return 1;
}
/* SetAutoConfiguration (NS.cs:567-570) */
NSO_API void nso_set_auto_config(ns_oracle *o, int enable) { o->auto_config = enable; }

enum { OPT_ADAPTIVE = 1, OPT_SMALL_DENSE = 2, OPT_CACHING = 4 };   /* OptimizationTypes.cs:12-14 */

/* NS.cs:237-242: ProblemAnalyzer.Analyze (ProblemAnalyzer.cs:21-106) + OptimizationSelector.SelectConfiguration
 * (OptimizationSelector.cs:14-95), the fields BlockSearchPivot and CreatePivotRuleFinder read. */
static void auto_configure(ns_oracle *o)
{
    int n = o->n, m = o->m;
    long long max_possible = (long long)n * (n - 1);
    double density = max_possible > 0 ? (double)m / max_possible : 0;        /* ProblemAnalyzer.cs:35-36 */
    int *deg = calloc(n > 0 ? n : 1, sizeof(int));
    for (int e = 0; e < m; e++) { deg[o->osrc[e]]++; deg[o->otgt[e]]++; }     /* :68-75 out + in */
    int total = 0;
    for (int v = 0; v < n; v++) total += deg[v];
    double avg = n > 0 ? (double)total / n : 0, var = 0;                      /* :80-92 */
    if (n > 0) { for (int v = 0; v < n; v++) { double d = deg[v] - avg; var += d * d; } var /= n; }
    double cv = avg > 0 ? sqrt(var) / avg : 0;                                /* :97 */
    free(deg);
    int dense = density > 0.01 || m > 10000, sparse = density < 0.005;        /* :58-60 */
    /* OptimizationSelector.cs:16-95 starts from a fresh OptimizationConfig */
    o->cfg_flags = 0; o->cfg_grow = 1.2; o->cfg_shrink = 0.8; o->cfg_consec = 3;
    if (dense) { o->cfg_flags |= OPT_SMALL_DENSE; o->cfg_min_block = 10; o->cfg_max_block = 50; }
    else { o->cfg_min_block = 25; o->cfg_max_block = 100; }
    if (cv > 0.5) { o->cfg_flags |= OPT_ADAPTIVE; o->cfg_grow = 1.3; o->cfg_shrink = 0.7; o->cfg_consec = 2; }
    else if (cv > 0.3) o->cfg_flags |= OPT_ADAPTIVE;
    if (sparse && m < 50000) o->cfg_flags |= OPT_CACHING;
    o->cfg_low = m > 10000 ? 0.03 : 0.05;
    o->cfg_high = m > 10000 ? 0.25 : 0.3;
    o->cfg_ratio = m > 100000 ? 0.0625 : (m > 10000 ? 0.125 : 0.25);
}

/* ------------------------------------------------------------------ numbering */

/* LEMON: ListDigraph hands out nodes newest-first (list_graph.h:111-117,171-190) and the
 * DIMACS reader adds node k as the k-th node (dimacs.h:149-152), so NodeIt order is
 * original id n-1, n-2, ..., 0 and _node_id follows it (ns.h:935-938).  ArcIt walks the
 * nodes in that order and each node's out-arcs newest-first (list_graph.h:120-138,192-214).
 * Arc mixing: ns.h:939-957.   C#: identity (NS.cs:605-613, DimacsReader.cs:96-99). */
static void build_numbering(ns_oracle *o)
{
    int n = o->n, m = o->m;
    if (o->sem != NSO_SEM_LEMON) {
        for (int i = 0; i < n; i++) o->node_id[i] = i;
        for (int e = 0; e < m; e++) o->arc_id[e] = e;
        return;
    }
    for (int v = 0; v < n; v++) o->node_id[v] = n - 1 - v;
    /* per-source newest-first lists */
    int32_t *head = malloc(sizeof(int32_t) * (n + 1)), *next = malloc(sizeof(int32_t) * (m + 1));
    for (int v = 0; v < n; v++) head[v] = -1;
    for (int e = 0; e < m; e++) { next[e] = head[o->osrc[e]]; head[o->osrc[e]] = e; }
    int mixing = o->arc_mixing && n > 1;
    int skip = m / n > 3 ? m / n : 3;
    int i = 0, j = 0;
    for (int v = n - 1; v >= 0; v--) {
        for (int e = head[v]; e != -1; e = next[e]) {
            o->arc_id[e] = i;
            if (mixing) { if ((i += skip) >= m) i = ++j; }
            else i++;
        }
    }
    free(head); free(next);
}

/* ------------------------------------------------------------------ init */

/* ns.h:1059-1236 (LEMON) */
static int init_lemon(ns_oracle *o)
{
    int n = o->n, m = o->m;
    if (n == 0) return 0;                                            /* ns.h:1060 */
    for (int v = 0; v < n; v++) o->supply[o->node_id[v]] = o->osupply[v];
    for (int e = 0; e < m; e++) {
        int i = o->arc_id[e];
        o->src[i] = o->node_id[o->osrc[e]]; o->tgt[i] = o->node_id[o->otgt[e]];
        o->lower[i] = o->olower[e]; o->cost[i] = o->ocost[e];
    }
    o->sum_supply = 0;
    for (int i = 0; i < n; i++) o->sum_supply += o->supply[i];       /* ns.h:1063-1066 */
    if (!((o->stype == NSO_GEQ && o->sum_supply <= 0) ||
          (o->stype == NSO_LEQ && o->sum_supply >= 0))) return 0;    /* ns.h:1067-1068 */
    for (int e = 0; e < m; e++) {                                    /* ns.h:1075-1085 (_has_lower) */
        int i = o->arc_id[e];
        int64_t c = o->lower[i], up = o->oupper[e];
        if (c >= 0) o->cap[i] = up < o->vmax ? up - c : o->vinf;
        else        o->cap[i] = up < o->vmax + c ? up - c : o->vinf;
        o->supply[o->src[i]] -= c;
        o->supply[o->tgt[i]] += c;
    }
    int64_t ART = INT64_MAX / 2 + 1;                                 /* ns.h:1094-1095 */
    o->art_cost = ART;
    for (int i = 0; i < m; i++) { o->flow[i] = 0; o->state[i] = ST_LOWER; }   /* ns.h:1105-1108 */
    int root = o->root = n;                                          /* ns.h:1111-1119 */
    o->parent[root] = -1; o->pred[root] = -1; o->thread[root] = 0; o->rev_thread[0] = root;
    o->succ_num[root] = n + 1; o->last_succ[root] = root - 1;
    o->supply[root] = -o->sum_supply; o->pi[root] = 0;
    if (o->sum_supply == 0) {                                        /* EQ, ns.h:1122-1151 */
        o->search_arc_num = m; o->all_arc_num = m + n;
        for (int u = 0, e = m; u != n; ++u, ++e) {
            o->parent[u] = root; o->pred[u] = e; o->thread[u] = u + 1; o->rev_thread[u + 1] = u;
            o->succ_num[u] = 1; o->last_succ[u] = u; o->cap[e] = o->vinf; o->state[e] = ST_TREE;
            if (o->supply[u] >= 0) {
                o->pred_dir[u] = DIR_UP; o->pi[u] = 0; o->src[e] = u; o->tgt[e] = root;
                o->flow[e] = o->supply[u]; o->cost[e] = 0;
            } else {
                o->pred_dir[u] = DIR_DOWN; o->pi[u] = ART; o->src[e] = root; o->tgt[e] = u;
                o->flow[e] = -o->supply[u]; o->cost[e] = ART;
            }
        }
    } else if (o->sum_supply > 0) {                                  /* LEQ, ns.h:1152-1192 */
        o->search_arc_num = m + n;
        int f = m + n;
        for (int u = 0, e = m; u != n; ++u, ++e) {
            o->parent[u] = root; o->thread[u] = u + 1; o->rev_thread[u + 1] = u;
            o->succ_num[u] = 1; o->last_succ[u] = u;
            if (o->supply[u] >= 0) {
                o->pred_dir[u] = DIR_UP; o->pi[u] = 0; o->pred[u] = e; o->src[e] = u; o->tgt[e] = root;
                o->cap[e] = o->vinf; o->flow[e] = o->supply[u]; o->cost[e] = 0; o->state[e] = ST_TREE;
            } else {
                o->pred_dir[u] = DIR_DOWN; o->pi[u] = ART; o->pred[u] = f; o->src[f] = root; o->tgt[f] = u;
                o->cap[f] = o->vinf; o->flow[f] = -o->supply[u]; o->cost[f] = ART; o->state[f] = ST_TREE;
                o->src[e] = u; o->tgt[e] = root; o->cap[e] = o->vinf; o->flow[e] = 0; o->cost[e] = 0;
                o->state[e] = ST_LOWER;
                ++f;
            }
        }
        o->all_arc_num = f;
    } else {                                                         /* GEQ, ns.h:1193-1233 */
        o->search_arc_num = m + n;
        int f = m + n;
        for (int u = 0, e = m; u != n; ++u, ++e) {
            o->parent[u] = root; o->thread[u] = u + 1; o->rev_thread[u + 1] = u;
            o->succ_num[u] = 1; o->last_succ[u] = u;
            if (o->supply[u] <= 0) {
                o->pred_dir[u] = DIR_DOWN; o->pi[u] = 0; o->pred[u] = e; o->src[e] = root; o->tgt[e] = u;
                o->cap[e] = o->vinf; o->flow[e] = -o->supply[u]; o->cost[e] = 0; o->state[e] = ST_TREE;
            } else {
                o->pred_dir[u] = DIR_UP; o->pi[u] = -ART; o->pred[u] = f; o->src[f] = u; o->tgt[f] = root;
                o->cap[f] = o->vinf; o->flow[f] = o->supply[u]; o->state[f] = ST_TREE; o->cost[f] = ART;
                o->src[e] = root; o->tgt[e] = u; o->cap[e] = o->vinf; o->flow[e] = 0; o->cost[e] = 0;
                o->state[e] = ST_LOWER;
                ++f;
            }
        }
        o->all_arc_num = f;
    }
    o->feas_lo = o->search_arc_num; o->feas_hi = o->all_arc_num;     /* ns.h:1610-1612 */
    return 1;
}

/* NS.cs:624-699 (CheckBounds, TransformToStandardForm, Initialize), :713-845 (GEQ/LEQ) */
static int init_csharp(ns_oracle *o)
{
    int n = o->n, m = o->m;
    for (int e = 0; e < m; e++) if (o->oupper[e] != INT64_MAX && o->oupper[e] < o->olower[e]) return 0; /* NS.cs:624-634 */
    for (int v = 0; v < n; v++) o->supply[v] = o->osupply[v];
    for (int e = 0; e < m; e++) {
        o->src[e] = o->osrc[e]; o->tgt[e] = o->otgt[e]; o->cost[e] = o->ocost[e];
        o->lower[e] = o->olower[e];
        o->cap[e] = o->oupper[e] == INT64_MAX ? o->vinf : o->oupper[e];      /* NS.cs:616 default INF */
        if (o->olower[e] != 0) {                                             /* NS.cs:639-653 */
            o->supply[o->src[e]] -= o->olower[e];
            o->supply[o->tgt[e]] += o->olower[e];
            o->cap[e] -= o->olower[e];
        }
    }
    o->sum_supply = 0;
    for (int v = 0; v < n; v++) o->sum_supply += o->supply[v];               /* NS.cs:656-660 */
    int64_t maxc = 0;
    for (int e = 0; e < m; e++) { int64_t a = o->cost[e] < 0 ? -o->cost[e] : o->cost[e]; if (a > maxc) maxc = a; }
    int64_t ART = o->art_cost = (maxc + 1) * (int64_t)n;                     /* NS.cs:663-668 */
    int root = o->root = n;                                                  /* NS.cs:674-683 */
    o->parent[root] = -1; o->pred[root] = -1; o->thread[root] = 0; o->rev_thread[0] = root;
    o->succ_num[root] = n + 1; o->last_succ[root] = n - 1; o->pred_dir[root] = 0; o->pi[root] = 0;
    for (int e = 0; e < m; e++) { o->state[e] = ST_LOWER; o->flow[e] = 0; }  /* NS.cs:716-720 */
    o->search_arc_num = m + n;                                               /* NS.cs:722 */
    int f = m + n;
    for (int u = 0; u < n; u++) o->thread[u] = u + 1;                        /* NS.cs:726-736 */
    if (n > 0) o->thread[n - 1] = root;
    for (int u = 0; u < n; u++) o->rev_thread[o->thread[u]] = u;
    for (int u = 0, e = m; u < n; u++, e++) {
        o->parent[u] = root; o->succ_num[u] = 1; o->last_succ[u] = u;
        if (o->stype == NSO_GEQ) {                                           /* NS.cs:744-771 */
            if (o->supply[u] <= 0) {
                o->pred_dir[u] = DIR_DOWN; o->pi[u] = 0; o->pred[u] = e; o->src[e] = root; o->tgt[e] = u;
                o->cap[e] = o->vinf; o->flow[e] = -o->supply[u]; o->cost[e] = 0; o->state[e] = ST_TREE;
            } else {
                o->pred_dir[u] = DIR_UP; o->pi[u] = -ART; o->pred[u] = f; o->src[f] = u; o->tgt[f] = root;
                o->cap[f] = o->vinf; o->flow[f] = o->supply[u]; o->state[f] = ST_TREE; o->cost[f] = ART;
                o->src[e] = root; o->tgt[e] = u; o->cap[e] = o->vinf; o->flow[e] = 0; o->cost[e] = 0;
                o->state[e] = ST_LOWER;
                f++;
            }
        } else {                                                             /* NS.cs:811-838 */
            if (o->supply[u] >= 0) {
                o->pred_dir[u] = DIR_UP; o->pi[u] = 0; o->pred[u] = e; o->src[e] = u; o->tgt[e] = root;
                o->cap[e] = o->vinf; o->flow[e] = o->supply[u]; o->cost[e] = 0; o->state[e] = ST_TREE;
            } else {
                o->pred_dir[u] = DIR_DOWN; o->pi[u] = ART; o->pred[u] = f; o->src[f] = root; o->tgt[f] = u;
                o->cap[f] = o->vinf; o->flow[f] = -o->supply[u]; o->state[f] = ST_TREE; o->cost[f] = ART;
                o->src[e] = u; o->tgt[e] = root; o->cap[e] = o->vinf; o->flow[e] = 0; o->cost[e] = 0;
                o->state[e] = ST_LOWER;
                f++;
            }
        }
    }
    if (n > 0) { o->thread[n - 1] = root; o->rev_thread[root] = n - 1; }     /* NS.cs:776-777 */
    o->all_arc_num = f;
    /* NS.cs:689 overwrites _allArcNum with _searchArcNum, so CheckFeasibility (NS.cs:1272-1283)
     * only looks at [m, m+n) -- difference D9. */
    o->feas_lo = m; o->feas_hi = o->search_arc_num;
    return 1;
}

static void init_rule(ns_oracle *o)
{
    o->next_arc = 0;
    int base = (int)sqrt((double)o->search_arc_num);
    o->low_hits = o->high_hits = 0; o->arcs_checked = 0; o->would_cache = 0;
    if (o->sem == NSO_SEM_CSHARP) {
        /* NS.cs:1304-1336 (BlockSearchPivot constructor) with the configuration in force */
        if (o->auto_config) auto_configure(o);
        o->dyn_min_block = (int)(base * o->cfg_ratio);
        if (o->dyn_min_block < o->cfg_min_block) o->dyn_min_block = o->cfg_min_block;
        int b = base;
        if (o->cfg_flags & OPT_SMALL_DENSE) {
            double density = (double)o->search_arc_num / o->n;
            if (density > 10) b = base / 4 < 50 ? base / 4 : 50;
        }
        if (b < o->dyn_min_block) b = o->dyn_min_block;
        o->block_size = o->block_size_req > 0 ? o->block_size_req : b;
        if (o->rule == NSO_RULE_BLOCK && (o->cfg_flags & OPT_CACHING)) {
            /* NS.cs:855-883: the reference would hand the search to CachedBlockSearchPivot here (not restated: SURVEY.md 8a, a8) */
            int nn = (int)((unsigned)o->n * (unsigned)o->n);
            double d2 = (double)o->search_arc_num / nn;
            o->would_cache = d2 < 0.01 && o->search_arc_num < 10000;
        }
    } else if (o->block_size_req > 0) {
        o->block_size = o->block_size_req;
    } else {
        o->block_size = base > 10 ? base : 10;             /* ns.h:369-374, BSPO.cs:27-28 */
    }
    o->initial_block_size = o->block_size;
}

NSO_API int nso_init(ns_oracle *o)
{
    build_numbering(o);
    int ok = (o->sem == NSO_SEM_LEMON) ? init_lemon(o) : init_csharp(o);
    o->initialized = 1;
    if (!ok) { o->status = NSO_INFEASIBLE; return 0; }
    init_rule(o);
    {   /* NS.cs:280 iteration guard (D12); LEMON has none */
        int64_t nm = (int64_t)o->n * (int64_t)o->m;
        o->max_iter = (o->sem == NSO_SEM_LEMON) ? INT64_MAX : (nm > 1000000 ? nm : 1000000);
    }
    return 1;
}

/* ------------------------------------------------------------------ entering arc */

static inline int64_t rc(const ns_oracle *o, int e)
{   /* NS.cs:1351-1352, ns.h:383 */
    return (int64_t)o->state[e] * (o->cost[e] + o->pi[o->src[e]] - o->pi[o->tgt[e]]);
}

/* ns.h:326-336, NS.cs:1644-1667, BSPO.cs:251-289 (the state==0 skip there changes nothing:
 * a tree arc has c == 0, never < min) */
static int find_best(ns_oracle *o)
{
    int64_t min = 0; int best = -1;
    for (int e = 0; e < o->search_arc_num; e++) {
        int64_t c = rc(o, e);
        if (c < min) { min = c; best = e; }
    }
    if (min < 0) { o->in_arc = best; return 1; }
    return 0;
}

/* ns.h:278-297, NS.cs:1607-1636, BSPO.cs:176-232 */
static int find_first(ns_oracle *o)
{
    for (int e = o->next_arc; e < o->search_arc_num; e++)
        if (rc(o, e) < 0) { o->in_arc = e; o->next_arc = e + 1; return 1; }
    for (int e = 0; e < o->next_arc; e++)
        if (rc(o, e) < 0) { o->in_arc = e; o->next_arc = e + 1; return 1; }
    return 0;
}

/* ns.h:378-409 and NS.cs:1339-1441: one continuous cyclic scan, next_arc = last scanned arc.
 * C# only: arcsChecked (NS.cs:1345-1372) and the adaptive block size (NS.cs:1400-1438). */
static int find_block_plain(ns_oracle *o)
{
    int64_t min = 0; int cnt = o->block_size, e, best = -1, checked = 0;
    for (e = o->next_arc; e < o->search_arc_num; e++) {
        checked++;
        int64_t c = rc(o, e);
        if (c < min) { min = c; best = e; }
        if (--cnt == 0) { if (min < 0) goto search_end; cnt = o->block_size; }
    }
    for (e = 0; e < o->next_arc; e++) {
        checked++;
        int64_t c = rc(o, e);
        if (c < min) { min = c; best = e; }
        if (--cnt == 0) { if (min < 0) goto search_end; cnt = o->block_size; }
    }
    if (min >= 0) { if (o->sem == NSO_SEM_CSHARP) o->arcs_checked += checked; return 0; }   /* NS.cs:290 adds the count either way */
search_end:
    if (o->sem == NSO_SEM_CSHARP) o->arcs_checked += checked;
    o->next_arc = e; o->in_arc = best;
    if (o->sem == NSO_SEM_CSHARP && (o->cfg_flags & OPT_ADAPTIVE)) {
        int scanned = checked;
        double hit = scanned > 0 ? 1.0 / scanned : 0;
        if (hit < o->cfg_low) {
            o->high_hits = 0; o->low_hits++;
            if (o->low_hits >= o->cfg_consec) {
                int ns = (int)(o->block_size * o->cfg_shrink);
                o->block_size = ns > o->dyn_min_block ? ns : o->dyn_min_block;
                o->low_hits = 0;
            }
        } else if (hit > o->cfg_high) {
            o->low_hits = 0; o->high_hits++;
            if (o->high_hits >= o->cfg_consec) {
                int ns = (int)(o->block_size * o->cfg_grow);
                o->block_size = ns < o->cfg_max_block ? ns : o->cfg_max_block;
                o->high_hits = 0;
            }
        } else { o->low_hits = 0; o->high_hits = 0; }
    }
    return 1;
}

/* BSPO.cs:69-156 ProcessArcRange + ProcessArcRangeSIMD on bare arrays.  V = Vector<long>.Count of the host the reference runs on
 * (4 on AVX2 / AVX-512 x64 under .NET 8, 2 on NEON), 0 = Vector.IsHardwareAccelerated == false.
 * The "SIMD" function (BSPO.cs:113-156) is a scalar loop over groups of V arcs (the vector it loads at :122 is never used) --
 * but it is NOT the same loop as the scalar one: on a block-boundary hit it RETURNS idx + 1 (:143-148) into ProcessArcRange, which
 * falls through into `for (; e < end; e++)` (:80) with cnt == 0.  `--cnt == 0` (:98) is then never true again, so the scan runs on to
 * `end`, keeps lowering min, and ProcessArcRange returns `end` (:109).  Only a hit in the scalar tail behind the last full group of V
 * arcs (or a range shorter than 2 V, :74) stops at the block boundary and returns e + 1 (:98-103). */
static int opt_range_raw(const int8_t *state, const int64_t *cost, const int32_t *src, const int32_t *tgt, const int64_t *pi,
                         int start, int end, int block_size, int V, int64_t *min, int *cnt, int *best)
{
    int e = start;
    if (V > 0 && end - start >= V * 2) {                                  /* BSPO.cs:74 */
        for (; e <= end - V; e += V) {                                    /* BSPO.cs:119 */
            for (int i = 0; i < V; i++) {                                 /* BSPO.cs:125 */
                int idx = e + i;
                int64_t c = (int64_t)state[idx] * (cost[idx] + pi[src[idx]] - pi[tgt[idx]]);
                if (c < *min) { *min = c; *best = idx; }
                if (--*cnt == 0) {                                        /* BSPO.cs:143-151 */
                    if (*min < 0) { e = idx + 1; goto simd_returned; }    /* return idx + 1 */
                    *cnt = block_size;
                }
            }
        }
simd_returned: ;
    }
    for (; e < end; e++) {                                                /* BSPO.cs:80-107, cnt as the SIMD part left it */
        int64_t c = (int64_t)state[e] * (cost[e] + pi[src[e]] - pi[tgt[e]]);
        if (c < *min) { *min = c; *best = e; }
        if (--*cnt == 0) { if (*min < 0) return e + 1; *cnt = block_size; }
    }
    return e;
}

/* BSPO.cs:39-66 FindEnteringArc: wraps only if the first range ended with min >= 0 (D6); next_arc = what ProcessArcRange returned (D5) */
static int block_opt_raw(const int8_t *state, const int64_t *cost, const int32_t *src, const int32_t *tgt, const int64_t *pi,
                         int m_s, int block_size, int V, int *next_arc, int *in_arc, int64_t *rcost)
{
    int64_t min = 0; int cnt = block_size, best = -1, e;
    e = opt_range_raw(state, cost, src, tgt, pi, *next_arc, m_s, block_size, V, &min, &cnt, &best);
    if (e >= m_s && min >= 0)
        e = opt_range_raw(state, cost, src, tgt, pi, 0, *next_arc, block_size, V, &min, &cnt, &best);
    if (min >= 0) return 0;
    *next_arc = e; *in_arc = best; if (rcost) *rcost = min;
    return 1;
}

static int find_block_opt(ns_oracle *o)
{
    return block_opt_raw(o->state, o->cost, o->src, o->tgt, o->pi, o->search_arc_num, o->block_size, o->vector_width,
                         &o->next_arc, &o->in_arc, NULL);
}

static double now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e9 * (double)ts.tv_sec + (double)ts.tv_nsec;
}

NSO_API int nso_find_entering(ns_oracle *o, int32_t *in_arc)
{
    int found;
    double t0 = o->timing ? now_ns() : 0;       /* NS.cs:285-289 */
    switch (o->rule) {
    case NSO_RULE_BEST:  found = find_best(o); break;
    case NSO_RULE_FIRST: found = find_first(o); break;
    default: found = (o->sem == NSO_SEM_CSHARP_OPT) ? find_block_opt(o) : find_block_plain(o); break;
    }
    if (o->timing) o->t_search += now_ns() - t0;
    if (found && in_arc) *in_arc = o->in_arc;
    return found;
}

/* ------------------------------------------------------------------ one pivot */

/* ns.h:1248-1259, NS.cs:925-941 */
static void find_join(ns_oracle *o)
{
    int u = o->src[o->in_arc], v = o->tgt[o->in_arc];
    while (u != v) {
        if (o->succ_num[u] < o->succ_num[v]) u = o->parent[u]; else v = o->parent[v];
    }
    o->join = u;
}

/* ns.h:1263-1317, NS.cs:943-1010 */
static int find_leaving(ns_oracle *o)
{
    int first, second;
    if (o->state[o->in_arc] == ST_LOWER) { first = o->src[o->in_arc]; second = o->tgt[o->in_arc]; }
    else { first = o->tgt[o->in_arc]; second = o->src[o->in_arc]; }
    o->delta = o->cap[o->in_arc];
    int result = 0; int64_t c, d; int e;
    for (int u = first; u != o->join; u = o->parent[u]) {
        e = o->pred[u]; d = o->flow[e];
        if (o->pred_dir[u] == DIR_DOWN) { c = o->cap[e]; d = c >= o->vmax ? o->vinf : c - d; }
        if (d < o->delta) { o->delta = d; o->u_out = u; result = 1; }
    }
    for (int u = second; u != o->join; u = o->parent[u]) {
        e = o->pred[u]; d = o->flow[e];
        if (o->pred_dir[u] == DIR_UP) { c = o->cap[e]; d = c >= o->vmax ? o->vinf : c - d; }
        if (d <= o->delta) { o->delta = d; o->u_out = u; result = 2; }
    }
    if (result == 1) { o->u_in = first; o->v_in = second; }
    else { o->u_in = second; o->v_in = first; }
    return result != 0;
}

/* ns.h:1320-1340, NS.cs:1012-1040 */
static void change_flow(ns_oracle *o, int change)
{
    if (o->delta > 0) {
        int64_t val = o->state[o->in_arc] * o->delta;
        o->flow[o->in_arc] += val;
        for (int u = o->src[o->in_arc]; u != o->join; u = o->parent[u]) o->flow[o->pred[u]] -= o->pred_dir[u] * val;
        for (int u = o->tgt[o->in_arc]; u != o->join; u = o->parent[u]) o->flow[o->pred[u]] += o->pred_dir[u] * val;
    }
    if (change) {
        o->state[o->in_arc] = ST_TREE;
        o->state[o->pred[o->u_out]] = (o->flow[o->pred[o->u_out]] == 0) ? ST_LOWER : ST_UPPER;
    } else {
        o->state[o->in_arc] = (int8_t)-o->state[o->in_arc];
    }
}

/* ns.h:1343-1466, NS.cs:1042-1183 */
static void update_tree(ns_oracle *o)
{
    int32_t *parent = o->parent, *pred = o->pred, *thread = o->thread, *rev = o->rev_thread;
    int32_t *succ = o->succ_num, *last_succ = o->last_succ; int8_t *pdir = o->pred_dir;
    int u_in = o->u_in, v_in = o->v_in, u_out = o->u_out, in_arc = o->in_arc, join = o->join;
    int old_rev_thread = rev[u_out], old_succ_num = succ[u_out], old_last_succ = last_succ[u_out];
    int v_out = o->v_out = parent[u_out];

    if (u_in == u_out) {
        parent[u_in] = v_in; pred[u_in] = in_arc;
        pdir[u_in] = u_in == o->src[in_arc] ? DIR_UP : DIR_DOWN;
        if (thread[v_in] != u_out) {
            int after = thread[old_last_succ];
            thread[old_rev_thread] = after; rev[after] = old_rev_thread;
            after = thread[v_in];
            thread[v_in] = u_out; rev[u_out] = v_in;
            thread[old_last_succ] = after; rev[after] = old_last_succ;
        }
    } else {
        int thread_continue = old_rev_thread == v_in ? thread[old_last_succ] : thread[v_in];
        int stem = u_in, par_stem = v_in, next_stem, last = last_succ[u_in];
        int before, after = thread[last];
        thread[v_in] = u_in;
        int nd = 0; o->dirty[nd++] = v_in;
        while (stem != u_out) {
            next_stem = parent[stem];
            thread[last] = next_stem; o->dirty[nd++] = last;
            before = rev[stem]; thread[before] = after; rev[after] = before;
            parent[stem] = par_stem; par_stem = stem; stem = next_stem;
            last = last_succ[stem] == last_succ[par_stem] ? rev[par_stem] : last_succ[stem];
            after = thread[last];
        }
        parent[u_out] = par_stem;
        thread[last] = thread_continue; rev[thread_continue] = last;
        last_succ[u_out] = last;
        if (old_rev_thread != v_in) { thread[old_rev_thread] = after; rev[after] = old_rev_thread; }
        for (int i = 0; i < nd; i++) { int u = o->dirty[i]; rev[thread[u]] = u; }
        int tmp_sc = 0, tmp_ls = last_succ[u_out];
        for (int u = u_out, p = parent[u]; u != u_in; u = p, p = parent[u]) {
            pred[u] = pred[p]; pdir[u] = (int8_t)-pdir[p];
            tmp_sc += succ[u] - succ[p]; succ[u] = tmp_sc;
            last_succ[p] = tmp_ls;
        }
        pred[u_in] = in_arc;
        pdir[u_in] = u_in == o->src[in_arc] ? DIR_UP : DIR_DOWN;
        succ[u_in] = old_succ_num;
    }
    int up_limit_out = last_succ[join] == v_in ? join : -1;
    int last_succ_out = last_succ[u_out];
    for (int u = v_in; u != -1 && last_succ[u] == v_in; u = parent[u]) last_succ[u] = last_succ_out;
    if (join != old_rev_thread && v_in != old_rev_thread) {
        for (int u = v_out; u != up_limit_out && last_succ[u] == old_last_succ; u = parent[u]) last_succ[u] = old_rev_thread;
    } else if (last_succ_out != old_last_succ) {
        for (int u = v_out; u != up_limit_out && last_succ[u] == old_last_succ; u = parent[u]) last_succ[u] = last_succ_out;
    }
    for (int u = v_in; u != join; u = parent[u]) succ[u] += old_succ_num;
    for (int u = v_out; u != join; u = parent[u]) succ[u] -= old_succ_num;
}

/* ns.h:1469-1476, NS.cs:1185-1209 */
static void update_potential(ns_oracle *o)
{
    int64_t sigma = o->pi[o->v_in] - o->pi[o->u_in] - o->pred_dir[o->u_in] * o->cost[o->in_arc];
    int end = o->thread[o->last_succ[o->u_in]], k = 0;
    for (int u = o->u_in; u != end; u = o->thread[u]) { o->pi[u] += sigma; k++; }
    o->last_sigma = sigma; o->last_subtree = k;
}

/* body of the loops at ns.h:1598-1607 and NS.cs:319-340.  Returns 0 = continue, 3 = unbounded. */
NSO_API int nso_apply_pivot(ns_oracle *o, int32_t in_arc)
{
    o->in_arc = in_arc;
    find_join(o);
    int change = find_leaving(o);
    if (o->sem == NSO_SEM_LEMON) { if (o->delta >= o->vmax) return NSO_UNBOUNDED; }   /* ns.h:1601 */
    else if (!change && o->delta == 0) return NSO_UNBOUNDED;                          /* NS.cs:321-325 (D7) */
    change_flow(o, change);
    o->last_subtree = 0; o->last_sigma = 0;
    if (change && !o->timing) { update_tree(o); update_potential(o); }
    else if (change) {               /* NS.cs:330-339: tree update and potential update are timed separately */
        double t0 = now_ns();
        update_tree(o);
        double t1 = now_ns();
        update_potential(o);
        o->t_tree += t1 - t0; o->t_pot += now_ns() - t1;
    }
    o->pivots++;
    return 0;
}

/* ------------------------------------------------------------------ LEMON heuristic initial pivots */

/* ns.h:1479-1570.  Returns 0 when it detects unboundedness. */
static int initial_pivots(ns_oracle *o)
{
    int n = o->n, m = o->m;
    int64_t curr, total = 0;
    int32_t *sup = malloc(sizeof(int32_t) * (n + 1)), *dem = malloc(sizeof(int32_t) * (n + 1));
    int ns = 0, nd = 0;
    for (int v = n - 1; v >= 0; v--) {           /* NodeIt order = original id descending */
        curr = o->supply[o->node_id[v]];
        if (curr > 0) { total += curr; sup[ns++] = v; }
        else if (curr < 0) dem[nd++] = v;
    }
    if (o->sum_supply > 0) total -= o->sum_supply;
    if (total <= 0) { free(sup); free(dem); return 1; }

    /* in-/out-arc lists, newest first (list_graph.h:140-152, 206-214) */
    int32_t *in_head = malloc(sizeof(int32_t) * (n + 1)), *in_next = malloc(sizeof(int32_t) * (m + 1));
    int32_t *out_head = malloc(sizeof(int32_t) * (n + 1)), *out_next = malloc(sizeof(int32_t) * (m + 1));
    for (int v = 0; v < n; v++) in_head[v] = out_head[v] = -1;
    for (int e = 0; e < m; e++) {
        in_next[e] = in_head[o->otgt[e]]; in_head[o->otgt[e]] = e;
        out_next[e] = out_head[o->osrc[e]]; out_head[o->osrc[e]] = e;
    }
    int32_t *arcv = malloc(sizeof(int32_t) * (m + n + 1)); int na = 0;
    if (o->sum_supply >= 0) {
        if (ns == 1 && nd == 1) {
            char *reached = calloc(n + 1, 1);
            int32_t *stack = malloc(sizeof(int32_t) * (m + n + 1)); int sp = 0;
            int s = sup[0], t = dem[0];
            reached[t] = 1; stack[sp++] = t;
            while (sp > 0) {
                int v = stack[--sp];
                if (v == s) break;
                for (int a = in_head[v]; a != -1; a = in_next[a]) {
                    int u = o->osrc[a];
                    if (reached[u]) continue;
                    int j = o->arc_id[a];
                    if (o->cap[j] >= total) { arcv[na++] = j; reached[u] = 1; stack[sp++] = u; }
                }
            }
            free(reached); free(stack);
        } else {
            for (int i = 0; i < nd; i++) {
                int v = dem[i]; int64_t min_cost = INT64_MAX; int min_arc = -1;
                for (int a = in_head[v]; a != -1; a = in_next[a]) {
                    int64_t c = o->cost[o->arc_id[a]];
                    if (c < min_cost) { min_cost = c; min_arc = a; }
                }
                if (min_arc != -1) arcv[na++] = o->arc_id[min_arc];
            }
        }
    } else {
        for (int i = 0; i < ns; i++) {
            int u = sup[i]; int64_t min_cost = INT64_MAX; int min_arc = -1;
            for (int a = out_head[u]; a != -1; a = out_next[a]) {
                int64_t c = o->cost[o->arc_id[a]];
                if (c < min_cost) { min_cost = c; min_arc = a; }
            }
            if (min_arc != -1) arcv[na++] = o->arc_id[min_arc];
        }
    }
    int ok = 1;
    for (int i = 0; i < na; i++) {
        int e = arcv[i];
        if (rc(o, e) >= 0) continue;
        if (nso_apply_pivot(o, e) == NSO_UNBOUNDED) { ok = 0; break; }
        o->init_pivots++;
    }
    free(sup); free(dem); free(in_head); free(in_next); free(out_head); free(out_next); free(arcv);
    return ok;
}

/* LEMON only: run the heuristic initial pivots (ns.h:1595).  No-op in the C# modes (D4).
 * Returns 1 ok, 0 unbounded. */
NSO_API int nso_initial_pivots(ns_oracle *o)
{
    if (o->sem != NSO_SEM_LEMON) return 1;
    int ok = initial_pivots(o);
    if (!ok) o->status = NSO_UNBOUNDED;
    return ok;
}

/* ------------------------------------------------------------------ finish */

/* ns.h:1609-1650, NS.cs:359-393 */
NSO_API int nso_finish(ns_oracle *o)
{
    for (int e = o->feas_lo; e < o->feas_hi; e++)
        if (o->flow[e] != 0) { o->status = NSO_INFEASIBLE; return o->status; }
    for (int i = 0; i < o->m; i++) {
        int64_t c = o->lower[i];
        if (c != 0) { o->flow[i] += c; o->supply[o->src[i]] += c; o->supply[o->tgt[i]] -= c; }
    }
    if (o->sem == NSO_SEM_LEMON && o->sum_supply == 0) {               /* ns.h:1628-1648 (D10) */
        if (o->stype == NSO_GEQ) {
            int64_t mx = -INT64_MAX;
            for (int i = 0; i < o->n; i++) if (o->pi[i] > mx) mx = o->pi[i];
            if (mx > 0) for (int i = 0; i < o->n; i++) o->pi[i] -= mx;
        } else {
            int64_t mn = INT64_MAX;
            for (int i = 0; i < o->n; i++) if (o->pi[i] < mn) mn = o->pi[i];
            if (mn < 0) for (int i = 0; i < o->n; i++) o->pi[i] -= mn;
        }
    }
    o->status = NSO_OPTIMAL;
    return o->status;
}

/* Whole solve.  trace (optional) receives the internal index of every entering arc of the MAIN loop
 * (initial pivots excluded), up to trace_cap entries.  Returns the status. */
NSO_API int nso_solve(ns_oracle *o, int32_t *trace, int64_t trace_cap, int64_t *n_pivots)
{
    if (!o->initialized && !nso_init(o)) { if (n_pivots) *n_pivots = 0; return o->status; }
    if (o->status == NSO_INFEASIBLE) { if (n_pivots) *n_pivots = 0; return o->status; }
    if (!nso_initial_pivots(o)) { if (n_pivots) *n_pivots = 0; return o->status; }
    int64_t it = 0; int32_t e;
    while (nso_find_entering(o, &e)) {
        if (trace && it < trace_cap) trace[it] = e;
        it++;
        if (it > o->max_iter) { o->status = NSO_INFEASIBLE; if (n_pivots) *n_pivots = it; return o->status; } /* NS.cs:311-317 */
        if (nso_apply_pivot(o, e) == NSO_UNBOUNDED) { o->status = NSO_UNBOUNDED; if (n_pivots) *n_pivots = it; return o->status; }
    }
    if (n_pivots) *n_pivots = it;
    return nso_finish(o);
}

/* At most max_pivots iterations of the main loop (after nso_init / nso_initial_pivots): used to time a bounded sample
 * of a solve.  Returns 1 when the loop ended by itself (no entering arc / unbounded), 0 when the cap was hit. */
NSO_API int nso_run_pivots(ns_oracle *o, int64_t max_pivots, int64_t *done)
{
    int64_t it = 0; int32_t e; int ended = 0;
    while (it < max_pivots) {
        if (!nso_find_entering(o, &e)) { ended = 1; break; }
        it++;
        if (nso_apply_pivot(o, e) == NSO_UNBOUNDED) { o->status = NSO_UNBOUNDED; ended = 1; break; }
    }
    if (done) *done = it;
    return ended;
}

/* ------------------------------------------------------------------ results / introspection */

NSO_API int nso_status(const ns_oracle *o) { return o->status; }
NSO_API int64_t nso_pivots(const ns_oracle *o) { return o->pivots; }
NSO_API int64_t nso_init_pivot_count(const ns_oracle *o) { return o->init_pivots; }
NSO_API int nso_search_arc_num(const ns_oracle *o) { return o->search_arc_num; }
NSO_API int nso_all_arc_num(const ns_oracle *o) { return o->all_arc_num; }
NSO_API int nso_block_size(const ns_oracle *o) { return o->block_size; }
NSO_API int nso_initial_block_size(const ns_oracle *o) { return o->initial_block_size; }
NSO_API int nso_config_flags(const ns_oracle *o) { return o->cfg_flags; }
NSO_API int nso_would_cache(const ns_oracle *o) { return o->would_cache; }
NSO_API int64_t nso_arcs_checked(const ns_oracle *o) { return o->arcs_checked; }
/* SolverMetrics' three phase buckets (OptimizationTypes.cs:45-52), in microseconds; collected only after nso_enable_timing(o, 1) */
NSO_API void nso_enable_timing(ns_oracle *o, int on) { o->timing = on; }
NSO_API void nso_phase_us(const ns_oracle *o, double out[3]) { out[0] = o->t_search / 1e3; out[1] = o->t_tree / 1e3; out[2] = o->t_pot / 1e3; }
NSO_API int nso_next_arc(const ns_oracle *o) { return o->next_arc; }
NSO_API int64_t nso_art_cost(const ns_oracle *o) { return o->art_cost; }
NSO_API int nso_last_subtree(const ns_oracle *o) { return o->last_subtree; }
NSO_API int64_t nso_last_sigma(const ns_oracle *o) { return o->last_sigma; }

/* ns.h:989-996, NS.cs:452-465 */
NSO_API int64_t nso_total_cost(const ns_oracle *o)
{
    int64_t c = 0;
    for (int i = 0; i < o->m; i++) c += o->flow[i] * o->cost[i];
    return c;
}
/* flow / potential in the caller's numbering */
NSO_API void nso_get_flow(const ns_oracle *o, int64_t *out) { for (int e = 0; e < o->m; e++) out[e] = o->flow[o->arc_id[e]]; }
NSO_API void nso_get_potential(const ns_oracle *o, int64_t *out) { for (int v = 0; v < o->n; v++) out[v] = o->pi[o->node_id[v]]; }
NSO_API void nso_get_arc_id(const ns_oracle *o, int32_t *out) { memcpy(out, o->arc_id, sizeof(int32_t) * o->m); }
/* internal SoA views (what the engine must hold): valid until the next pivot */
NSO_API const int32_t *nso_src(const ns_oracle *o) { return o->src; }
NSO_API const int32_t *nso_tgt(const ns_oracle *o) { return o->tgt; }
NSO_API const int64_t *nso_cost(const ns_oracle *o) { return o->cost; }
NSO_API const int8_t *nso_state(const ns_oracle *o) { return o->state; }
NSO_API const int64_t *nso_pi(const ns_oracle *o) { return o->pi; }
NSO_API const int64_t *nso_flow_internal(const ns_oracle *o) { return o->flow; }
NSO_API const int32_t *nso_thread(const ns_oracle *o) { return o->thread; }
NSO_API const int32_t *nso_parent(const ns_oracle *o) { return o->parent; }

/* ------------------------------------------------------------------ raw scans on caller arrays
 * Same rules as above but on bare arrays, so a test can hand the HIP kernel and the oracle the very
 * same random SoA input without building a tree.  cost/pi are int64; the int32 device mode is
 * checked by feeding values that fit.  Outputs: *arc = entering arc, *rcost = its reduced cost,
 * *next_arc updated in place for the stateful rules.  Returns found (0/1). */
#define RC(e) ((int64_t)state[e] * (cost[e] + pi[src[e]] - pi[tgt[e]]))

NSO_API int nso_scan_best(int m_s, const int8_t *state, const int64_t *cost, const int32_t *src,
                          const int32_t *tgt, const int64_t *pi, int32_t *arc, int64_t *rcost)
{
    int64_t min = 0; int best = -1;
    for (int e = 0; e < m_s; e++) { int64_t c = RC(e); if (c < min) { min = c; best = e; } }
    if (min >= 0) return 0;
    *arc = best; if (rcost) *rcost = min;
    return 1;
}

NSO_API int nso_scan_first(int m_s, const int8_t *state, const int64_t *cost, const int32_t *src,
                           const int32_t *tgt, const int64_t *pi, int32_t *next_arc, int32_t *arc, int64_t *rcost)
{
    for (int e = *next_arc; e < m_s; e++) { int64_t c = RC(e); if (c < 0) { *arc = e; *next_arc = e + 1; if (rcost) *rcost = c; return 1; } }
    for (int e = 0; e < *next_arc; e++) { int64_t c = RC(e); if (c < 0) { *arc = e; *next_arc = e + 1; if (rcost) *rcost = c; return 1; } }
    return 0;
}

NSO_API int nso_scan_block(int m_s, const int8_t *state, const int64_t *cost, const int32_t *src,
                           const int32_t *tgt, const int64_t *pi, int block_size, int optimized, int vector_width,
                           int32_t *next_arc, int32_t *arc, int64_t *rcost)
{
    if (optimized) {
        int na = *next_arc, a = -1;
        const int f = block_opt_raw(state, cost, src, tgt, pi, m_s, block_size, vector_width, &na, &a, rcost);
        if (f) { *next_arc = na; *arc = a; }
        return f;
    }
    int64_t min = 0; int cnt = block_size, e, best = -1;
    for (e = *next_arc; e < m_s; e++) {
        int64_t c = RC(e); if (c < min) { min = c; best = e; }
        if (--cnt == 0) { if (min < 0) goto hit; cnt = block_size; }
    }
    for (e = 0; e < *next_arc; e++) {
        int64_t c = RC(e); if (c < min) { min = c; best = e; }
        if (--cnt == 0) { if (min < 0) goto hit; cnt = block_size; }
    }
    if (min >= 0) return 0;
hit:
    *next_arc = e; *arc = best; if (rcost) *rcost = min;
    return 1;
}
