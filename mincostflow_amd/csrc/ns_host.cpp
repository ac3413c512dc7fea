// ns_host.cpp -- the sequential half of the primal network simplex, kept on the CPU.
//
// Restates the host side of the reference solver (NS.cs = src/MinCostFlow.Core/Lemon/Algorithms/
// NetworkSimplex.cs): problem set-up, transformation to standard form, the artificial-root start
// basis, and per pivot the cycle search, flow augmentation and spanning-tree surgery on the
// thread-index representation.  The two data-parallel pieces -- FindEnteringArc and the potential
// update -- are NOT here: they are calls into the device engine (engine.hip).  There is no CPU
// entering-arc search in this file or anywhere else in the library.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <x86intrin.h>

#include "common.h"

namespace {

inline double ticks() { return (double)__rdtsc(); }     // invariant TSC; scaled to ns once per solve

constexpr int8_t kUp = 1, kDown = -1;   // SpanningTree.cs:67-71 DIR_UP / DIR_DOWN
constexpr int64_t kMax = INT64_MAX;     // NS.cs:126
constexpr int64_t kInf = INT64_MAX / 2; // NS.cs:127

}  // namespace

struct mcf_ns {
    int n = 0, m = 0, root = 0;
    int search_arcs = 0, all_arcs = 0;
    int supply_type = MCF_SUPPLY_GEQ, rule = MCF_RULE_BLOCK_SEARCH;   // NS.cs:38, :77
    bool optimized_pivot = false;                                      // NS.cs:34
    int vector_width = MCF_VECTOR_DEFAULT;                             // Vector<long>.Count of the reference's host (BSPO.cs:74): mcf_ns_set_vector_width
    int device = 0, int_width = 0, block_size = 0, engine_flags = 0;
    bool auto_config = true;                                           // NS.cs:90
    mcf_block_config config{};                                         // _optimizationConfig, NS.cs:89
    // arcs: m + 2n entries (NS.cs:130)
    mcf::hvec<int32_t> tail, head;
    mcf::hvec<int64_t> lower, upper, cost, flow, orig_lower;
    mcf::hvec<int8_t> state;
    // nodes: n + 1 entries, the last one is the artificial root (NS.cs:137,144)
    mcf::hvec<int64_t> supply, pi;
    mcf::hvec<int32_t> par, par_arc, nxt, prv, sub, fin;   // Parent, Pred, Thread, RevThread, SuccNum, LastSucc
    mcf::hvec<int8_t> par_dir;
    std::vector<int32_t> scratch;
    // Node ids in use inside a solve may differ from the caller's: renumber_nodes() relabels the nodes in thread (preorder) order so that the
    // subtree walks, the cycle searches and the engines' per-node tables run through memory front to back instead of chasing pointers.
    // new_of[caller's id] = id in use, orig_of = the inverse; empty = identity.  Arc ids never change, so no pivot rule can tell.
    std::vector<int32_t> new_of, orig_of;
    int64_t walked_since_renumber = 0, jumps_since_renumber = 0, renumbers = 0;
    bool allow_renumber = false;
    bool fused_cycle_search = true;   // find_join + find_leaving in one climb (MCF_NS_FUSED_CYCLE=0: two climbs, as the reference does it)
    bool use_runs = true;             // ... and hand the big ones over as runs of consecutive ids (MCF_NS_RUNS=0: as nodes)
    bool seq_walk = true;             // after the first relabelling the big walks go in runs of consecutive ids (MCF_NS_SEQWALK=0: keep the hinted walk)
    double renumber_every = 128.0;    // relabel when the walks since the last relabelling covered this many times the node count
    double renumber_ticks = 0, renumber_last_ticks = 0, renumber_last_at = 0, renumber_jump_budget = 0;
    int64_t renumber_at_pivot = 0;
    bool renumber_forced = false;     // MCF_NS_RENUMBER set: relabel at that interval whatever it costs (tests)
    int64_t sum_supply = 0, art_cost = 0;
    int status = MCF_NOT_SOLVED;
    bool begun = false, transformed = false, prepared = false, solved = false;
    // the pivot being carried out
    int in_arc = -1, join = -1, u_in = -1, v_in = -1, u_out = -1, v_out = -1;
    int64_t delta = 0;
    int8_t in_state_before = 0;       // State[in_arc] when the pivot started
    bool out_on_tail_path = false;    // the leaving arc lies on the cycle half that starts at the entering arc's tail
    bool change = false;              // the pivot changes the basis (find_leaving found a blocking arc)
    // what the last pivot changed (the engine calls of a host)
    int n_state = 0;
    int32_t st_arc[2] = {0, 0};
    int8_t st_val[2] = {0, 0};
    mcf::hvec<int32_t> follow;      // prefetch hints of shift_potentials
    mcf::hvec<int32_t> moved;       // capacity n+1, the first moved_n entries are valid
    mcf::hvec<int64_t> moved_val;   // their new potentials
    int moved_n = 0;
    bool shift_smaller_side = false;  // inside mcf_ns_solve with 64-bit engines: see shift_potentials
    int reload_min = 0;               // inside mcf_ns_solve: walks of at least this many nodes are announced as a reload of _pi (0: never)
    int reload_min_engines = 0;       // what the engines asked for (mcf_engine_reload_threshold), 0 when one of them only takes lists
    bool moved_as_reload = false;     // the last walk wrote no node list: the engines reload _pi
    bool allow_smaller_side = false;
    // MCF_NS_DEBUG only (reset by every mcf_ns_solve): moved subtrees by log2 of their size -- how many, how many nodes, and where their pivots'
    // time went (ticks): the walk incl. hand-over, the wait for the search that follows, everything else of the pivot
    struct Debug {
        bool on = false;
        int64_t reload_walks = 0, over_half = 0, over_half_nodes = 0;
        int64_t n[32] = {0}, nodes[32] = {0};
        double walk[32] = {0}, wait[32] = {0}, rest[32] = {0};
    } dbg;
    int resident_workgroups = 0;      // mcf_ns_set_device_share: workgroups of this solver's resident grid (0 = the whole device)
    int moved_sent = 0;               // how many of them the engine already has (handed over during the walk)
    bool moved_without_values = false;// the walk wrote no moved_val (run walk: the engines read the bound _pi where they need a value)
    // a big walk after a relabelling writes RUNS of consecutive ids instead of nodes (mcf_engine_shift_potential_runs): run_first / run_len,
    // runs_n of them, runs_sent already handed over
    bool moved_as_runs = false;
    int runs_n = 0, runs_sent = 0;
    mcf::hvec<int32_t> run_first, run_len;
    int engine_rc = 0;                // first error of an engine call made from inside a pivot
    double piece_ticks = 0;           // time inside the hand-over calls made during the walks (part of the potential-update bucket)
    bool hand_over = false;           // a device engine is attached: state writes and potential pieces go to it as they arise
    int64_t sigma = 0;
    // engine(s) + sharding.  kRccl / kHost: one process per GPU, this rank's engine holds one arc shard and the candidates are exchanged
    // by ncclAllGather / through shared memory; kGroup: this process drives every shard itself (peers[] next to engine)
    enum ShardMode { kWhole = 0, kRccl, kHost, kGroup };
    mcf_engine *engine = nullptr;
    std::vector<mcf_engine *> peers;          // kGroup: shards 1 .. R-1 (shard 0 is `engine`)
    std::vector<int32_t> group_devices;
    std::vector<mcf_candidate> cands;
    int shard_mode = kWhole;
    bool sharded = false;                     // kRccl
    uint8_t nccl_id[128];
    std::string exchange_name;
    mcf_exchange *exchange = nullptr;
    int rank = 0, world = 1;
    // trace / metrics
    int32_t *trace = nullptr;
    int64_t trace_cap = 0, trace_len = 0;
    int64_t pivot_limit = 0;          // 0 = none; otherwise Solve() stops after that many pivots with status NotSolved
    mcf_ns_metrics metrics{};
};

namespace {

// ---- the engine calls of a pivot, fanned out to every shard this process drives
int engines_patch_state(mcf_ns *s, int32_t count, const int32_t *arcs, const int8_t *states)
{
    int rc = mcf_engine_patch_state(s->engine, count, arcs, states);       // every engine keeps the writes that fall into its shard
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_patch_state(s->peers[i], count, arcs, states);
    return rc;
}
int engines_append_potential(mcf_ns *s, int32_t count, const int32_t *nodes, const int64_t *values)
{
    int rc = mcf_engine_shift_potential(s->engine, count, nodes, values, s->sigma); // the potentials are replicated on every shard
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_shift_potential(s->peers[i], count, nodes, values, s->sigma);
    return rc;
}
int engines_shift_runs(mcf_ns *s, int32_t n_runs, const int32_t *first, const int32_t *length)
{
    int rc = mcf_engine_shift_potential_runs(s->engine, n_runs, first, length, s->sigma);
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_shift_potential_runs(s->peers[i], n_runs, first, length, s->sigma);
    return rc;
}
int engines_reload_potentials(mcf_ns *s, int32_t changed)
{
    int rc = mcf_engine_reload_potentials(s->engine, changed);
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_reload_potentials(s->peers[i], changed);
    return rc;
}
int engines_search_begin(mcf_ns *s)
{
    int rc = mcf_engine_search_begin(s->engine);     // (RCCL-sharded engines too: their exchange has a stream of its own; a dispatching one launches its scan here)
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_search_begin(s->peers[i]);
    return rc;
}
int engines_search_end(mcf_ns *s, int32_t *found, int32_t *arc)
{
    switch (s->shard_mode) {
    case mcf_ns::kWhole: return mcf_engine_search_end(s->engine, found, arc, nullptr);
    case mcf_ns::kRccl: return mcf_engine_find_entering_sharded(s->engine, found, arc, nullptr);
    case mcf_ns::kHost: {
        mcf_candidate mine;
        int rc = mcf_engine_search_end_local(s->engine, &mine);
        if (!rc) rc = mcf_exchange_all_gather(s->exchange, &mine, s->cands.data());
        if (!rc) rc = mcf_engine_resolve(s->engine, s->world, s->cands.data(), found, arc, nullptr);
        return rc;
    }
    default: {
        int rc = mcf_engine_search_end_local(s->engine, &s->cands[0]);
        for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_search_end_local(s->peers[i], &s->cands[i + 1]);
        // every shard's rule state (next_arc, adaptive block size) advances with the same global answer
        if (!rc) rc = mcf_engine_resolve(s->engine, (int32_t)s->cands.size(), s->cands.data(), found, arc, nullptr);
        for (size_t i = 0; i < s->peers.size() && !rc; ++i) {
            int32_t f2 = 0, a2 = -1;
            rc = mcf_engine_resolve(s->peers[i], (int32_t)s->cands.size(), s->cands.data(), &f2, &a2, nullptr);
        }
        return rc;
    }
    }
}
int engines_renumber(mcf_ns *s, const int32_t *new_of)
{
    int rc = mcf_engine_renumber_nodes(s->engine, new_of);
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) rc = mcf_engine_renumber_nodes(s->peers[i], new_of);
    return rc;
}
void engines_park(mcf_ns *s)
{
    if (s->engine) mcf_engine_park(s->engine);
    for (mcf_engine *e : s->peers) mcf_engine_park(e);
}
void engines_destroy(mcf_ns *s)
{
    if (s->engine) mcf_engine_destroy(s->engine);
    for (mcf_engine *e : s->peers) mcf_engine_destroy(e);
    s->engine = nullptr;
    s->peers.clear();
    if (s->exchange) { mcf_exchange_close(s->exchange); s->exchange = nullptr; }
}

// ---- NS.cs:624-669
bool bounds_ok(const mcf_ns *s)
{
    for (int e = 0; e < s->m; ++e)
        if (s->upper[e] < s->lower[e]) return false;
    return true;
}

void to_standard_form(mcf_ns *s)
{
    for (int e = 0; e < s->m; ++e) {
        const int64_t lo = s->lower[e];
        if (lo == 0) continue;
        s->supply[s->tail[e]] -= lo;
        s->supply[s->head[e]] += lo;
        s->upper[e] -= lo;
        s->lower[e] = 0;
    }
    s->sum_supply = 0;
    for (int v = 0; v < s->n; ++v) s->sum_supply += s->supply[v];
    int64_t biggest = 0;
    for (int e = 0; e < s->m; ++e) biggest = std::max<int64_t>(biggest, s->cost[e] < 0 ? -s->cost[e] : s->cost[e]);
    s->art_cost = (biggest + 1) * (int64_t)s->n;
    s->transformed = true;
}

// ---- NS.cs:671-845: star basis on the artificial root.  GEQ: nodes with supply <= 0 hang on a zero-cost
// root->v arc, the others on an ART_COST v->root arc and get a zero-cost root->v arc at its lower bound; LEQ mirrored.
void start_basis(mcf_ns *s)
{
    const int n = s->n, m = s->m, root = s->root = n;
    s->par[root] = -1; s->par_arc[root] = -1; s->nxt[root] = 0; s->prv[0] = root;
    s->sub[root] = n + 1; s->fin[root] = n - 1; s->par_dir[root] = 0; s->pi[root] = 0;
    for (int e = 0; e < m; ++e) { s->state[e] = MCF_STATE_LOWER; s->flow[e] = 0; }
    s->search_arcs = m + n;
    int extra = m + n;
    for (int v = 0; v < n; ++v) { s->nxt[v] = v + 1 < n ? v + 1 : root; }
    for (int v = 0; v < n; ++v) s->prv[s->nxt[v]] = v;
    const bool geq = s->supply_type == MCF_SUPPLY_GEQ;
    for (int v = 0; v < n; ++v) {
        const int link = m + v;
        s->par[v] = root; s->sub[v] = 1; s->fin[v] = v;
        const bool plain = geq ? s->supply[v] <= 0 : s->supply[v] >= 0;
        // direction of the zero-cost link: GEQ root->v, LEQ v->root
        const int lt = geq ? root : v, lh = geq ? v : root;
        s->tail[link] = lt; s->head[link] = lh; s->upper[link] = kInf; s->cost[link] = 0;
        if (plain) {
            s->par_dir[v] = geq ? kDown : kUp;
            s->pi[v] = 0;
            s->par_arc[v] = link;
            s->flow[link] = geq ? -s->supply[v] : s->supply[v];
            s->state[link] = MCF_STATE_TREE;
        } else {
            s->par_dir[v] = geq ? kUp : kDown;
            s->pi[v] = geq ? -s->art_cost : s->art_cost;
            s->par_arc[v] = extra;
            s->tail[extra] = lh; s->head[extra] = lt;   // the opposite direction
            s->upper[extra] = kInf;
            s->flow[extra] = geq ? s->supply[v] : -s->supply[v];
            s->cost[extra] = s->art_cost;
            s->state[extra] = MCF_STATE_TREE;
            s->flow[link] = 0;
            s->state[link] = MCF_STATE_LOWER;
            ++extra;
        }
    }
    if (n > 0) s->prv[root] = n - 1;
    s->all_arcs = extra;
}

// ---- NS.cs:925-941
void find_join(mcf_ns *s)
{
    int a = s->tail[s->in_arc], b = s->head[s->in_arc];
    while (a != b) {
        if (s->sub[a] < s->sub[b]) a = s->par[a];
        else b = s->par[b];
    }
    s->join = a;
}

// ---- NS.cs:943-1010.  Ties: strict '<' on the first path, '<=' on the second, so the last blocking arc in
// cycle direction leaves (keeps the basis strongly feasible).
bool find_leaving(mcf_ns *s)
{
    int first, second;
    if (s->state[s->in_arc] == MCF_STATE_LOWER) { first = s->tail[s->in_arc]; second = s->head[s->in_arc]; }
    else { first = s->head[s->in_arc]; second = s->tail[s->in_arc]; }
    s->delta = s->upper[s->in_arc];
    int side = 0;
    for (int u = first; u != s->join; u = s->par[u]) {
        const int e = s->par_arc[u];
        int64_t room = s->flow[e];
        if (s->par_dir[u] == kDown) room = s->upper[e] >= kMax ? kInf : s->upper[e] - room;
        if (room < s->delta) { s->delta = room; s->u_out = u; side = 1; }
    }
    for (int u = second; u != s->join; u = s->par[u]) {
        const int e = s->par_arc[u];
        int64_t room = s->flow[e];
        if (s->par_dir[u] == kUp) room = s->upper[e] >= kMax ? kInf : s->upper[e] - room;
        if (room <= s->delta) { s->delta = room; s->u_out = u; side = 2; }
    }
    if (side == 1) { s->u_in = first; s->v_in = second; }
    else { s->u_in = second; s->v_in = first; }
    s->out_on_tail_path = side != 0 && ((side == 1) == (first == s->tail[s->in_arc]));
    return side != 0;
}

// ---- NS.cs:1030-1039, before the flows are touched: the State[] writes of the pivot.  The leaving arc's new state depends on its
// flow after ChangeFlow (0 -> LOWER, else UPPER), which is its flow now -/+ delta along its half of the cycle (same sums as push_flow).
void decide_states(mcf_ns *s, bool change)
{
    s->n_state = 0;
    auto set_state = [&](int arc, int8_t v) {
        s->state[arc] = v;
        s->st_arc[s->n_state] = arc;
        s->st_val[s->n_state] = v;
        s->n_state++;
    };
    const int8_t in_state = s->state[s->in_arc];
    s->in_state_before = in_state;
    if (change) {
        const int out = s->par_arc[s->u_out];
        const int64_t val = in_state * s->delta;
        const int64_t after = s->out_on_tail_path ? s->flow[out] - s->par_dir[s->u_out] * val : s->flow[out] + s->par_dir[s->u_out] * val;
        set_state(s->in_arc, MCF_STATE_TREE);
        set_state(out, after == 0 ? MCF_STATE_LOWER : MCF_STATE_UPPER);
    } else {
        set_state(s->in_arc, (int8_t)-in_state);
    }
}

// ---- NS.cs:1012-1029: the flow change around the cycle (old tree; nothing the device needs)
void push_flow(mcf_ns *s)
{
    if (s->delta > 0) {
        const int64_t val = s->in_state_before * s->delta;
        s->flow[s->in_arc] += val;
        for (int u = s->tail[s->in_arc]; u != s->join; u = s->par[u]) s->flow[s->par_arc[u]] -= s->par_dir[u] * val;
        for (int u = s->head[s->in_arc]; u != s->join; u = s->par[u]) s->flow[s->par_arc[u]] += s->par_dir[u] * val;
    }
}

// ---- NS.cs:1042-1183.  The subtree of u_out is cut off v_out, re-rooted at u_in and hung below v_in; the preorder
// (thread) list is spliced accordingly and SuccNum / LastSucc are repaired along the two root paths.
void rehang_subtree(mcf_ns *s)
{
    auto &par = s->par; auto &parc = s->par_arc; auto &nxt = s->nxt; auto &prv = s->prv;
    auto &sub = s->sub; auto &fin = s->fin; auto &pdir = s->par_dir;
    const int u_in = s->u_in, v_in = s->v_in, u_out = s->u_out, in_arc = s->in_arc, join = s->join;
    const int before_out = prv[u_out], size_out = sub[u_out], fin_out_old = fin[u_out];
    const int v_out = s->v_out = par[u_out];
    const int8_t dir_in = u_in == s->tail[in_arc] ? kUp : kDown;

    if (u_in == u_out) {
        // the whole subtree moves as it is
        par[u_in] = v_in; parc[u_in] = in_arc; pdir[u_in] = dir_in;
        if (nxt[v_in] != u_out) {
            int after = nxt[fin_out_old];
            nxt[before_out] = after; prv[after] = before_out;          // unlink [u_out .. fin_out_old]
            after = nxt[v_in];
            nxt[v_in] = u_out; prv[u_out] = v_in;                       // relink right behind v_in
            nxt[fin_out_old] = after; prv[after] = fin_out_old;
        }
    } else {
        // before_out == v_in also means join == v_out
        const int resume = before_out == v_in ? nxt[fin_out_old] : nxt[v_in];
        int stem = u_in, new_par = v_in, last = fin[u_in], after = nxt[last];
        nxt[v_in] = u_in;
        int n_dirty = 0;
        s->scratch[n_dirty++] = v_in;
        while (stem != u_out) {
            const int up = par[stem];
            nxt[last] = up;                       // the next stem node follows this stem's subtree
            s->scratch[n_dirty++] = last;
            const int before = prv[stem];         // drop the stem's subtree from its old place
            nxt[before] = after; prv[after] = before;
            par[stem] = new_par;
            new_par = stem;
            stem = up;
            last = fin[stem] == fin[new_par] ? prv[new_par] : fin[stem];
            after = nxt[last];
        }
        par[u_out] = new_par;
        nxt[last] = resume; prv[resume] = last;
        fin[u_out] = last;
        if (before_out != v_in) { nxt[before_out] = after; prv[after] = before_out; }
        for (int i = 0; i < n_dirty; ++i) { const int u = s->scratch[i]; prv[nxt[u]] = u; }
        // reverse the parent arcs along the stem, rebuild sizes
        int acc = 0;
        const int fin_new = fin[u_out];
        for (int u = u_out, p = par[u]; u != u_in; u = p, p = par[u]) {
            parc[u] = parc[p];
            pdir[u] = (int8_t)-pdir[p];
            acc += sub[u] - sub[p];
            sub[u] = acc;
            fin[p] = fin_new;
        }
        parc[u_in] = in_arc; pdir[u_in] = dir_in; sub[u_in] = size_out;
    }

    const int stop_out = fin[join] == v_in ? join : -1;
    const int fin_moved = fin[u_out];
    for (int u = v_in; u != -1 && fin[u] == v_in; u = par[u]) fin[u] = fin_moved;
    if (join != before_out && v_in != before_out) {
        for (int u = v_out; u != stop_out && fin[u] == fin_out_old; u = par[u]) fin[u] = before_out;
    } else if (fin_moved != fin_out_old) {
        for (int u = v_out; u != stop_out && fin[u] == fin_out_old; u = par[u]) fin[u] = fin_moved;
    }
    for (int u = v_in; u != join; u = par[u]) sub[u] += size_out;
    for (int u = v_out; u != join; u = par[u]) sub[u] -= size_out;
}

// ---- NS.cs:1185-1209: host copy of pi is kept current (sigma needs pi[v_in], pi[u_in]); the node list is what
// mcf_engine_update_potential ships to the device.
// The subtree of u_in is the thread segment u_in .. LastSucc[u_in], SuccNum[u_in] nodes long; NS.cs:1196-1208 walks it front to back,
// one dependent load per node, and that latency chain is what a big subtree costs (2 % of config 3's pivots move 24 000 nodes on
// average and carry 92 % of all moved nodes).  The same walk here, with two aids that change neither the set of nodes nor their values:
//  * small subtrees are walked from both ends at once (Thread forwards, RevThread backwards): two independent chains;
//  * larger ones use PREFETCH HINTS: follow[v] remembers which node came kWalkAhead steps after v the last time a walk passed v.
//    The thread order of a subtree changes little between pivots (92 % of the hints are still right on config 3), so the walk
//    prefetches the lines it will need kWalkAhead steps from now; a stale hint costs a useless prefetch, nothing else.
// The order of the resulting list is irrelevant to the engine (final values).
constexpr int kWalkAhead = 8, kWalkHintMin = 48;
constexpr double kJumpNs = 24.0;   // what a step of a walk that leaves the id order costs (a mispredicted branch and a miss; measured 23 - 25 ns)
constexpr int kRunsMin = 512;      // walks of this many nodes and more hand over runs of consecutive ids (after a relabelling)
int walk_piece()                   // a big walk hands its nodes to the engine in pieces of this size (2048; mcf_engine_append_potential);
{
    static const int v = [] { int x = 2048; if (const char *u = getenv("MCF_NS_WALK_PIECE")) { const int y = atoi(u); if (y >= 64 && y <= (1 << 20)) x = y; } return x; }();
    return v;
}
                                   // 1024 .. 8192 measure alike on config 3, no hand-over at all costs 0.9 us per pivot

// ---- runs of consecutive ids (after a relabelling in thread order the successor of node a is a + 1 almost everywhere).  Inside a run the
// next address does not depend on the loaded successor (the exit test is a predicted branch, not a data dependency), so the loads of
// consecutive nodes overlap and the hardware prefetcher sees a linear stream; a jump costs one mispredicted branch and a miss: 23-25 ns on the
// hosts measured, which is what such a walk costs -- 2 400 jumps in a walk of 40 000 nodes after some 30 000 pivots.  (Measured and dropped:
// eight successors compared per AVX2 step, run by run or fused with the update -- 2x faster on runs of 100 nodes, 1.4-2x SLOWER on the runs of
// 16 that the solves have: the scalar loop's one branch per node predicts better than a vector loop's exit.)
// the nodes first .. (count of them, in thread order) move by sigma, run by run; nodes (optional) receives them; returns the number of runs
// that ended in a jump.  `next` receives the node behind the last one visited.
int64_t walk_runs(mcf_ns *s, int first, int count, int64_t sigma, int32_t *nodes, int *next)
{
    int64_t *const pi = s->pi.data();
    const int32_t *const nxt = s->nxt.data();
    int64_t jumps = 0;
    int i = 0, a = first;
    if (nodes) {
        while (i < count) {
            int nx;
            do {
                nodes[i] = a;
                pi[a] += sigma;
                nx = nxt[a];
                ++i;
                if (nx != a + 1) { ++jumps; break; }
                ++a;
            } while (i < count);
            a = nx;
        }
    } else {
        while (i < count) {
            int nx;
            do {
                pi[a] += sigma;
                nx = nxt[a];
                ++i;
                if (nx != a + 1) { ++jumps; break; }
                ++a;
            } while (i < count);
            a = nx;
        }
    }
    *next = a;
    return jumps;
}

void shift_potentials(mcf_ns *s)
{
    // runs BEFORE the re-hanging: the nodes that move are the subtree of u_out as it hangs now, and u_in's new parent direction is known
    const int8_t dir_in = s->u_in == s->tail[s->in_arc] ? kUp : kDown;
    s->sigma = s->pi[s->v_in] - s->pi[s->u_in] - dir_in * s->cost[s->in_arc];
    int count = s->sub[s->u_out];
    int first = s->u_out, last = s->fin[s->u_out];
    // (the common offset pi[root] that this builds up is bounded: past 2^60 the complement is never walked again, so it stops growing and every
    // potential stays representable -- the reference's values differ from ours by exactly that offset until normalise_potentials takes it out)
    if (s->shift_smaller_side && (s->pi[s->root] > (1ll << 60) || s->pi[s->root] < -(1ll << 60))) s->shift_smaller_side = false;
    if (s->shift_smaller_side && 2 * (int64_t)count > (int64_t)s->n + 1) {
        // Reduced costs only see differences of potentials: moving the subtree by sigma and moving EVERYTHING ELSE by -sigma give the same
        // search results.  Inside mcf_ns_solve the smaller side is walked -- the rest of the preorder list, from the node after the
        // subtree's last one round to the node before u_out, the root included -- and the common offset (it is pi[root], which the
        // reference keeps at 0) is taken out again before the solve returns (normalise_potentials).
        s->sigma = -s->sigma;
        first = s->nxt[last];
        last = s->prv[s->u_out];
        count = s->n + 1 - count;
    }
    const int64_t sigma = s->sigma;
    s->moved_n = count;
    s->moved_as_reload = false;
    s->moved_without_values = false;
    s->moved_as_runs = false;
    if (s->reload_min > 0 && count >= s->reload_min) {
        // A walk this long is cheaper for the engines as "reload _pi" than as a list (mcf_engine_reload_potentials): nothing is written down,
        // the walk only moves the potentials.  The hints need the node kWalkAhead steps back: a ring of that many.
        int64_t *const pi = s->pi.data();
        const int32_t *const nxt = s->nxt.data();
        if (s->renumbers > 0 && s->seq_walk) {
            int behind = 0;
            s->jumps_since_renumber += walk_runs(s, first, count, sigma, nullptr, &behind);
            s->moved_as_reload = true;
            s->moved_sent = count;
            if (s->dbg.on) s->dbg.reload_walks += 1;
            return;
        }
        int32_t *const follow = s->follow.data();
        int32_t ring[kWalkAhead];
        int a = first;
        for (int i = 0; i < count; ++i) {
            const int h = follow[a];
            __builtin_prefetch(&nxt[h]);
            __builtin_prefetch(&pi[h]);
            __builtin_prefetch(&follow[h]);
            pi[a] += sigma;
            if (i >= kWalkAhead) follow[ring[i & (kWalkAhead - 1)]] = a;
            ring[i & (kWalkAhead - 1)] = a;
            a = nxt[a];
        }
        s->moved_as_reload = true;
        s->moved_sent = count;
        if (s->dbg.on) s->dbg.reload_walks += 1;
        return;
    }
    int32_t *const nodes = s->moved.data();
    int64_t *const vals = s->moved_val.data();
    int64_t *const pi = s->pi.data();
    const int32_t *const nxt = s->nxt.data(), *const prv = s->prv.data();
    if (count < kWalkHintMin) {
        int lo = 0, hi = count - 1;
        int a = first, b = last;
        while (lo < hi) {
            nodes[lo] = a; vals[lo] = (pi[a] += sigma); a = nxt[a]; ++lo;
            nodes[hi] = b; vals[hi] = (pi[b] += sigma); b = prv[b]; --hi;
        }
        if (lo == hi) { nodes[lo] = a; vals[lo] = (pi[a] += sigma); }
        return;
    }
    const int piece = walk_piece();
    if (s->renumbers > 0 && s->seq_walk) {
        // After a relabelling in thread order the successor of node a is a + 1 almost everywhere: walk in RUNS.  Inside a run the next
        // address does not depend on the loaded successor (the exit test is a predicted branch, not a data dependency), so the loads of
        // consecutive nodes overlap and the hardware prefetcher sees a linear stream; a jump costs one unpredicted miss.
        if (s->use_runs && count >= kRunsMin) {
            // RUNS: one {first id, length} pair per run instead of the nodes -- the walk stores nothing per node but the potential itself, and the
            // engines get a list an order of magnitude shorter (the register-resident candidate grid takes the pairs as they are)
            int32_t *const rf = s->run_first.data(), *const rl = s->run_len.data();
            int i = 0, a = first, nr = 0;
            int64_t jumps = 0;
            s->moved_as_runs = true;
            s->runs_sent = 0;
            while (i < count) {
                const int stop = s->hand_over && count - (s->moved_sent + piece) >= piece / 2 ? std::max(i, s->moved_sent + piece) : count;
                while (i < stop) {
                    const int start = a;
                    int len = 0;
                    for (;;) {
                        pi[a] += sigma;
                        const int nx = nxt[a];
                        ++i; ++len;
                        if (nx != a + 1) { a = nx; ++jumps; break; }
                        ++a;
                        if (i >= stop) break;
                    }
                    rf[nr] = start; rl[nr] = len; ++nr;
                }
                if (i < count) {
                    const double tp = ticks();
                    if (!s->engine_rc) s->engine_rc = engines_shift_runs(s, nr - s->runs_sent, rf + s->runs_sent, rl + s->runs_sent);
                    s->piece_ticks += ticks() - tp;
                    s->moved_sent = i;
                    s->runs_sent = nr;
                }
            }
            s->runs_n = nr;
            s->jumps_since_renumber += jumps;
            return;
        }
        // The engines have _pi bound (mcf_engine_bind_potentials): the list goes without values.
        int i = 0, a = first;
        int64_t jumps = 0;
        s->moved_without_values = true;
        while (i < count) {
            const int stop = s->hand_over && count - (s->moved_sent + piece) >= piece / 2 ? std::max(i, s->moved_sent + piece) : count;
            if (stop > i) { jumps += walk_runs(s, a, stop - i, sigma, nodes + i, &a); i = stop; }
            if (i < count) {
                const double tp = ticks();
                if (!s->engine_rc) s->engine_rc = engines_append_potential(s, i - s->moved_sent, nodes + s->moved_sent, nullptr);
                s->piece_ticks += ticks() - tp;
                s->moved_sent = i;
            }
        }
        s->jumps_since_renumber += jumps;
        return;
    }
    int32_t *const follow = s->follow.data();
    int a = first;
    // the first kWalkAhead nodes have no node that far behind them to leave a hint with (count >= kWalkHintMin > kWalkAhead)
    int i = 0;
    for (; i < kWalkAhead; ++i) {
        const int h = follow[a];
        __builtin_prefetch(&nxt[h]);
        __builtin_prefetch(&pi[h]);
        __builtin_prefetch(&follow[h]);
        nodes[i] = a;
        vals[i] = (pi[a] += sigma);
        a = nxt[a];
    }
    while (i < count) {
        // up to the next point where a piece may be handed over (a piece's worth of nodes since the last one, at least half a piece still to come)
        const int stop = s->hand_over && count - (s->moved_sent + piece) >= piece / 2 ? std::max(i, s->moved_sent + piece) : count;
        for (; i < stop; ++i) {
            const int h = follow[a];
            __builtin_prefetch(&nxt[h]);
            __builtin_prefetch(&pi[h]);
            __builtin_prefetch(&follow[h]);
            nodes[i] = a;
            vals[i] = (pi[a] += sigma);
            follow[nodes[i - kWalkAhead]] = a;
            a = nxt[a];
        }
        if (i < count) {
            // the grid applies this piece while the walk goes on (resident mode); the search after the pivot finishes the list
            const double tp = ticks();
            if (!s->engine_rc) s->engine_rc = engines_append_potential(s, i - s->moved_sent, nodes + s->moved_sent, vals + s->moved_sent);
            s->piece_ticks += ticks() - tp;
            s->moved_sent = i;
        }
    }
}

// pi[root] back to 0 (where the reference keeps it): every potential moves by -pi[root], the engines hear of it as one list of all nodes
int normalise_potentials(mcf_ns *s)
{
    const int64_t off = s->pi[s->root];
    if (off == 0) return MCF_OK;
    const int total = s->n + 1;
    if (s->reload_min > 0) {
        for (int u = 0; u < total; ++u) s->pi[u] -= off;
        s->sigma = -off;
        s->moved_n = total;
        s->moved_sent = total;
        return engines_reload_potentials(s, total);
    }
    for (int u = 0; u < total; ++u) { s->moved[u] = u; s->moved_val[u] = (s->pi[u] -= off); }
    s->sigma = -off;
    s->moved_n = total;
    s->moved_sent = 0;
    const int rc = engines_append_potential(s, total, s->moved.data(), s->moved_val.data());
    s->moved_sent = total;
    return rc;
}

// Relabels the nodes so that id order = current thread order (root keeps id n).  Everything indexed by node is permuted, every arc's end
// points are translated; arc ids, states, flows, costs stay where they are.  perm_out (optional) receives new id per CURRENT id.
void renumber_nodes(mcf_ns *s, std::vector<int32_t> *perm_out)
{
    const int n = s->n, N = n + 1;
    std::vector<int32_t> to(N);                      // current id -> new id
    {
        int u = s->nxt[s->root];
        for (int k = 0; k < n; ++k) { to[u] = k; u = s->nxt[u]; }
        to[s->root] = n;
    }
    auto permute = [&](auto &vec) {                 // in place: the engines have the potentials' storage bound (and registered with HIP)
        using E = typename std::remove_reference<decltype(vec)>::type::value_type;
        std::vector<E> tmp((size_t)N);
        for (int u = 0; u < N; ++u) tmp[to[u]] = vec[u];
        std::copy(tmp.begin(), tmp.end(), vec.begin());
    };
    auto translate = [&](auto &vec) { for (int u = 0; u < N; ++u) if (vec[u] >= 0) vec[u] = to[vec[u]]; };
    permute(s->supply); permute(s->pi); permute(s->par); permute(s->par_arc); permute(s->nxt); permute(s->prv); permute(s->sub); permute(s->fin);
    permute(s->par_dir); permute(s->follow);
    translate(s->par); translate(s->nxt); translate(s->prv); translate(s->fin); translate(s->follow);
    const size_t A = s->tail.size();
    for (size_t e = 0; e < A; ++e) { s->tail[e] = to[s->tail[e]]; s->head[e] = to[s->head[e]]; }
    if (s->new_of.empty()) {
        s->new_of = to;
    } else {
        for (int v = 0; v < N; ++v) s->new_of[v] = to[s->new_of[v]];
    }
    s->orig_of.assign(N, 0);
    for (int v = 0; v < N; ++v) s->orig_of[s->new_of[v]] = v;
    // the pivot in progress (none between pivots, but keep the fields meaningful)
    auto tr1 = [&](int &x) { if (x >= 0) x = to[x]; };
    tr1(s->join); tr1(s->u_in); tr1(s->v_in); tr1(s->u_out); tr1(s->v_out);
    s->walked_since_renumber = 0; s->jumps_since_renumber = 0; s->renumbers += 1;
    if (perm_out) perm_out->swap(to);
}

// back to the caller's ids (end of a solve)
void restore_node_ids(mcf_ns *s)
{
    if (s->new_of.empty()) return;
    const int N = s->n + 1;
    const std::vector<int32_t> &back = s->orig_of;      // id in use -> caller's id
    auto permute = [&](auto &vec) {
        using E = typename std::remove_reference<decltype(vec)>::type::value_type;
        std::vector<E> tmp((size_t)N);
        for (int u = 0; u < N; ++u) tmp[back[u]] = vec[u];
        std::copy(tmp.begin(), tmp.end(), vec.begin());
    };
    auto translate = [&](auto &vec) { for (int u = 0; u < N; ++u) if (vec[u] >= 0) vec[u] = back[vec[u]]; };
    permute(s->supply); permute(s->pi); permute(s->par); permute(s->par_arc); permute(s->nxt); permute(s->prv); permute(s->sub); permute(s->fin);
    permute(s->par_dir); permute(s->follow);
    translate(s->par); translate(s->nxt); translate(s->prv); translate(s->fin); translate(s->follow);
    const size_t A = s->tail.size();
    for (size_t e = 0; e < A; ++e) { s->tail[e] = back[s->tail[e]]; s->head[e] = back[s->head[e]]; }
    s->new_of.clear(); s->orig_of.clear();
}

// find_join + find_leaving in ONE climb.  The reference climbs twice: first to the join node (NS.cs:925-941: whichever side has the smaller
// SuccNum steps up), then from both end points of the entering arc to the join again, taking the minimum residual of each path (NS.cs:943-1010).
// Both climbs visit the same nodes in the same bottom-up order per side, and the second one only needs to know where each side stops -- which
// the first one discovers as it goes.  So the residuals are folded into the first climb: a step on the FIRST path (the side the entering arc's
// state makes "first") compares with '<', a step on the second with '<=', exactly as NS.cs:957-997 -- with one difference in ORDER: the
// reference finishes the first path before it starts the second, here the two interleave.  That matters for ties between the paths: the
// reference lets a second-path arc with residual EQUAL to the first path's minimum win (d <= delta), whenever it comes.  Interleaved, each
// side keeps its own minimum and the two are combined at the end with the same rule (second path wins ties), which is the same arc:
//   first-path winner  = the lowest node u on it with residual < everything below it      (strict: the first among equals, bottom-up)
//   second-path winner = the highest node u on it with residual <= everything below it and <= the first path's minimum (the last among equals)
// and the second path's own '<=' chain must be evaluated against min(first-path minimum, running): since min is associative the result
// is: delta = min(d1, d2); leaving = second-path's LAST node with d == d2 if d2 <= d1, else first-path's FIRST node with d == d1.
bool find_join_and_leaving(mcf_ns *s)
{
    const int in_arc = s->in_arc;
    const bool lower = s->state[in_arc] == MCF_STATE_LOWER;
    const int tail = s->tail[in_arc], head = s->head[in_arc];
    // side A climbs from the tail, side B from the head; the FIRST path starts at the tail when the arc is at its lower bound
    int a = tail, b = head;
    const int32_t *const par = s->par.data(), *const sub = s->sub.data(), *const parc = s->par_arc.data();
    const int8_t *const pdir = s->par_dir.data();
    const int64_t *const flow = s->flow.data(), *const upper = s->upper.data();
    const int64_t cap_in = s->upper[in_arc];
    // residual of the tree arc above u when flow is pushed along the cycle: on the first path arcs pointing DOWN gain flow, on the second arcs pointing UP
    int64_t d_first = kMax, d_second = kMax;
    int u_first = -1, u_second = -1;
    const int8_t gain_a = lower ? kDown : kUp;            // the direction whose arcs GAIN flow (residual = upper - flow) on side A ...
    const int8_t gain_b = lower ? kUp : kDown;            // ... and on side B
    while (a != b) {
        if (sub[a] < sub[b]) {
            const int e = parc[a];
            int64_t room = flow[e];
            if (pdir[a] == gain_a) room = upper[e] >= kMax ? kInf : upper[e] - room;
            if (lower) { if (room < d_first) { d_first = room; u_first = a; } }
            else { if (room <= d_second) { d_second = room; u_second = a; } }
            a = par[a];
        } else {
            const int e = parc[b];
            int64_t room = flow[e];
            if (pdir[b] == gain_b) room = upper[e] >= kMax ? kInf : upper[e] - room;
            if (lower) { if (room <= d_second) { d_second = room; u_second = b; } }
            else { if (room < d_first) { d_first = room; u_first = b; } }
            b = par[b];
        }
    }
    s->join = a;
    const int first = lower ? tail : head, second = lower ? head : tail;
    // NS.cs:952: delta starts at the entering arc's capacity; the first path replaces it only with something strictly smaller, the second
    // with anything not larger
    int64_t delta = cap_in;
    int side = 0;
    if (u_first >= 0 && d_first < delta) { delta = d_first; s->u_out = u_first; side = 1; }
    if (u_second >= 0 && d_second <= delta) { delta = d_second; s->u_out = u_second; side = 2; }
    s->delta = delta;
    if (side == 1) { s->u_in = first; s->v_in = second; }
    else { s->u_in = second; s->v_in = first; }
    s->out_on_tail_path = side != 0 && ((side == 1) == (first == tail));
    return side != 0;
}

// One pivot with a given entering arc, in two halves.  pivot_front does what the next search depends on -- the cycle, the State[] writes and
// the potentials of the subtree that is about to move (handed to the engine as they arise) -- and returns true when the problem is
// found unbounded (NS.cs:321-325).  pivot_back does the rest (flows around the cycle, re-hanging the subtree): the solve loop runs it
// while the device is already searching.
bool pivot_front(mcf_ns *s, int arc, double *t_pot)
{
    s->in_arc = arc;
    s->moved_n = 0;
    s->moved_sent = 0;
    s->sigma = 0;
    const bool change = s->change = s->fused_cycle_search ? find_join_and_leaving(s) : (find_join(s), find_leaving(s));
    if (!change && s->delta == 0) return true;
    decide_states(s, change);
    // the engine hears about the state writes before any piece of the potential list (the pieces may start travelling at once)
    if (s->hand_over && !s->engine_rc) s->engine_rc = engines_patch_state(s, s->n_state, s->st_arc, s->st_val);
    if (s->delta == 0) s->metrics.degenerate_pivots++;
    if (change) {
        const double t1 = ticks();
        shift_potentials(s);
        if (t_pot) *t_pot += ticks() - t1;
    }
    return false;
}

void pivot_back(mcf_ns *s, double *t_tree)
{
    push_flow(s);
    if (s->change) {
        const double t0 = ticks();
        rehang_subtree(s);
        if (t_tree) *t_tree += ticks() - t0;
    }
}

bool pivot(mcf_ns *s, int arc, double *t_tree, double *t_pot)
{
    if (pivot_front(s, arc, t_pot)) return true;
    pivot_back(s, t_tree);
    return false;
}

int begin(mcf_ns *s, int32_t *status)
{
    s->status = MCF_NOT_SOLVED;
    s->metrics = mcf_ns_metrics{};
    s->trace_len = 0;
    if (s->begun) return mcf::fail(MCF_ERR_STATE, "Solve() is single-shot: the reference mutates bounds and supplies in place (NS.cs:649, D11); create a new solver");
    s->begun = true;
    if (!bounds_ok(s)) { s->status = MCF_INFEASIBLE; if (status) *status = s->status; return MCF_OK; }   // NS.cs:227-231
    to_standard_form(s);
    start_basis(s);
    if (status) *status = s->status;
    return MCF_OK;
}

void finish(mcf_ns *s)
{
    // NS.cs:1272-1283 with _allArcNum overwritten by _searchArcNum at NS.cs:689 (difference D9): only the n root links
    for (int e = s->m; e < s->search_arcs; ++e)
        if (s->flow[e] != 0) { s->status = MCF_INFEASIBLE; return; }
    s->status = MCF_OPTIMAL;
    for (int e = 0; e < s->m; ++e) {                      // NS.cs:364-388
        const int64_t lo = s->orig_lower[e];
        if (lo == 0) continue;
        s->flow[e] += lo;
        s->supply[s->tail[e]] += lo;
        s->supply[s->head[e]] -= lo;
    }
}

int pick_int_width(const mcf_ns *s)
{
    if (s->int_width == 32 || s->int_width == 64) return s->int_width;
    // |pi| <= ART_COST + n * max|cost| <= 2 * ART_COST; the device forms cost + pi - pi in 64 bits, so each
    // operand just has to fit int32.  ART_COST = (max|cost| + 1) * n (NS.cs:668).
    const __int128 bound = (__int128)4 * s->art_cost;
    return bound < (__int128)INT32_MAX ? 32 : 64;
}

}  // namespace

extern "C" {

int mcf_ns_create(mcf_ns **out, int32_t node_count, int32_t arc_count, const int32_t *source, const int32_t *target)
{
    if (!out) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_create: null argument");
    *out = nullptr;
    if (node_count < 0 || arc_count < 0 || (arc_count && (!source || !target))) return mcf::fail(MCF_ERR_INVALID, "graph must not be null (NS.cs:121)");
    if ((int64_t)arc_count + 2 * (int64_t)node_count > INT32_MAX - 4096) return mcf::fail(MCF_ERR_INVALID, "graph too large for 32-bit arc ids");
    for (int e = 0; e < arc_count; ++e)
        if ((unsigned)source[e] >= (unsigned)node_count || (unsigned)target[e] >= (unsigned)node_count)
            return mcf::fail(MCF_ERR_INVALID, "arc %d: end point out of range", e);
    mcf_ns *s = new mcf_ns();
    s->n = node_count; s->m = arc_count;
    mcf_block_config_default(&s->config);
    const size_t A = (size_t)arc_count + 2 * (size_t)node_count, N = (size_t)node_count + 1;
    s->tail.assign(A, 0); s->head.assign(A, 0);
    std::copy(source, source + arc_count, s->tail.begin());
    std::copy(target, target + arc_count, s->head.begin());
    s->lower.assign(A, 0); s->upper.assign(A, kInf); s->cost.assign(A, 0); s->flow.assign(A, 0);   // NS.cs:614-617
    s->orig_lower.assign(arc_count, 0);
    s->state.assign(A, 0);
    s->supply.assign(N, 0); s->pi.assign(N, 0);
    s->par.assign(N, -1); s->par_arc.assign(N, -1); s->nxt.assign(N, 0); s->prv.assign(N, 0); s->moved.assign(N, 0); s->moved_val.assign(N, 0); s->follow.assign(N, 0);
    s->run_first.assign(N, 0); s->run_len.assign(N, 0);
    s->sub.assign(N, 0); s->fin.assign(N, 0); s->par_dir.assign(N, 0); s->scratch.assign(N + 1, 0);
    *out = s;
    return MCF_OK;
}

void mcf_ns_destroy(mcf_ns *s)
{
    if (!s) return;
    engines_destroy(s);
    delete s;
}

int mcf_ns_set_arc_bounds(mcf_ns *s, int32_t arc, int64_t lower, int64_t upper)
{
    if (!s || arc < 0 || arc >= s->m) return mcf::fail(MCF_ERR_INVALID, "Invalid arc");
    s->lower[arc] = lower;
    s->upper[arc] = upper == MCF_INF_CAP ? kInf : upper;
    s->orig_lower[arc] = lower;
    return MCF_OK;
}
int mcf_ns_set_arc_cost(mcf_ns *s, int32_t arc, int64_t cost)
{
    if (!s || arc < 0 || arc >= s->m) return mcf::fail(MCF_ERR_INVALID, "Invalid arc");
    s->cost[arc] = cost;
    return MCF_OK;
}
int mcf_ns_set_node_supply(mcf_ns *s, int32_t node, int64_t supply)
{
    if (!s || node < 0 || node >= s->n) return mcf::fail(MCF_ERR_INVALID, "Invalid node");
    s->supply[node] = supply;
    return MCF_OK;
}
int mcf_ns_set_problem(mcf_ns *s, const int64_t *lower, const int64_t *upper, const int64_t *cost, const int64_t *supply)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    for (int e = 0; e < s->m; ++e) {
        if (lower) { s->lower[e] = lower[e]; s->orig_lower[e] = lower[e]; }
        if (upper) s->upper[e] = upper[e] == MCF_INF_CAP ? kInf : upper[e];
        if (cost) s->cost[e] = cost[e];
    }
    if (supply) std::copy(supply, supply + s->n, s->supply.begin());
    return MCF_OK;
}
int mcf_ns_set_supply_type(mcf_ns *s, int32_t type)
{
    if (!s || (type != MCF_SUPPLY_GEQ && type != MCF_SUPPLY_LEQ)) return mcf::fail(MCF_ERR_INVALID, "bad supply type");
    s->supply_type = type;
    return MCF_OK;
}
int mcf_ns_set_pivot_rule(mcf_ns *s, int32_t rule)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (rule < 0 || rule > 2) return mcf::fail(MCF_ERR_INVALID, "Pivot rule %d not implemented yet (NS.cs:884)", rule);
    s->rule = rule;
    return MCF_OK;
}
int mcf_ns_enable_optimized_pivot(mcf_ns *s, int32_t enable)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    s->optimized_pivot = enable != 0;
    return MCF_OK;
}
int mcf_ns_set_vector_width(mcf_ns *s, int32_t vector_width)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (vector_width != MCF_VECTOR_DEFAULT && vector_width != MCF_VECTOR_NONE && vector_width != 2 && vector_width != 4 && vector_width != 8)
        return mcf::fail(MCF_ERR_INVALID, "vector_width %d is not Vector<long>.Count of any machine (2, 4, 8, MCF_VECTOR_NONE, or 0 = 4)", vector_width);
    s->vector_width = vector_width;
    return MCF_OK;
}
int mcf_ns_set_optimization_config(mcf_ns *s, const mcf_block_config *config)       // NS.cs:557-561
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (!config) return mcf::fail(MCF_ERR_INVALID, "config must not be null (ArgumentNullException, NS.cs:559)");
    s->config = *config;
    s->auto_config = false;
    return MCF_OK;
}
int mcf_ns_enable_optimizations(mcf_ns *s, int32_t flags)                            // NS.cs:549-552
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    s->config.flags = flags;
    return MCF_OK;
}
int mcf_ns_set_auto_configuration(mcf_ns *s, int32_t enable)                         // NS.cs:567-570
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    s->auto_config = enable != 0;
    return MCF_OK;
}
int mcf_ns_set_device(mcf_ns *s, int32_t device, int32_t int_width, int32_t block_size, int32_t engine_flags)
{
    if (!s || (int_width != 0 && int_width != 32 && int_width != 64) || block_size < 0 || device < 0) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_set_device: bad arguments");
    s->device = device; s->int_width = int_width; s->block_size = block_size; s->engine_flags = engine_flags;
    return MCF_OK;
}
int mcf_ns_set_device_share(mcf_ns *s, int32_t resident_workgroups)
{
    if (!s || resident_workgroups < 0 || resident_workgroups > 256 || (resident_workgroups > 0 && resident_workgroups < 8))
        return mcf::fail(MCF_ERR_INVALID, "mcf_ns_set_device_share: 0 (the whole device) or 8 .. 256 workgroups");
    s->resident_workgroups = resident_workgroups;
    return MCF_OK;
}
int mcf_ns_set_sharding(mcf_ns *s, const uint8_t id[128], int32_t rank, int32_t world)
{
    if (!s || !id || world < 1 || rank < 0 || rank >= world) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_set_sharding: bad arguments");
    memcpy(s->nccl_id, id, 128);
    s->rank = rank; s->world = world; s->sharded = true; s->shard_mode = mcf_ns::kRccl;   // world 1 still goes through the RCCL exchange (tests)
    return MCF_OK;
}
int mcf_ns_set_sharding_host(mcf_ns *s, const char *exchange_name, int32_t rank, int32_t world)
{
    if (!s || !exchange_name || exchange_name[0] != '/' || world < 1 || rank < 0 || rank >= world) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_set_sharding_host: bad arguments");
    s->exchange_name = exchange_name;
    s->rank = rank; s->world = world; s->shard_mode = mcf_ns::kHost; s->sharded = false;
    return MCF_OK;
}
int mcf_ns_set_shard_group(mcf_ns *s, int32_t shards, const int32_t *devices)
{
    if (!s || shards < 1 || shards > 64 || !devices) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_set_shard_group: bad arguments");
    for (int i = 0; i < shards; ++i) if (devices[i] < 0) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_set_shard_group: negative device");
    s->group_devices.assign(devices, devices + shards);
    s->world = shards; s->rank = 0; s->shard_mode = mcf_ns::kGroup; s->sharded = false;
    return MCF_OK;
}
int mcf_ns_set_pivot_limit(mcf_ns *s, int64_t max_pivots)
{
    if (!s || max_pivots < 0) return mcf::fail(MCF_ERR_INVALID, "bad pivot limit");
    s->pivot_limit = max_pivots;
    return MCF_OK;
}
int mcf_ns_set_trace(mcf_ns *s, int32_t *trace, int64_t capacity)
{
    if (!s || capacity < 0) return mcf::fail(MCF_ERR_INVALID, "bad trace buffer");
    s->trace = trace; s->trace_cap = trace ? capacity : 0; s->trace_len = 0;
    return MCF_OK;
}
int mcf_ns_get_trace_length(mcf_ns *s, int64_t *length) { if (!s || !length) return mcf::fail(MCF_ERR_INVALID, "null argument"); *length = s->trace_len; return MCF_OK; }

int mcf_ns_begin(mcf_ns *s, int32_t *status)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    return begin(s, status);
}

int mcf_ns_apply_pivot(mcf_ns *s, int32_t arc, int32_t *unbounded)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (!s->transformed) return mcf::fail(MCF_ERR_STATE, "mcf_ns_begin has not been called (or found the bounds infeasible)");
    if (arc < 0 || arc >= s->search_arcs) return mcf::fail(MCF_ERR_INVALID, "entering arc %d outside the search range [0, %d)", arc, s->search_arcs);
    if (s->state[arc] == MCF_STATE_TREE) return mcf::fail(MCF_ERR_INVALID, "arc %d is in the basis and cannot enter", arc);
    const bool unb = pivot(s, arc, nullptr, nullptr);
    if (unb) s->status = MCF_UNBOUNDED; else s->metrics.iterations++;
    if (unbounded) *unbounded = unb ? 1 : 0;
    return MCF_OK;
}

// Measurement / test aid: `count` pivots with the given entering arcs applied back to back without an engine -- the sequential half alone
// (cycle search, flows, subtree walk, tree surgery), with the walk aids of mcf_ns_solve (smaller side, node renumbering every `renumber_every`
// walked nodes per node of the graph; 0 = never).  Phase times land in the metrics; ids are the caller's again when it returns.
int mcf_ns_replay(mcf_ns *s, const int32_t *arcs, int64_t count, int32_t smaller_side, double renumber_every)
{
    if (!s || count < 0 || (count && !arcs)) return mcf::fail(MCF_ERR_INVALID, "mcf_ns_replay: bad arguments");
    if (!s->transformed) return mcf::fail(MCF_ERR_STATE, "mcf_ns_begin has not been called");
    const double t_start = mcf::now_ns(), tick_start = ticks();
    double t_tree = 0, t_pot = 0, t_renum = 0;
    s->shift_smaller_side = smaller_side != 0;
    s->walked_since_renumber = 0;
    int64_t it = 0, moved = 0;
    for (; it < count; ++it) {
        const int arc = arcs[it];
        if (arc < 0 || arc >= s->search_arcs || s->state[arc] == MCF_STATE_TREE) { s->shift_smaller_side = false; return mcf::fail(MCF_ERR_INVALID, "pivot %lld: arc %d cannot enter", (long long)it, arc); }
        if (pivot(s, arc, &t_tree, &t_pot)) { s->status = MCF_UNBOUNDED; break; }
        moved += s->moved_n;
        s->walked_since_renumber += s->moved_n;
        if (renumber_every > 0 && (double)s->walked_since_renumber > renumber_every * (s->n + 1)) {
            const double t0 = ticks();
            renumber_nodes(s, nullptr);
            t_renum += ticks() - t0;
        }
    }
    s->shift_smaller_side = false;
    const double t0 = ticks();
    restore_node_ids(s);
    {   // pi[root] back to 0 (the smaller-side walks let it drift)
        const int64_t off = s->pi[s->root];
        if (off != 0) for (int u = 0; u <= s->n; ++u) s->pi[u] -= off;
    }
    t_renum += ticks() - t0;
    const double ns_per_tick = (mcf::now_ns() - t_start) / std::max(1.0, ticks() - tick_start);
    s->metrics.iterations += it;
    s->metrics.potential_nodes += moved;
    s->metrics.tree_update_us = t_tree * ns_per_tick / 1e3;
    s->metrics.potential_update_us = t_pot * ns_per_tick / 1e3;
    s->metrics.setup_us = t_renum * ns_per_tick / 1e3;           // here: time spent renumbering
    s->metrics.loop_us = (mcf::now_ns() - t_start) / 1e3;
    s->metrics.reserved = (int32_t)s->renumbers;
    return MCF_OK;
}

int mcf_ns_finish(mcf_ns *s, int32_t *status)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (!s->transformed) return mcf::fail(MCF_ERR_STATE, "mcf_ns_begin has not been called");
    if (s->status != MCF_UNBOUNDED) finish(s);
    if (status) *status = s->status;
    return MCF_OK;
}

int mcf_ns_internal(mcf_ns *s, int32_t *search_arc_num, int32_t *arc_capacity, const int32_t **source, const int32_t **target,
                    const int64_t **cost, const int8_t **state, const int64_t **pi)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (search_arc_num) *search_arc_num = s->search_arcs;
    if (arc_capacity) *arc_capacity = (int32_t)s->tail.size();
    if (source) *source = s->tail.data();
    if (target) *target = s->head.data();
    if (cost) *cost = s->cost.data();
    if (state) *state = s->state.data();
    if (pi) *pi = s->pi.data();
    return MCF_OK;
}

int mcf_ns_tree(mcf_ns *s, const int32_t **parent, const int32_t **pred_arc, const int32_t **succ_num, const int8_t **pred_dir,
                const int64_t **flow, const int64_t **upper)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (!s->transformed) return mcf::fail(MCF_ERR_STATE, "mcf_ns_begin has not been called");
    if (parent) *parent = s->par.data();
    if (pred_arc) *pred_arc = s->par_arc.data();
    if (succ_num) *succ_num = s->sub.data();
    if (pred_dir) *pred_dir = s->par_dir.data();
    if (flow) *flow = s->flow.data();
    if (upper) *upper = s->upper.data();
    return MCF_OK;
}

int mcf_ns_last_pivot(mcf_ns *s, int32_t *n_state, int32_t arcs[2], int8_t states[2], int32_t *n_nodes, const int32_t **nodes, int64_t *sigma)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (n_state) *n_state = s->n_state;
    for (int i = 0; i < s->n_state; ++i) { if (arcs) arcs[i] = s->st_arc[i]; if (states) states[i] = s->st_val[i]; }
    if (n_nodes) *n_nodes = (int32_t)s->moved_n;
    if (nodes) *nodes = s->moved.data();
    if (sigma) *sigma = s->sigma;
    return MCF_OK;
}

// ---- everything of Solve() (NS.cs:215-411) that precedes the pivot loop
int mcf_ns_prepare(mcf_ns *s)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (s->prepared) return MCF_OK;
    if (mcf_device_count() <= s->device)
        return mcf::fail(MCF_ERR_NO_DEVICE, "HIP device %d not available (%d visible); the entering-arc search only exists on the device", s->device, mcf_device_count());
    const double t_start = mcf::now_ns();
    int rc = begin(s, nullptr);
    if (rc) return rc;
    s->prepared = true;
    if (s->status == MCF_INFEASIBLE) return MCF_OK;

    // CreatePivotRuleFinder (NS.cs:847-886): the finder is the device engine
    mcf_engine_desc d{};
    d.node_count = s->n + 1;
    d.arc_capacity = (int32_t)s->tail.size();
    d.search_arc_num = s->search_arcs;
    d.int_width = pick_int_width(s);
    // 32-bit engines check that every potential fits: the common offset of shift_potentials' smaller-side walk could break that, so they walk the subtree
    s->allow_smaller_side = d.int_width == 64 && !(getenv("MCF_NS_SMALLER_SIDE") && getenv("MCF_NS_SMALLER_SIDE")[0] == '0');
    d.rule = s->rule;
    d.semantics = s->optimized_pivot ? MCF_SEM_OPTIMIZED : MCF_SEM_PLAIN;
    d.vector_width = s->vector_width;
    d.block_size = s->block_size;
    d.device = s->device;
    d.flags = s->engine_flags;
    // NS.cs:237-250: the solver configures itself from the problem's shape unless told otherwise.  (The reference analyses after
    // TransformToStandardForm; the analysis only reads the graph, which that step does not touch.)
    if (s->auto_config) { rc = mcf_block_config_auto(&s->config, s->n, s->m, s->tail.data(), s->head.data()); if (rc) return rc; }
    engines_destroy(s);
    const int shards = s->shard_mode == mcf_ns::kGroup ? (int)s->group_devices.size() : 1;
    for (int r = 0; r < shards; ++r) {
        mcf_engine_desc dr = d;
        if (s->shard_mode != mcf_ns::kWhole) {
            rc = mcf_shard_range(s->search_arcs, s->shard_mode == mcf_ns::kGroup ? r : s->rank, s->world, &dr.shard_begin, &dr.shard_end);
            if (rc) return rc;
        }
        // mcf_ns_set_device_share: several independent solvers of one process on one device, each with a grid of that many workgroups
        // (256 / K: every grid gets CUs of its own; an instance whose arcs no longer fit the registers of so few workgroups keeps reduced
        // costs per arc instead, section 3.8 of DESIGN.md)
        if (s->shard_mode == mcf_ns::kWhole && dr.resident_workgroups == 0) {
            dr.resident_workgroups = s->resident_workgroups;
            if (const char *u = getenv("MCF_NS_RESIDENT_WORKGROUPS")) { const int v = atoi(u); if (v >= 8 && v <= 256) dr.resident_workgroups = v; }      // (measurement aid)
        }
        if (s->shard_mode == mcf_ns::kRccl && dr.resident_workgroups == 0) {
            // the collective's kernel needs room beside the resident grid: leave one CU per XCD alone (mcf_engine_comm_init checks)
            const int cus = mcf_device_compute_units(dr.device);
            int leave = 8;
            if (const char *u = getenv("MCF_NS_RCCL_FREE_CUS")) { const int v = atoi(u); if (v >= 0 && v < cus) leave = v; }
            if (cus >= 64) dr.resident_workgroups = cus - leave;
        }
        if (s->shard_mode == mcf_ns::kGroup) {
            dr.device = s->group_devices[r];
            const int sharing = (int)std::count(s->group_devices.begin(), s->group_devices.end(), dr.device);
            if (sharing > 1) {                       // shards rehearsed on one GPU: every grid must fit beside the others
                dr.flags |= MCF_ENGINE_SHARE_DEVICE;
                // workgroups go to the 8 XCDs (32 CUs each) round-robin: an even share per XCD, or one XCD ends up with more 1024-thread
                // workgroups than CUs and a grid never becomes fully resident
                dr.resident_workgroups = std::max(8, 32 / sharing * 8);
                if (sharing > 3) dr.flags |= MCF_ENGINE_DISPATCH;      // more grids than the device has hardware queues for this process (engine.hip: resident slots)
            }
        }
        mcf_engine *e = nullptr;
        rc = mcf_engine_create(&e, &dr);
        if (rc) return rc;
        if (r == 0) s->engine = e; else s->peers.push_back(e);
        rc = mcf_engine_upload(e, s->tail.data(), s->head.data(), s->cost.data(), s->state.data(), s->pi.data());
        if (rc) return rc;
        if (s->shard_mode == mcf_ns::kRccl) { rc = mcf_engine_comm_init(e, s->nccl_id, s->rank, s->world); if (rc) return rc; }
        rc = mcf_engine_set_block_config(e, &s->config, s->n);
        if (rc) return rc;
        rc = mcf_engine_bind_potentials(e, s->pi.data());       // this solver's _pi is current whenever a search begins and still while it is in flight
        if (rc) return rc;
    }
    {
        // every engine of the solver has to agree to reloads (they all follow the same potentials): the largest of their thresholds, or none
        int32_t lo = 0;
        bool all = !(getenv("MCF_NS_RELOAD") && getenv("MCF_NS_RELOAD")[0] == '0');
        for (int r = 0; r < shards && all; ++r) {
            int32_t v = 0;
            rc = mcf_engine_reload_threshold(r == 0 ? s->engine : s->peers[r - 1], &v);
            if (rc) return rc;
            if (v <= 0) all = false;
            lo = std::max(lo, v);
        }
        s->reload_min_engines = all ? std::max<int32_t>(lo, kWalkHintMin) : 0;
    }
    {
        // node relabelling in thread order (renumber_nodes) needs every engine's consent; MCF_NS_RENUMBER=0 switches it off, =x sets the
        // interval in walked nodes per node of the graph
        bool all = !(getenv("MCF_NS_RENUMBER") && getenv("MCF_NS_RENUMBER")[0] == '0' && getenv("MCF_NS_RENUMBER")[1] == 0);
        for (int r = 0; r < shards && all; ++r) {
            int32_t yes = 0;
            rc = mcf_engine_can_renumber(r == 0 ? s->engine : s->peers[r - 1], &yes);
            if (rc) return rc;
            all = yes != 0;
        }
        s->allow_renumber = all;
        s->renumber_every = 128.0;
        if (const char *u = getenv("MCF_NS_RENUMBER")) { const double v = atof(u); if (v > 0) { s->renumber_every = v; s->renumber_forced = true; } }
        s->seq_walk = !(getenv("MCF_NS_SEQWALK") && getenv("MCF_NS_SEQWALK")[0] == '0');
        // runs of consecutive ids instead of node lists: for the grid that takes them as they are (every other engine would expand them again)
        {
            mcf_engine_stats es;
            s->use_runs = mcf_engine_get_stats(s->engine, &es) == MCF_OK && es.shift_grid != 0 && s->peers.empty() && !(getenv("MCF_NS_RUNS") && getenv("MCF_NS_RUNS")[0] == '0');
        }
        s->fused_cycle_search = !(getenv("MCF_NS_FUSED_CYCLE") && getenv("MCF_NS_FUSED_CYCLE")[0] == '0');
    }
    s->cands.assign((size_t)std::max(1, s->world), mcf_candidate{0, 0xFFFFFFFFu, -1, 0, 0xFFFFFFFFu, -1});
    if (s->shard_mode == mcf_ns::kHost) { rc = mcf_exchange_open(&s->exchange, s->exchange_name.c_str(), s->rank, s->world); if (rc) return rc; }
    s->metrics.config_flags = s->config.flags;
    if (!s->optimized_pivot && (s->config.flags & MCF_OPT_REDUCED_COST_CACHING) && s->rule == MCF_RULE_BLOCK_SEARCH) {
        // NS.cs:855-883: `_nodeCount * _nodeCount` is an int product (it wraps) before the conversion to double
        const int32_t nn = (int32_t)((uint32_t)s->n * (uint32_t)s->n);
        const double density = (double)s->search_arcs / (double)nn;
        s->metrics.reference_selects_cached_pivot = (density < 0.01 && s->search_arcs < 10000) ? 1 : 0;
    }
    s->metrics.search_arc_num = s->search_arcs;
    s->metrics.int_width = d.int_width;
    mcf_engine_get_block_size(s->engine, &s->metrics.block_size);
    s->metrics.setup_us = (mcf::now_ns() - t_start) / 1e3;
    return MCF_OK;
}

// ---- Solve(): NS.cs:215-411
int mcf_ns_solve(mcf_ns *s, int32_t *status)
{
    if (!s) return mcf::fail(MCF_ERR_INVALID, "null solver");
    if (s->solved) return mcf::fail(MCF_ERR_STATE, "Solve() is single-shot (NS.cs:649, D11); create a new solver");
    int rc = mcf_ns_prepare(s);
    if (rc) return rc;
    s->solved = true;
    if (s->status == MCF_INFEASIBLE) { if (status) *status = s->status; return MCF_OK; }
    const double t_start = mcf::now_ns();
    const double tick_start = ticks();

    const int64_t max_iter = std::max<int64_t>(1000000, (int64_t)s->n * (int64_t)s->m);   // NS.cs:280
    int64_t it = 0;
    bool limited = false;
    double t_search = 0, t_tree = 0, t_pot = 0, t_hand = 0, t_begin = 0;      // t_hand, t_begin: the last hand-over and the posting of the search, parts of t_pot
    s->hand_over = true;
    s->engine_rc = 0;
    s->piece_ticks = 0;
    s->shift_smaller_side = s->allow_smaller_side;
    s->reload_min = s->reload_min_engines;
    s->walked_since_renumber = 0;
    s->renumber_ticks = 0;
    s->renumber_at_pivot = 0;
    s->dbg = mcf_ns::Debug{};
    s->dbg.on = getenv("MCF_NS_DEBUG") != nullptr;
    int dbg_class = -1;              // size class of the pivot whose search is being waited for
    double dbg_t_prev = 0;
    // The search for pivot k+1 is posted as soon as the device has what it depends on (State[] writes, potentials); the rest of pivot k
    // (flows around the cycle, re-hanging the subtree) runs while the device is searching.  Engines sharded over RCCL search in one
    // blocking call (the all-gather runs on their stream), so for them the two halves simply follow each other.
    rc = engines_search_begin(s);
    while (!rc) {
        const double t0 = ticks();
        int32_t found = 0, arc = -1;
        rc = engines_search_end(s, &found, &arc);
        const double t_got = ticks();
        t_search += t_got - t0;
        if (s->dbg.on && dbg_class >= 0) { s->dbg.wait[dbg_class] += t_got - t0; s->dbg.rest[dbg_class] += (t0 - dbg_t_prev); }
        if (rc || !found) break;
        if (s->trace && it < s->trace_cap) s->trace[it] = arc;
        ++it;
        if (it > max_iter) { s->status = MCF_INFEASIBLE; break; }                          // NS.cs:311-317
        if (s->pivot_limit && it > s->pivot_limit) { --it; limited = true; break; }
        // When to relabel.  The first time: when the walks have covered renumber_every times the node count, or a quarter of n pivots have passed
        // (the cycle searches and the candidate cache's re-evaluation of the moved nodes' arcs chase the same ids).  After that the walks count
        // their JUMPS -- the steps that leave the id order, 23 - 25 ns each, which is what a relabelling buys back -- and the next relabelling
        // comes when the jumps since the last one have cost five times what that one took (2.9 ms with 400 k arcs, 0.3 s with 9 M: the engines
        // rebuild their per-node tables).  Config 5, whole solve: 28.9 s with the 4 relabellings of the first policy (a fixed interval, and never
        // before 25 times the last one's duration), 26.7 with 6, 27.9 with 12; config 3: flat between 6 and 13.  MCF_NS_RENUMBER=x forces an interval.
        if (s->dbg.on && (it % 100000) == 0) fprintf(stderr, "[ns] %lld pivots, %.2f s\n", (long long)it, (mcf::now_ns() - t_start) / 1e9);
        const bool relabel_now = !s->allow_renumber ? false
            : s->renumber_forced ? (double)s->walked_since_renumber > s->renumber_every * (s->n + 1)
            : s->renumbers == 0 ? ((double)s->walked_since_renumber > s->renumber_every * (s->n + 1) || 4 * (it - s->renumber_at_pivot) > (int64_t)s->n)
            : (double)s->jumps_since_renumber > s->renumber_jump_budget;
        if (relabel_now) {
            // no search in flight, nothing of a pivot half done: relabel the nodes in thread order, here and in the engines
            const double tr0 = ticks();
            std::vector<int32_t> perm;
            renumber_nodes(s, &perm);
            rc = engines_renumber(s, perm.data());
            s->renumber_last_ticks = ticks() - tr0;
            s->renumber_last_at = ticks();
            s->renumber_at_pivot = it;
            {   // jumps that cost five times this relabelling (kJumpNs each)
                const double ns_per_tick_now = (mcf::now_ns() - t_start) / std::max(1.0, ticks() - tick_start);
                s->renumber_jump_budget = 5.0 * s->renumber_last_ticks * ns_per_tick_now / kJumpNs;
            }
            s->renumber_ticks += s->renumber_last_ticks;
            if (rc) break;
        }
        const double t_pot_before = t_pot;
        if (pivot_front(s, arc, &t_pot)) { s->status = MCF_UNBOUNDED; break; }
        const double t1 = ticks();
        rc = s->engine_rc;
        if (!rc && s->moved_as_reload) rc = engines_reload_potentials(s, (int32_t)s->moved_n);
        else if (!rc && s->moved_as_runs) { if (s->runs_n > s->runs_sent) rc = engines_shift_runs(s, s->runs_n - s->runs_sent, s->run_first.data() + s->runs_sent, s->run_len.data() + s->runs_sent); }
        else if (!rc && s->moved_n > s->moved_sent)
            rc = engines_append_potential(s, (int32_t)(s->moved_n - s->moved_sent), s->moved.data() + s->moved_sent, s->moved_without_values ? nullptr : s->moved_val.data() + s->moved_sent);
        const double t2 = ticks();
        if (!rc) rc = engines_search_begin(s);
        const double t3 = ticks();
        t_pot += t3 - t1;
        t_hand += t2 - t1;
        t_begin += t3 - t2;
        if (rc) break;                     // the pivot stays half done: the solver is unusable after an engine error, but nothing is left running
        pivot_back(s, &t_tree);
        s->metrics.potential_nodes += (int64_t)s->moved_n;
        s->walked_since_renumber += s->moved_n;
        if (s->dbg.on) {
            const int b = s->moved_n > 0 ? 31 - __builtin_clz((unsigned)s->moved_n) + 1 : 0;      // class 0: nothing moved; class b: 2^(b-1) .. 2^b - 1 nodes
            dbg_class = b;
            s->dbg.n[b] += 1; s->dbg.nodes[b] += s->moved_n;
            s->dbg.walk[b] += (t_pot - t_pot_before);           // walk + hand-over + posting the search
            dbg_t_prev = t_got + (t_pot - t_pot_before);        // what remains until the next wait begins is "rest" (cycle search, flows, tree)
            if (2 * (int64_t)s->moved_n > s->n) { s->dbg.over_half += 1; s->dbg.over_half_nodes += s->moved_n; }
        }
    }
    // ONE way out, error or not: no hand-over pending, no resident grid left spinning, trace length and iteration count filled in
    s->hand_over = false;
    s->shift_smaller_side = false;
    if (!rc) rc = normalise_potentials(s);
    else {
        // after an engine error the engines are not told any more, but the caller's view of _pi is the reference's: pi[root] = 0
        const int64_t off = s->pi[s->root];
        if (off != 0) for (int u = 0; u <= s->n; ++u) s->pi[u] -= off;
    }
    s->reload_min = 0;
    restore_node_ids(s);             // the caller's node ids again (the parked engines keep the relabelled ones: Solve() is single-shot)
    const char *first_error = rc ? mcf_last_error() : nullptr;
    std::string keep_error = first_error ? first_error : "";
    engines_park(s);                 // a resident scan grid must not outlive Solve()
    s->trace_len = std::min(it, s->trace_cap);
    s->metrics.iterations = it;
    if (!rc && s->status == MCF_NOT_SOLVED && !limited) finish(s);
    // the phase buckets were counted in time-stamp-counter ticks (a clock call per phase costs 20+ ns, seven of them per pivot): scale them
    const double ns_per_tick = (mcf::now_ns() - t_start) / std::max(1.0, ticks() - tick_start);
    s->metrics.pivot_search_us = t_search * ns_per_tick / 1e3;
    s->metrics.tree_update_us = t_tree * ns_per_tick / 1e3;
    s->metrics.potential_update_us = t_pot * ns_per_tick / 1e3;
    if (s->dbg.on && it > 1000)
        fprintf(stderr, "[ns] per pivot ns: search wait %.0f | walk %.0f | pieces handed over during walks %.0f | last hand-over %.0f | search begin %.0f | tree %.0f | everything else %.0f\n",
                t_search * ns_per_tick / it, (t_pot - t_hand - t_begin - s->piece_ticks) * ns_per_tick / it, s->piece_ticks * ns_per_tick / it, t_hand * ns_per_tick / it, t_begin * ns_per_tick / it, t_tree * ns_per_tick / it,
                ((ticks() - tick_start) - t_search - t_pot - t_tree) * ns_per_tick / it);
    if (s->dbg.on && it > 1000) {
        fprintf(stderr, "[ns] pivots by size of the moved subtree: nodes | pivots | share of pivots | nodes moved | us per pivot: walk+hand-over, search wait, rest | share of the solve\n");
        const double all_ticks = std::max(1.0, ticks() - tick_start);
        for (int b = 0; b < 32; ++b) {
            if (!s->dbg.n[b]) continue;
            const double k = (double)s->dbg.n[b], us = ns_per_tick / 1e3;
            char label[48];
            if (b == 0) snprintf(label, sizeof(label), "0");
            else if (b == 1) snprintf(label, sizeof(label), "1");
            else snprintf(label, sizeof(label), "%d-%d", 1 << (b - 1), (1 << b) - 1);
            fprintf(stderr, "[ns]   %14s | %8lld | %5.1f %% | %11lld | %7.2f %7.2f %7.2f | %5.1f %%\n", label, (long long)s->dbg.n[b], 100.0 * k / (double)it, (long long)s->dbg.nodes[b],
                    s->dbg.walk[b] * us / k, s->dbg.wait[b] * us / k, s->dbg.rest[b] * us / k, 100.0 * (s->dbg.walk[b] + s->dbg.wait[b] + s->dbg.rest[b]) / all_ticks);
        }
        fprintf(stderr, "[ns] more than half of the %d nodes: %lld pivots / %lld nodes | walks announced as a reload of _pi (%d nodes and more): %lld | nodes relabelled in thread order %lld times, %.1f ms (%.1f %% of the steps since the last time left the id order)\n", s->n, (long long)s->dbg.over_half,
                (long long)s->dbg.over_half_nodes, s->reload_min_engines, (long long)s->dbg.reload_walks, (long long)s->renumbers, s->renumber_ticks * ns_per_tick / 1e6, 100.0 * (double)s->jumps_since_renumber / std::max<double>(1.0, (double)s->walked_since_renumber));
    }
    mcf_engine_get_stats(s->engine, &s->metrics.engine);
    // the rest of SolverMetrics: NS.cs:262-270 (initial block size), :276 (expected iterations), :344-357
    const bool plain_block = s->rule == MCF_RULE_BLOCK_SEARCH && !s->optimized_pivot;
    s->metrics.initial_block_size = plain_block ? s->metrics.engine.initial_block_size : 0;
    s->metrics.final_block_size = plain_block ? s->metrics.engine.current_block_size : 0;
    s->metrics.total_arcs_checked = s->metrics.engine.arcs_checked;
    s->metrics.average_arcs_checked_per_pivot = it > 0 ? (double)s->metrics.total_arcs_checked / (double)it : 0;
    {
        const double expected = std::sqrt((double)s->search_arcs) * s->n * 0.5;
        s->metrics.baseline_iterations = expected < 2147483648.0 ? (int32_t)expected : INT32_MIN;     // what (int) of an oversized double gives on x64
    }
    s->metrics.iteration_ratio = s->metrics.baseline_iterations > 0 ? (double)it / s->metrics.baseline_iterations : 1.0;
    s->metrics.loop_us = (mcf::now_ns() - t_start) / 1e3;
    s->metrics.total_solve_us = s->metrics.loop_us + s->metrics.setup_us;
    if (status) *status = s->status;
    if (rc) { mcf::set_error("%s", keep_error.c_str()); return rc; }
    return MCF_OK;
}

int mcf_ns_status(mcf_ns *s, int32_t *status) { if (!s || !status) return mcf::fail(MCF_ERR_INVALID, "null argument"); *status = s->status; return MCF_OK; }

static int need_optimal(const mcf_ns *s)
{
    if (s->status != MCF_OPTIMAL) return mcf::fail(MCF_ERR_STATE, "Solution not optimal");   // NS.cs:418-421
    return MCF_OK;
}
int mcf_ns_get_flow(mcf_ns *s, int32_t arc, int64_t *flow)
{
    if (!s || !flow) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (int rc = need_optimal(s)) return rc;
    if (arc < 0 || arc >= s->m) return mcf::fail(MCF_ERR_INVALID, "Invalid arc");
    *flow = s->flow[arc];
    return MCF_OK;
}
int mcf_ns_get_potential(mcf_ns *s, int32_t node, int64_t *potential)
{
    if (!s || !potential) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (int rc = need_optimal(s)) return rc;
    if (node < 0 || node >= s->n) return mcf::fail(MCF_ERR_INVALID, "Invalid node");
    *potential = s->pi[node];
    return MCF_OK;
}
int mcf_ns_get_total_cost(mcf_ns *s, int64_t *cost)
{
    if (!s || !cost) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (int rc = need_optimal(s)) return rc;
    int64_t total = 0;
    for (int e = 0; e < s->m; ++e) total += s->flow[e] * s->cost[e];   // NS.cs:459-464
    *cost = total;
    return MCF_OK;
}
int mcf_ns_get_flows(mcf_ns *s, int64_t *out)
{
    if (!s || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (int rc = need_optimal(s)) return rc;
    std::copy(s->flow.begin(), s->flow.begin() + s->m, out);
    return MCF_OK;
}
int mcf_ns_get_potentials(mcf_ns *s, int64_t *out)
{
    if (!s || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (int rc = need_optimal(s)) return rc;
    std::copy(s->pi.begin(), s->pi.begin() + s->n, out);
    return MCF_OK;
}
int mcf_ns_get_arc_upper_bound(mcf_ns *s, int32_t arc, int64_t *upper)
{
    if (!s || !upper) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (arc < 0 || arc >= s->m) return mcf::fail(MCF_ERR_INVALID, "Invalid arc");
    *upper = s->upper[arc];
    return MCF_OK;
}
// SolutionValidator(graph, solver).Validate() (SolutionValidator.cs:20-52) with the checks run on the device
int mcf_ns_validate(mcf_ns *s, mcf_validation *out)
{
    if (!s || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    for (int k = 0; k < MCF_VAL_KINDS; ++k) out->first[k] = -1;
    out->supply_type = s->supply_type;
    if (s->status != MCF_OPTIMAL) {                      // :28-33: nothing else is looked at
        out->errors[MCF_VAL_STATUS] = 1;
        out->first[MCF_VAL_STATUS] = 0;
        return MCF_OK;
    }
    mcf_validator *v = nullptr;
    int rc = mcf_validator_create(&v, s->device, s->n, s->m);
    if (rc) return rc;
    int64_t total = 0;                                   // GetTotalCost(), NS.cs:452-465
    for (int e = 0; e < s->m; ++e) total = (int64_t)((uint64_t)total + (uint64_t)s->flow[e] * (uint64_t)s->cost[e]);
    rc = mcf_validator_upload(v, s->tail.data(), s->head.data(), s->orig_lower.data(), s->upper.data(), s->cost.data(), s->supply.data(),
                              s->flow.data(), s->pi.data());
    if (!rc) rc = mcf_validator_run(v, s->supply_type, total, out);
    mcf_validator_destroy(v);
    return rc;
}
int mcf_ns_check_reduced_costs(mcf_ns *s, int64_t *mismatches)
{
    if (!s || !mismatches) return mcf::fail(MCF_ERR_INVALID, "null argument");
    *mismatches = 0;
    if (!s->engine) return MCF_OK;
    int64_t m = 0;
    int rc = mcf_engine_check_reduced_costs(s->engine, &m, nullptr);
    *mismatches += m;
    for (size_t i = 0; i < s->peers.size() && !rc; ++i) { rc = mcf_engine_check_reduced_costs(s->peers[i], &m, nullptr); *mismatches += m; }
    return rc;
}
int mcf_ns_get_metrics(mcf_ns *s, mcf_ns_metrics *out)
{
    if (!s || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    *out = s->metrics;
    return MCF_OK;
}

}  // extern "C"
