// candidate_cache.hip.h -- host side of the exact candidate cache of the resident engine (included by engine.hip only, inside its anonymous
// namespace, after the resident-grid helpers it posts and collects with).  DESIGN.md section 3.4.
#pragma once

// ------------------------------------------------------------------------------------------------ candidate cache
// Best Eligible picks argmin (c, arc) over ALL search arcs with the CURRENT potentials (NS.cs:1644-1667).  A pivot changes the
// potentials of one subtree and one or two states, so only arcs that touch those nodes change their key; every other arc keeps the key
// the last device search saw.  A device search returns a sorted candidate list that is COMPLETE below a threshold (every eligible arc
// with a smaller key is on it) as of the moment it was posted (its epoch).  The host keeps
//   * the list, and for every node / arc the epoch of its last change: a list entry is CLEAN when nothing of it changed after the list's epoch;
//   * a min-heap with the current keys of all arcs touched since (re-evaluated from the host mirrors when they are touched; at most
//     cand_max_nodes nodes' adjacency per pivot -- a bigger subtree is not evaluated here, the device searches instead).
// Then, exactly:   min over untouched arcs = first clean list entry (all unlisted untouched arcs lie above the threshold)
//                  min over touched arcs   = top of the heap
// and the entering arc is the smaller of the two -- the reference's pivot, arc for arc (every parity test runs with the cache on and off).
// The list is refreshed ASYNCHRONOUSLY: when it runs low the next device search is posted while the host keeps answering from the current
// list; its answer is installed when it has arrived.  The host only waits for the device when it cannot decide: after a big subtree
// moved, or when the list ran out above the threshold.
constexpr int64_t kCandDegreePerNode = 8;       // adjacency entries re-evaluated per pivot at most: this many per node of cand_max_nodes (2048 by default)
constexpr int kCandMaxAvgDegree = 24;           // denser graphs: a single moved node already touches too many arcs

inline const int64_t *cand_pi(const mcf_engine *e) { return e->ext_pi ? e->ext_pi : e->pi.data(); }
inline bool cand_key_less(const mcf_engine::CandKey &a, const mcf_engine::CandKey &b) { return a.c < b.c || (a.c == b.c && a.p < b.p); }
struct CandHeapAfter {        // std::*_heap keep the LARGEST on top: order by "comes later"
    bool operator()(const mcf_engine::HeapEnt &a, const mcf_engine::HeapEnt &b) const { return b.c < a.c || (b.c == a.c && b.p < a.p); }
};

// a potential / state change reported by the caller (the mirrors e->pi / e->h_state already hold the new value)
inline void cand_note_node(mcf_engine *e, int u, bool sigma_known = false, int64_t sigma = 0)
{
    if (e->rc_mode) {            // the RC layout's resident grid takes {node, shift} entries: one per occurrence (shifts add up)
        if (sigma_known) e->rc_sync.push_back(mcf_engine::NodeShift{u, sigma});
        else e->rc_shift_unknown = true;
    }
    if (e->blind_count > 0 && e->blind_epoch == e->cand_now) e->blind_sets = 2;      // may repeat a node of the big list with a newer value: that list's values are read again
    // what cand_decide verified last time stays verified until one of ITS nodes is touched (four compares here instead of six scattered loads there)
    if ((u == e->tp_u) | (u == e->tp_v)) e->tp_ok = false;
    if ((u == e->hd_u) | (u == e->hd_v)) e->hd_ok = false;
    if (e->node_at[u] != e->cand_now) {
        e->node_at[u] = e->cand_now;
        e->sync_nodes.push_back(u);
        if (!e->pivot_overflow) {
            e->pivot_nodes.push_back(u);
            e->pivot_degree += e->adj_start[u + 1] - e->adj_start[u];
            if ((int)e->pivot_nodes.size() > e->cand_max_nodes || e->pivot_degree > kCandDegreePerNode * e->cand_max_nodes) e->pivot_overflow = true;
        }
    }
}
inline void cand_note_arc(mcf_engine *e, int a)
{
    if (a == e->tp_a) e->tp_ok = false;
    if (a == e->hd_a) e->hd_ok = false;
    if (e->arc_at[a] != e->cand_now) { e->arc_at[a] = e->cand_now; e->sync_arcs.push_back(a); e->pivot_arcs.push_back(a); }
}
// Long lists (a big subtree): nothing will be evaluated here -- the device searches next, and a list of its epoch or later makes the
// touched nodes' stamps irrelevant (cand_decide) -- so the list is taken over as it is, without a look at its nodes.
constexpr int kShiftRunsPerLine = 7;        // {first, length} pairs per shift line (the shift grid's range-encoded lines)
constexpr int kShiftRunMax = 256;           // nodes per pair: a longer run comes as several (a thread of the grid sets one pair's bits)

// a big list that came as runs, as node ids after all (for the paths that read ids): the first blind_count entries of pend_node
inline void cand_materialise_blind(mcf_engine *e)
{
    if (!e->blind_lazy) return;
    e->blind_lazy = false;
    std::vector<int32_t> ids;
    ids.reserve(e->blind_count);
    for (size_t r = 0; r + 1 < e->blind_runs.size(); r += 2)
        for (uint32_t k = 0; k < e->blind_runs[r + 1]; ++k) ids.push_back((int32_t)(e->blind_runs[r] + k));
    e->blind_runs.clear();
    e->pend_node.insert(e->pend_node.begin(), ids.begin(), ids.end());
    e->pend_val.insert(e->pend_val.begin(), ids.size(), 0);
    if (!e->shift_grid) { const int64_t *pi = cand_pi(e); for (size_t i = 0; i < ids.size(); ++i) e->pend_val[i] = pi[ids[i]]; }
}

int resident_stop(mcf_engine *e);
// mcf_engine_shift_potential_runs on the shift grid: the runs are the list (one shift for all of it, bound potentials)
inline int cand_note_runs_blind(mcf_engine *e, int32_t n_runs, const int32_t *first, const int32_t *length, int64_t total, bool continuation)
{
    e->pivot_overflow = true;
    if (e->blind_count == 0) { e->blind_epoch = e->cand_now; e->blind_sets = 0; }
    if (!continuation) {
        e->blind_sets += 1;
        if (e->blind_sets > 1 && e->stream_lines > 0) { const int rc = resident_stop(e); if (rc) return rc; }
    }
    if (e->blind_count > 0 && !e->blind_lazy) {
        // ids are there already (another call brought them): one form per list -- these runs become ids too
        for (int r = 0; r < n_runs; ++r) for (int k = 0; k < length[r]; ++k) e->pend_node.push_back(first[r] + k);
        e->pend_val.resize(e->pend_node.size(), 0);
        e->blind_count = e->pend_node.size();
    } else {
        e->blind_lazy = true;
        for (int r = 0; r < n_runs; ++r)
            for (int32_t at = 0; at < length[r]; at += kShiftRunMax) {
                e->blind_runs.push_back((uint32_t)(first[r] + at));
                e->blind_runs.push_back((uint32_t)std::min<int32_t>(kShiftRunMax, length[r] - at));
            }
        e->blind_count += (size_t)total;
    }
    if (e->async_posted && cand_records_ready(e, 0)) { const int rc = cand_collect(e, e->async_at); if (rc) return rc; }
    shift_stream(e);
    return MCF_OK;
}

inline int cand_note_nodes_blind(mcf_engine *e, int32_t count, const int32_t *nodes, const int64_t *values, bool continuation)
{
    if (e->blind_lazy) {
        // runs are there already: lines of them may have travelled in their own format, which the ids that follow cannot continue
        if (e->shift_streamed > 0) { const int rc = resident_stop(e); if (rc) return rc; }
        cand_materialise_blind(e);
    }
    e->pivot_overflow = true;
    if (e->blind_count == 0) { e->blind_epoch = e->cand_now; e->blind_sets = 0; }
    if (!continuation) {
        e->blind_sets += 1;
        // a second list may repeat nodes of the first with other values: whatever of the first has travelled is applied again, in order
        if (e->blind_sets > 1 && e->stream_lines > 0) { const int rc = resident_stop(e); if (rc) return rc; }
    }
    e->pend_node.insert(e->pend_node.end(), nodes, nodes + count);
    if (values) e->pend_val.insert(e->pend_val.end(), values, values + count);
    else {
        // the bound array holds them; the shift grid never reads a big list's values (cand_post_shift sends the shift, or takes the values
        // of that moment from the array itself)
        const size_t at = e->pend_val.size();
        e->pend_val.resize(at + (size_t)count);
        if (!e->shift_grid) for (int i = 0; i < count; ++i) e->pend_val[at + i] = e->ext_pi[nodes[i]];
    }
    e->blind_count = e->pend_node.size();
    // a list refresh that has arrived meanwhile is taken in now, so that this list can start travelling
    if (e->async_posted && cand_records_ready(e, 0)) { const int rc = cand_collect(e, e->async_at); if (rc) return rc; }
    if (e->shift_grid) shift_stream(e);
    else resident_stream(e);
    return MCF_OK;
}

inline bool cand_heap_current(const mcf_engine *e, const mcf_engine::HeapEnt &t)
{
    return ((int)(e->arc_at[t.p] <= t.at) & (int)(e->node_at[t.u] <= t.at) & (int)(e->node_at[t.v] <= t.at)) != 0;      // three independent loads: they miss together
}

inline void cand_push(mcf_engine *e, int a)
{
    const int st = e->h_state[a];
    if (st == 0) return;
    const int64_t *pi = cand_pi(e);
    const int32_t u = e->h_src[a], v = e->h_tgt[a];
    const int64_t d = e->h_cost[a] + pi[u] - pi[v];
    const int64_t rc = st > 0 ? d : -d;
    if (rc >= 0) return;
    e->heap.push_back(mcf_engine::HeapEnt{rc, (uint32_t)a, e->cand_now, u, v});
    std::push_heap(e->heap.begin(), e->heap.end(), CandHeapAfter());
}

// all changes of the pivot are in: bring the heap up to date (or note that it is not)
void cand_absorb_pivot(mcf_engine *e)
{
    if (e->pivot_overflow) {
        e->heap_gap = e->cand_now;
        // arcs next to the moved nodes keep heap entries of older versions: they must not be taken for current ones
        // (entries are only trusted for snapshots taken at or after heap_gap, and those know the arcs' present keys -- see cand_decide)
    } else {
        const int64_t *pi = cand_pi(e);
        const uint32_t now = e->cand_now;
        // first the addresses the evaluation will miss on (version counters and far potentials live in arrays of the graph's size) ...
        for (int u : e->pivot_nodes)
            for (int i = e->adj_start[u], hi = e->adj_start[u + 1]; i < hi; ++i) {
                const mcf_engine::AdjEnt &x = e->adj[i];
                __builtin_prefetch(&pi[x.other & 0x1FFFFFFFu]);
            }
        for (int u : e->pivot_nodes) {
            const int64_t pu = pi[u];
            // everything an evaluation needs sits in the entry except the other end's potential and the arc's version counter; an arc
            // between two moved nodes is evaluated twice (the second entry outdates the first), which is cheaper than remembering it
            for (int i = e->adj_start[u], hi = e->adj_start[u + 1]; i < hi; ++i) {
                const mcf_engine::AdjEnt &x = e->adj[i];
                const int st = (int)((x.other >> 29) & 3u) - 1;
                if (st == 0) continue;
                const int32_t other = (int32_t)(x.other & 0x1FFFFFFFu);
                const int64_t po = pi[other];
                const int64_t d = (x.other >> 31) ? x.cost + po - pu : x.cost + pu - po;
                const int64_t rc = st > 0 ? d : -d;
                if (rc >= 0) continue;
                e->heap.push_back(mcf_engine::HeapEnt{rc, (uint32_t)x.arc, now, u, other});
                std::push_heap(e->heap.begin(), e->heap.end(), CandHeapAfter());
            }
        }
        for (int a : e->pivot_arcs) cand_push(e, a);
    }
    e->pivot_nodes.clear();
    e->pivot_arcs.clear();
    e->pivot_degree = 0;
    e->pivot_overflow = false;
    if (e->heap.size() > e->heap_compact_above) {          // drop what lazy deletion left behind
        size_t keep = 0;
        for (size_t i = 0; i < e->heap.size(); ++i)
            if (cand_heap_current(e, e->heap[i])) e->heap[keep++] = e->heap[i];
        e->heap.resize(keep);
        std::make_heap(e->heap.begin(), e->heap.end(), CandHeapAfter());
        e->st.heap_compactions += 1;
    }
}

// true: *k holds the entering arc (or "none": the scan would find nothing either) without asking the device
bool cand_decide(mcf_engine *e, Key *k)
{
    // the heap knows every change after snap_at only if none of them was skipped (heap_gap) -- and after a gap the entries of the arcs
    // next to the skipped nodes are outdated without being marked so; a snapshot at or after the gap makes all of that irrelevant:
    // an arc touched at or before snap_at is judged by the list (or lies above the threshold), whatever the heap says about it
    if (!e->cand_valid || e->snap_at < e->heap_gap) return false;
    mcf_engine::CandKey best_d{0, kNone};
    while (!e->heap.empty()) {
        const mcf_engine::HeapEnt &t = e->heap.front();
        const uint32_t a = t.p;
        // the entry that was found current last time, and none of its arc's three stamps has been touched since (cand_note_node / _arc)
        if (e->tp_ok && (int32_t)a == e->tp_a && t.at == e->tp_at) { best_d = mcf_engine::CandKey{t.c, a}; break; }
        // current version, and touched after the snapshot (an arc last touched before it is the list's business).  While an entry is
        // current its own epoch is the arc's last touch: a later touch either pushed a newer entry or was a skipped one, i.e. a gap -- and
        // no list older than a gap gets here (above), so both the entry's epoch and the true one are <= snap_at then.
        const bool after = t.at > e->snap_at;
        if (after && cand_heap_current(e, t)) {
            e->tp_ok = true; e->tp_a = (int32_t)a; e->tp_at = t.at; e->tp_u = t.u; e->tp_v = t.v;
            best_d = mcf_engine::CandKey{t.c, a};
            break;
        }
        std::pop_heap(e->heap.begin(), e->heap.end(), CandHeapAfter());
        e->heap.pop_back();
    }
    while (e->cand_ptr < e->cand_list.size()) {
        if (e->hd_ok && e->hd_ptr == e->cand_ptr) break;     // verified clean last time, untouched since
        const uint32_t a = e->cand_list[e->cand_ptr].p;
        // three independent loads (no short circuit: they miss together), and the next entry's lines are asked for meanwhile
        if (e->cand_ptr + 1 < e->cand_list.size()) {
            __builtin_prefetch(&e->arc_at[e->cand_list[e->cand_ptr + 1].p]);
            __builtin_prefetch(&e->node_at[e->cand_ends[2 * e->cand_ptr + 2]]);
            __builtin_prefetch(&e->node_at[e->cand_ends[2 * e->cand_ptr + 3]]);
        }
        const uint32_t t_arc = e->arc_at[a], t_src = e->node_at[e->cand_ends[2 * e->cand_ptr]], t_tgt = e->node_at[e->cand_ends[2 * e->cand_ptr + 1]];
        if ((t_arc <= e->snap_at) & (t_src <= e->snap_at) & (t_tgt <= e->snap_at)) {
            e->hd_ok = true; e->hd_ptr = e->cand_ptr; e->hd_a = (int32_t)a; e->hd_u = e->cand_ends[2 * e->cand_ptr]; e->hd_v = e->cand_ends[2 * e->cand_ptr + 1];
            break;
        }
        e->cand_ptr++;
    }
    mcf_engine::CandKey win{0, kNone};
    if (e->cand_ptr < e->cand_list.size()) {
        win = e->cand_list[e->cand_ptr];
        if (best_d.p != kNone && cand_key_less(best_d, win)) win = best_d;
    } else if (e->cand_thr.p == kNone) {
        win = best_d;                                   // the list was complete: nothing untouched is left
    } else if (best_d.p != kNone && cand_key_less(best_d, e->cand_thr)) {
        win = best_d;                                   // everything untouched and unlisted lies above the threshold
    } else {
        return false;
    }
    k->c = win.p == kNone ? 0 : win.c;
    k->r = 0;
    k->p = win.p;
    return true;
}

// the request for a device search: every node / arc changed since the last request, with their CURRENT values
int cand_build_patches(mcf_engine *e)
{
    // Every entry of a node must carry the same value (the device applies a list in no particular order).  Entries gathered here do (they
    // are read from the mirror now); the big lists do when they are ONE pivot's list of this very epoch (its pieces repeat no node
    // and nothing can have changed since) -- otherwise their values are read again too, and nothing of them may have travelled yet.
    const int64_t *pi = cand_pi(e);
    const size_t n_b = e->blind_count, n_s = e->sync_nodes.size();
    const bool blind_current = n_b == 0 || (e->blind_epoch == e->cand_now && e->blind_sets <= 1);
    // lists may repeat nodes (the update kernels take that: same value every time): squeeze the repeats out when they do not fit
    const bool squeeze = (int64_t)n_b + (int64_t)n_s > e->patch_capacity;
    if ((!blind_current || squeeze) && e->stream_lines > 0) { const int rc = resident_stop(e); if (rc) return rc; }
    if (squeeze) {
        e->pend_node.insert(e->pend_node.end(), e->sync_nodes.begin(), e->sync_nodes.end());
        std::sort(e->pend_node.begin(), e->pend_node.end());
        e->pend_node.erase(std::unique(e->pend_node.begin(), e->pend_node.end()), e->pend_node.end());
        e->pend_val.resize(e->pend_node.size());
        for (size_t i = 0; i < e->pend_node.size(); ++i) e->pend_val[i] = pi[e->pend_node[i]];
    } else {
        if (!blind_current) for (size_t i = 0; i < n_b; ++i) e->pend_val[i] = pi[e->pend_node[i]];
        e->pend_node.resize(n_b + n_s);
        e->pend_val.resize(n_b + n_s);
        for (size_t i = 0; i < n_s; ++i) { e->pend_node[n_b + i] = e->sync_nodes[i]; e->pend_val[n_b + i] = pi[e->sync_nodes[i]]; }
    }
    e->pend_arc.assign(e->sync_arcs.begin(), e->sync_arcs.end());
    e->pend_state.resize(e->pend_arc.size());
    for (size_t i = 0; i < e->pend_arc.size(); ++i) e->pend_state[i] = e->h_state[e->pend_arc[i]];
    e->sync_nodes.clear();
    e->sync_arcs.clear();
    blind_clear(e);
    e->rc_sync.clear();
    e->rc_shift_unknown = false;
    return MCF_OK;
}

bool cand_records_ready(const mcf_engine *e, int g)
{
    const volatile Slot *rec = e->h_slots + (size_t)g * kCandRecords;
    for (int r = 0; r < kCandRecords; ++r) { const int64_t c = rec[r].c; const uint32_t q = rec[r].p; if (rec[r].tag != record_tag(e->seq, c, q)) return false; }
    return true;
}

// wait for the candidate records of request e->seq and install them as the list of epoch `at`
int cand_collect(mcf_engine *e, uint32_t at)
{
    const double t0 = (double)__rdtsc();
    double t0_wall = 0;
    const volatile Slot *slots = e->h_slots;
    e->cand_list.clear();
    e->cand_thr = mcf_engine::CandKey{0, kNone};
    for (int g = 0; g < e->res_grid; ++g) {
        const volatile Slot *rec = slots + (size_t)g * kCandRecords;
        uint64_t spins = 0;
        while (!cand_records_ready(e, g)) {
            _mm_pause();
            if (e->resident_running && (spins & 0xFFF) == 0xFFF && ((const volatile uint32_t *)e->h_exit)[0] != 0) {
                if (((const volatile uint32_t *)e->h_exit)[0] == 4u) {
                    (void)resident_join(e, false);
                    e->resident_running = false;
                    resident_slot_release(e);
                    return mcf::fail(MCF_ERR_TIMEOUT, "the resident grid could not meet at its grid-wide barrier while a list was being applied (are its workgroups all resident?): the device arrays are undefined");
                }
                int rc = resident_restart(e);
                if (rc) return rc;
            }
            if ((++spins & 0xFFFFF) == 0) {
                const hipError_t q = hipStreamQuery(e->res_stream ? e->res_stream : e->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return mcf::fail(MCF_ERR_HIP, "resident grid failed: %s", hipGetErrorString(q));
                if (t0_wall == 0) t0_wall = mcf::now_ns();
                else if (mcf::now_ns() - t0_wall > 20e9) return mcf::fail(MCF_ERR_TIMEOUT, "no answer from the device after 20 s (workgroup %d of %d)", g, e->res_grid);
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int r = 0; r < kCandPerGroup; ++r)
            if (rec[r].p != kNone) e->cand_list.push_back(mcf_engine::CandKey{rec[r].c, rec[r].p});
        if (rec[kCandPerGroup].p != kNone) {
            const mcf_engine::CandKey t{rec[kCandPerGroup].c, rec[kCandPerGroup].p};
            if (e->cand_thr.p == kNone || cand_key_less(t, e->cand_thr)) e->cand_thr = t;
        }
    }
    e->wait_ticks += (double)__rdtsc() - t0;
    if (e->cand_thr.p != kNone) {      // keep only what is provably complete: keys below the smallest unreported key
        size_t keep = 0;
        for (size_t i = 0; i < e->cand_list.size(); ++i)
            if (cand_key_less(e->cand_list[i], e->cand_thr)) e->cand_list[keep++] = e->cand_list[i];
        e->cand_list.resize(keep);
    }
    std::sort(e->cand_list.begin(), e->cand_list.end(), cand_key_less);
    e->cand_ends.resize(2 * e->cand_list.size());
    for (size_t i = 0; i < e->cand_list.size(); ++i) { e->cand_ends[2 * i] = e->h_src[e->cand_list[i].p]; e->cand_ends[2 * i + 1] = e->h_tgt[e->cand_list[i].p]; }
    e->cand_ptr = 0;
    e->cand_valid = true;
    e->hd_ok = e->tp_ok = false;          // another list, another snapshot epoch: nothing verified yet
    e->snap_at = at;
    e->async_posted = false;
    e->tk_collect += (double)__rdtsc() - t0;
    return MCF_OK;
}

// posts a device search that carries everything the device has not heard yet; its list will be of epoch cand_now
// RC layout: the request carries {node, shift} entries -- every occurrence of a node since the last request with the shift of that pivot (the
// pivots' own small lists), plus this pivot's one big list with its common shift when it is short enough -- or, when that is not possible
// (a shift that was not announced, too many entries), the grid is stopped and the values go through update_rc_kernel.
int cand_post_rc(mcf_engine *e)
{
    // a reload of the bound potentials that the grid can carry out itself (cmd 3): the array is read when the request is served, so every
    // potential change noted up to now is part of it -- the lists are dropped, the state writes travel with the request
    const bool reload = e->reload_pi && e->d_ext_pi != nullptr && (int64_t)e->sync_arcs.size() <= e->mailbox_max_st;
    if (reload) {
        e->pend_node.clear(); e->pend_val.clear();
        e->sync_nodes.clear(); e->rc_sync.clear(); e->rc_shift_unknown = false;
        blind_clear(e);
        e->reload_pi = false;
    }
    const size_t n_b = e->blind_count, n_s = e->rc_sync.size();
    const bool blind_ok = n_b == 0 || (e->pend_shift && e->blind_epoch == e->cand_now);      // shifts add up: a node may sit in both lists
    const int64_t n_st = (int64_t)e->sync_arcs.size();
    const bool fast = !e->reload_pi && !e->rc_shift_unknown && blind_ok && (int64_t)(n_b + n_s) <= (int64_t)e->rc_list_max && n_st <= e->mailbox_max_st;
    if (fast) {
        e->pend_node.resize(n_b + n_s);
        e->pend_val.resize(n_b + n_s);
        for (size_t i = 0; i < n_b; ++i) e->pend_val[i] = e->pend_sigma;
        for (size_t i = 0; i < n_s; ++i) { e->pend_node[n_b + i] = e->rc_sync[i].node; e->pend_val[n_b + i] = e->rc_sync[i].shift; }
        e->pend_arc.assign(e->sync_arcs.begin(), e->sync_arcs.end());
        e->pend_state.resize(e->pend_arc.size());
        for (size_t i = 0; i < e->pend_arc.size(); ++i) e->pend_state[i] = e->h_state[e->pend_arc[i]];
        e->sync_nodes.clear(); e->sync_arcs.clear(); e->rc_sync.clear();
        blind_clear(e);
    } else {
        if (int rcb = cand_build_patches(e)) return rcb;        // {node, current value} lists
        int rc = resident_stop(e);
        if (!rc) rc = flush_pending(e);                          // update_rc_kernel: the device works the differences out itself
        if (rc) return rc;
    }
    e->prev_seq = e->seq;
    e->seq += 1;
    if (e->seq == 0) e->seq = 1;
    int rc = resident_start(e, e->prev_seq);
    if (rc) return rc;
    resident_post(e, e->seq, reload ? 3u : 0u, fast);
    if (reload) e->st.rc_reloads_in_grid += 1;
    if (fast && (!e->pend_node.empty() || !e->pend_arc.empty())) e->st.inline_updates += 1;
    e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
    e->posted_at = e->cand_now;
    e->st.arcs_scanned += e->end - e->begin;
    return MCF_OK;
}

// ---- the grid that is patched straight from the request (resident_cand_kernel; kernels.hip.h has the mailbox layout)

// After that grid has left: the arrays in device memory are as the LAST LAUNCH found them.  The host's mirrors are authoritative in candidate
// mode (potentials: the bound array or e->pi; states: h_state), so they are simply written again -- and with that the device has heard
// everything: whatever was waiting to be told is dropped.
int device_sync_from_mirrors(mcf_engine *e)
{
    const int64_t *pi = cand_pi(e);
    HIP_TRY(hipMemcpyAsync(e->d_pi, pi, sizeof(int64_t) * (size_t)e->d.node_count, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_state, e->h_state.data() + e->begin, (size_t)(e->end - e->begin), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
    e->pend_shift = false;
    e->sync_nodes.clear(); e->sync_arcs.clear(); e->rc_sync.clear(); e->rc_shift_unknown = false;
    blind_clear(e); e->blind_sets = 0;
    e->shift_streamed = 0;
    e->reload_pi = false;              // the whole array has just been copied
    e->st.mirror_uploads += 1;
    return MCF_OK;
}

// header (+ entry and shift lines when with_patches) of request `seq`; the value entries are pend_node / pend_val [val_lo, end), the shift
// list pend_node [0, n_shift)
void shift_post_request(mcf_engine *e, uint32_t seq, uint32_t cmd, size_t val_lo, size_t n_shift, int64_t sigma, bool with_patches, bool runs = false)
{
    alignas(16) uint32_t line[16], line1[16];
    const int n_val = with_patches ? (int)(e->pend_node.size() - val_lo) : 0, n_st = with_patches ? (int)e->pend_arc.size() : 0;
    const int extra_val = n_val > 1 ? n_val - 1 : 0, extra_st = n_st > 2 ? n_st - 2 : 0, entries = extra_val + extra_st;
    memset(line1, 0, sizeof(line1));
    for (int l = 0, i = 0; i < entries; ++l) {
        memset(line, 0, sizeof(line));
        for (int k = 0; k < kMailboxPatchesPerLine && i < entries; ++k, ++i) {
            if (i < extra_val) {
                const uint64_t v = (uint64_t)e->pend_val[val_lo + i + 1];
                line[3 * k] = (uint32_t)e->pend_node[val_lo + i + 1];
                line[3 * k + 1] = (uint32_t)v;
                line[3 * k + 2] = (uint32_t)(v >> 32);
            } else {
                const int j = i - extra_val + 2;
                line[3 * k] = (uint32_t)e->pend_arc[j];
                line[3 * k + 1] = (uint32_t)e->pend_state[j];
            }
        }
        line[15] = seq;
        if (l == 0) memcpy(line1, line, sizeof(line));
        else mailbox_write_line(e->mailbox + kMailboxTail + 16 * (size_t)(l - 1), line);
    }
    if (with_patches) {
        // the shift list: n_shift node ids, or (runs) n_shift {first, length} pairs out of blind_runs
        const int per_line = runs ? kShiftRunsPerLine : kShiftNodesPerLine;
        const int total = (int)((n_shift + per_line - 1) / per_line);
        for (int l = e->shift_streamed; l < total; ++l) {
            memset(line, 0, sizeof(line));
            for (int k = 0; k < per_line && (size_t)l * per_line + k < n_shift; ++k) {
                if (runs) { line[2 * k] = e->blind_runs[2 * ((size_t)l * per_line + k)]; line[2 * k + 1] = e->blind_runs[2 * ((size_t)l * per_line + k) + 1]; }
                else line[k] = (uint32_t)e->pend_node[(size_t)l * kShiftNodesPerLine + k];
            }
            line[15] = seq;
            mailbox_write_line(e->mailbox + e->shift_base + 16 * (size_t)l, line);
        }
        e->shift_streamed = 0;
    }
    memset(line, 0, sizeof(line));
    line[0] = seq;
    line[1] = cmd;
    line[2] = (uint32_t)n_val;
    line[3] = with_patches ? (uint32_t)n_shift : 0u;
    line[4] = with_patches && runs ? 1u : 0u;      // the shift list is {first, length} pairs
    line[5] = (uint32_t)n_st;
    for (int k = 0; k < n_st && k < 2; ++k) { line[6 + 2 * k] = (uint32_t)e->pend_arc[k]; line[7 + 2 * k] = (uint32_t)e->pend_state[k]; }
    if (n_val > 0) {
        const uint64_t v = (uint64_t)e->pend_val[val_lo];
        line[10] = (uint32_t)e->pend_node[val_lo];
        line[11] = (uint32_t)v;
        line[12] = (uint32_t)(v >> 32);
    }
    line[13] = (uint32_t)(uint64_t)sigma;
    line[14] = (uint32_t)((uint64_t)sigma >> 32);
    line[15] = seq;
    _mm_sfence();                                  // entry and shift lines leave the write-combining buffers before any header does
    for (int r = 0; r < e->poll_replicas; ++r) {
        uint32_t *unit = e->mailbox + (size_t)r * kReplicaStride;
        if (entries > 0) mailbox_write_line(unit + 16, line1);
        mailbox_write_line(unit, line);
    }
    _mm_sfence();
}
// a request without patches: quit, or a scan request put there again for a grid that has just been started with current arrays
void shift_post(mcf_engine *e, uint32_t seq, uint32_t cmd, bool) { shift_post_request(e, seq, cmd, 0, 0, 0, false); }

// the complete shift lines of this pivot's big list start travelling while the host is still walking the subtree (cmd 2: "shift lines
// 0 .. L-1 of the coming scan request are in place": the grid sets their bits and goes back to polling)
int shift_stream_min_lines()
{
    static const int v = [] { int x = 96; if (const char *u = getenv("MCF_HIP_SHIFT_STREAM_LINES")) { const int y = atoi(u); if (y >= 8 && y <= 65536) x = y; } return x; }();
    return v;
}
void shift_stream(mcf_engine *e)
{
    if (e->async_posted || e->blind_epoch != e->cand_now || e->blind_sets > 1 || !e->pend_shift) return;
    if (!e->resident_running || e->in_flight != mcf_engine::kNoSearch) return;
    const bool runs = e->blind_lazy;
    const int complete = runs ? (int)(e->blind_runs.size() / 2 / kShiftRunsPerLine) : (int)(e->blind_count / kShiftNodesPerLine);
    if (complete - e->shift_streamed < shift_stream_min_lines() || complete > e->max_shift_lines) return;
    uint32_t next_seq = e->seq + 1;
    if (next_seq == 0) next_seq = 1;
    alignas(16) uint32_t line[16];
    for (int l = e->shift_streamed; l < complete; ++l) {
        if (runs) { memcpy(line, e->blind_runs.data() + (size_t)l * 2 * kShiftRunsPerLine, sizeof(uint32_t) * 2 * kShiftRunsPerLine); line[14] = 0u; }
        else for (int k = 0; k < kShiftNodesPerLine; ++k) line[k] = (uint32_t)e->pend_node[(size_t)l * kShiftNodesPerLine + k];
        line[15] = next_seq;
        mailbox_write_line(e->mailbox + e->shift_base + 16 * (size_t)l, line);
    }
    e->stream_sub += 1;
    if (e->stream_sub == 0) e->stream_sub = 1;
    memset(line, 0, sizeof(line));
    line[0] = next_seq;
    line[1] = runs ? 4u : 2u;                      // "shift lines in place": node ids / {first, length} pairs
    line[3] = (uint32_t)complete;
    line[4] = e->stream_sub;
    line[15] = next_seq;
    _mm_sfence();
    for (int r = 0; r < e->poll_replicas; ++r) mailbox_write_line(e->mailbox + (size_t)r * kReplicaStride, line);
    _mm_sfence();
    e->shift_streamed = complete;
    e->stream_lines = complete;                    // "a list is travelling": what the other paths test before they change their mind about it
}

// posts a device search on that grid: this pivot's one big list as a shift list when it qualifies, everything else the device has not heard
// as {node, current value} entries; anything that does not fit the scheme stops the grid, which brings the device up to date wholesale
int cand_post_shift(mcf_engine *e)
{
    if (!e->resident_running) {                     // (re)start first: a start after a stop finds the arrays current and nothing left to tell
        const int rc = resident_start(e, e->seq);
        if (rc) return rc;
    }
    if (e->reload_pi) {
        // mcf_engine_reload_potentials: the grid reads the bound array when it serves the request (cmd 3), so every potential change noted up to
        // now is part of it -- the lists are dropped, the state writes travel with the request.  (A grid that has just been started read the
        // arrays the host wrote from its mirrors: nothing to reload, resident_start cleared the flag.)
        if (!e->d_ext_pi || e->shift_streamed > 0 || (int64_t)e->sync_arcs.size() > (int64_t)e->mailbox_max_st) {
            int rc = resident_stop(e);              // device_sync_from_mirrors: nothing is left to tell
            if (!rc) rc = resident_start(e, e->seq);
            if (rc) return rc;
        }
    }
    if (e->reload_pi) {
        e->pend_node.clear(); e->pend_val.clear();
        e->sync_nodes.clear();
        blind_clear(e);
        e->pend_arc.assign(e->sync_arcs.begin(), e->sync_arcs.end());
        e->pend_state.resize(e->pend_arc.size());
        for (size_t i = 0; i < e->pend_arc.size(); ++i) e->pend_state[i] = e->h_state[e->pend_arc[i]];
        e->prev_seq = e->seq;
        e->seq += 1;
        if (e->seq == 0) e->seq = 1;
        shift_post_request(e, e->seq, 3u, 0, 0, 0, true);
        e->st.rc_reloads_in_grid += 1;
        e->pend_arc.clear(); e->pend_state.clear();
        e->sync_arcs.clear();
        e->stream_lines = 0;
        e->reload_pi = false;
        e->posted_at = e->cand_now;
        e->st.arcs_scanned += e->end - e->begin;
        return MCF_OK;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        const size_t n_b = e->blind_count, n_s = e->sync_nodes.size();
        const bool blind_current = n_b == 0 || (e->blind_epoch == e->cand_now && e->blind_sets <= 1);
        const size_t n_pairs = e->blind_lazy ? e->blind_runs.size() / 2 : 0;
        const bool as_shift = n_b > 0 && blind_current && e->pend_shift &&
                              (e->blind_lazy ? n_pairs <= (size_t)e->max_shift_lines * kShiftRunsPerLine : n_b <= (size_t)e->max_shift_lines * kShiftNodesPerLine);
        const bool fits = (int64_t)(as_shift ? n_s : n_b + n_s) <= (int64_t)e->patch_capacity && (int64_t)e->sync_arcs.size() <= (int64_t)e->mailbox_max_st;
        if ((!as_shift && e->shift_streamed > 0) || !fits) {
            // lines of a list that is no shift list any more have travelled, or the request would not fit: bring the device up to date wholesale
            int rc = resident_stop(e);              // device_sync_from_mirrors: nothing is left to tell
            if (!rc) rc = resident_start(e, e->seq);
            if (rc) return rc;
            continue;
        }
        const int64_t *pi = cand_pi(e);
        const bool runs = as_shift && e->blind_lazy;          // the pairs travel as they are; pend_node then holds the value entries only
        if (!as_shift) cand_materialise_blind(e);              // ... as ids otherwise
        const size_t base = runs ? 0 : n_b, val_lo = as_shift ? base : 0;
        if (!as_shift) for (size_t i = 0; i < n_b; ++i) e->pend_val[i] = pi[e->pend_node[i]];      // current values (a node named twice carries the same one)
        e->pend_node.resize(base + n_s);
        e->pend_val.resize(base + n_s);
        for (size_t i = 0; i < n_s; ++i) { e->pend_node[base + i] = e->sync_nodes[i]; e->pend_val[base + i] = pi[e->sync_nodes[i]]; }
        e->pend_arc.assign(e->sync_arcs.begin(), e->sync_arcs.end());
        e->pend_state.resize(e->pend_arc.size());
        for (size_t i = 0; i < e->pend_arc.size(); ++i) e->pend_state[i] = e->h_state[e->pend_arc[i]];
        e->prev_seq = e->seq;
        e->seq += 1;
        if (e->seq == 0) e->seq = 1;
        shift_post_request(e, e->seq, 0u, val_lo, as_shift ? (runs ? n_pairs : n_b) : 0, e->pend_sigma, true, runs);
        if (!e->pend_node.empty() || !e->pend_arc.empty()) e->st.inline_updates += 1;
        if (as_shift) e->st.shift_lists += 1;
        e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
        e->sync_nodes.clear(); e->sync_arcs.clear();
        blind_clear(e);
        e->stream_lines = 0;
        e->posted_at = e->cand_now;
        e->st.arcs_scanned += e->end - e->begin;
        return MCF_OK;
    }
    return mcf::fail(MCF_ERR_STATE, "cand_post_shift: the request does not fit an empty mailbox");
}

int cand_post(mcf_engine *e)
{
    if (e->shift_grid) return cand_post_shift(e);
    if (e->rc_mode) return cand_post_rc(e);
    if (int rcb = cand_build_patches(e)) return rcb;
    if ((int)e->pend_arc.size() > e->mailbox_max_st) {      // hundreds of pivots' worth of state writes: cannot happen between two requests, kept for safety
        int rc = resident_stop(e);
        if (!rc) rc = flush_pending(e);
        if (rc) return rc;
    }
    e->prev_seq = e->seq;
    e->seq += 1;
    if (e->seq == 0) e->seq = 1;
    int rc = resident_start(e, e->prev_seq);
    if (rc) return rc;
    resident_post(e, e->seq, 0u, true);
    if (!e->pend_node.empty() || !e->pend_arc.empty()) e->st.inline_updates += 1;
    e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
    e->posted_at = e->cand_now;
    e->st.arcs_scanned += e->end - e->begin;
    return MCF_OK;
}

// the host mirrors' arc lists per node, with what a re-evaluation needs next to each other (upload, mcf_engine_patch_arcs)
void cand_build_adjacency(mcf_engine *e)
{
    const int n = e->d.node_count, m_s = e->d.search_arc_num;
    const int32_t *source = e->h_src.data(), *target = e->h_tgt.data();
    // an arc shard keeps its own cache: the list, the heap and this adjacency cover the arcs [begin, end) it holds, the answer is the shard's
    // candidate (mcf_engine_search_end_local) and the holders' MINLOC does the rest
    const int lo = e->begin, hi = std::min(e->end, m_s);
    e->adj_start.assign(n + 1, 0);
    for (int a = lo; a < hi; ++a) { e->adj_start[source[a] + 1]++; if (target[a] != source[a]) e->adj_start[target[a] + 1]++; }
    for (int u = 0; u < n; ++u) e->adj_start[u + 1] += e->adj_start[u];
    e->adj.assign(e->adj_start[n], mcf_engine::AdjEnt{0, 0u, 0});
    e->adj_pos.assign((size_t)2 * m_s, -1);
    std::vector<int32_t> fill(e->adj_start.begin(), e->adj_start.end() - 1);
    for (int a = lo; a < hi; ++a) {
        const uint32_t st_bits = (uint32_t)(e->h_state[a] + 1) << 29;
        e->adj_pos[2 * (size_t)a] = fill[source[a]];
        e->adj[fill[source[a]]++] = mcf_engine::AdjEnt{a, (uint32_t)target[a] | st_bits, e->h_cost[a]};
        if (target[a] != source[a]) {
            e->adj_pos[2 * (size_t)a + 1] = fill[target[a]];
            e->adj[fill[target[a]]++] = mcf_engine::AdjEnt{a, (uint32_t)source[a] | st_bits | 0x80000000u, e->h_cost[a]};
        }
    }
}

// forgets the list and the heap (upload, or the device state was changed behind the cache's back)
void cand_reset(mcf_engine *e)
{
    e->cand_valid = false;
    e->cand_list.clear();
    e->cand_ptr = 0;
    e->hd_ok = e->tp_ok = false;
    e->heap.clear();
    e->pivot_nodes.clear(); e->pivot_arcs.clear(); e->pivot_degree = 0; e->pivot_overflow = false;
    e->heap_gap = e->cand_now;
}

