// engine.hip -- device side of the network-simplex pivot seam for MI355X (gfx950).
//
// What runs here:
//   a1/a2/a3  entering-arc search: one coalesced pass over the SoA arc arrays computing
//             c = state[e] * (cost[e] + pi[source[e]] - pi[target[e]])      (NS.cs:1351-1352)
//             and an exact, tie-break preserving argmin (wave shuffle -> LDS -> 16-byte record per
//             workgroup written straight into pinned host memory; the host merges the records).
//   a4        potential update pi[u] += sigma over the moved subtree       (NS.cs:1185-1209)
//   a5        the one or two State[] writes of ChangeFlow                   (NS.cs:1030-1039)
//
// Per pivot there is normally ONE dispatch: the patches of the previous pivot ride in the kernel
// arguments of the next search and every workgroup applies them (idempotent stores of final values)
// before it reads anything -- see DESIGN.md "Inline patches".  Lists that do not fit go through
// update_kernel first.  No MFMA: there is no contraction on this path; it is bound by memory
// bandwidth / latency.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

constexpr int kThreads = 256;              // 4 wavefronts of 64
constexpr int kArcsPerThread = 4;          // 16-byte loads of source/target, 4-byte load of state
constexpr int kTile = kThreads * kArcsPerThread;   // 1024 arcs per workgroup per step
constexpr int kPad = 2 * kTile;            // device arrays are padded to this with state = 0
constexpr int kInlinePi = 96;              // potentials patched through the kernel arguments
constexpr int kInlineState = 4;
constexpr int kMaxWorkgroups = 2048;
constexpr uint32_t kNone = 0xFFFFFFFFu;

// 16-byte answer of one workgroup, written with one store into pinned host memory.
// tag is last so that a host that sees the tag sees the payload (one PCIe write, ascending addresses).
struct alignas(16) Slot {
    int64_t c;
    uint32_t p;
    uint32_t tag;
};

struct Key {
    int64_t c;
    uint32_t r;   // Block Search: rank of the block in scan order (doubled, +1 for the wrapped half in OPTIMIZED)
    uint32_t p;   // Best: arc index; First/Block: position in the cyclic scan that starts at next_arc
};

template <typename T>
struct ScanParams {
    const int32_t *src;
    const int32_t *tgt;
    const T *cost;
    int8_t *state;
    T *pi;
    Slot *slots;
    int32_t base;          // global index of local arc 0
    int32_t count_padded;  // local arcs incl. padding (multiple of kPad)
    int32_t m_s;           // global search_arc_num
    int32_t next_arc;      // start of the cyclic scan, already reduced to [0, m_s)
    int32_t block_size;
    int32_t rstar;         // OPTIMIZED Block Search: rank of the block cut by the end of the arrays, or -1
    uint32_t seq;
    int32_t n_pi, n_st;
    int32_t st_arc[kInlineState];
    int32_t st_val[kInlineState];
    int32_t pi_node[kInlinePi];
    T pi_val[kInlinePi];
};

// k = better(o, k) ? o : k, written as per-field selects.  (hipcc 7.2 mis-compiled the whole-struct form
// `if (better(o, k)) k = o;` in the int64 Block Search kernel: c was updated, p was not -- caught by the parity tests.)
template <int RULE>
__device__ __forceinline__ void take_if_better(Key &k, int64_t oc, uint32_t orank, uint32_t op)
{
    bool take;
    if (RULE == MCF_RULE_BEST_ELIGIBLE) take = oc < k.c || (oc == k.c && op < k.p);
    else if (RULE == MCF_RULE_FIRST_ELIGIBLE) take = op < k.p;
    else take = orank < k.r || (orank == k.r && (oc < k.c || (oc == k.c && op < k.p)));
    k.c = take ? oc : k.c;
    k.r = take ? orank : k.r;
    k.p = take ? op : k.p;
}

template <int RULE>
__device__ __forceinline__ Key wave_min(Key k)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int64_t oc = __shfl_xor(k.c, off, 64);
        const uint32_t op = __shfl_xor(k.p, off, 64);
        const uint32_t orank = (RULE == MCF_RULE_BLOCK_SEARCH) ? __shfl_xor(k.r, off, 64) : 0u;
        take_if_better<RULE>(k, oc, orank, op);
    }
    return k;
}

template <typename T> struct Vec4;
template <> struct Vec4<int32_t> {
    int32_t v[4];
    __device__ __forceinline__ void load(const int32_t *p) { const int4 a = *reinterpret_cast<const int4 *>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
};
template <> struct Vec4<int64_t> {
    int64_t v[4];
    __device__ __forceinline__ void load(const int64_t *p)
    {
        const longlong2 a = *reinterpret_cast<const longlong2 *>(p), b = *reinterpret_cast<const longlong2 *>(p + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
};

// One tile = 1024 consecutive arcs, 4 per thread.  All loads of a tile are issued before any use.
template <typename T, int RULE, bool OPT>
__device__ __forceinline__ void scan_tile(const ScanParams<T> &p, int i0, Key &best)
{
    const uint32_t st4 = *reinterpret_cast<const uint32_t *>(p.state + i0);
    Vec4<int32_t> s, t;
    Vec4<T> c;
    s.load(p.src + i0);
    t.load(p.tgt + i0);
    c.load(p.cost + i0);
    T ps[4], pt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ps[j] = p.pi[s.v[j]]; pt[j] = p.pi[t.v[j]]; }
    uint32_t pos0 = 0;
    const int e0 = p.base + i0;
    if (RULE != MCF_RULE_BEST_ELIGIBLE) {
        int d = e0 - p.next_arc;
        if (d < 0) d += p.m_s;
        pos0 = (uint32_t)d;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int st = (int)(int8_t)(st4 >> (8 * j));
        // 64-bit arithmetic in both widths: int32 inputs cannot overflow it
        const int64_t d = (int64_t)c.v[j] + (int64_t)ps[j] - (int64_t)pt[j];
        const int64_t rc = st > 0 ? d : (st < 0 ? -d : 0);
        if (RULE == MCF_RULE_BEST_ELIGIBLE) {
            if (rc < best.c) { best.c = rc; best.p = (uint32_t)(e0 + j); }   // strict <: lowest arc wins ties
        } else {
            uint32_t pos = pos0 + j;   // the group may straddle the wrap point
            if (pos >= (uint32_t)p.m_s) pos -= (uint32_t)p.m_s;
            if (RULE == MCF_RULE_FIRST_ELIGIBLE) {
                if (rc < 0) take_if_better<RULE>(best, rc, 0u, pos);
            } else {
                uint32_t r = pos / (uint32_t)p.block_size;
                r = 2 * r + ((OPT && (int)r == p.rstar && e0 + j < p.next_arc) ? 1u : 0u);
                if (rc < 0) take_if_better<RULE>(best, rc, r, pos);
            }
        }
    }
}

template <typename T, int RULE, bool OPT, int UNROLL>
__global__ __launch_bounds__(kThreads) void scan_kernel(const ScanParams<T> p)
{
    const int tid = threadIdx.x;
    // ---- inline patches of the previous pivot: final values, applied by EVERY workgroup before it reads
    if (p.n_pi | p.n_st) {
        if (tid < p.n_pi) p.pi[p.pi_node[tid]] = p.pi_val[tid];
        if (tid >= kThreads - kInlineState && tid - (kThreads - kInlineState) < p.n_st) {
            const int k = tid - (kThreads - kInlineState);
            const int a = p.st_arc[k] - p.base;
            if ((unsigned)a < (unsigned)p.count_padded) p.state[a] = (int8_t)p.st_val[k];
        }
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): this wave's stores are acknowledged
        __syncthreads();
    }

    Key best;
    best.c = 0;
    best.r = kNone;
    best.p = kNone;
    const int step = gridDim.x * kTile * UNROLL;
    for (int i0 = blockIdx.x * kTile * UNROLL + tid * kArcsPerThread; i0 < p.count_padded; i0 += step) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) scan_tile<T, RULE, OPT>(p, i0 + u * kTile, best);
    }

    best = wave_min<RULE>(best);
    __shared__ Key wave_best[kThreads / 64];
    if ((tid & 63) == 0) wave_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        Key k = wave_best[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) take_if_better<RULE>(k, wave_best[w].c, wave_best[w].r, wave_best[w].p);
        uint4 out;
        out.x = (uint32_t)(uint64_t)k.c;
        out.y = (uint32_t)((uint64_t)k.c >> 32);
        out.z = k.p;
        out.w = p.seq;
        *reinterpret_cast<uint4 *>(p.slots + blockIdx.x) = out;
    }
}

// pi[node[i]] = value[i], state[arc[j]] = s[j]; lists read straight from pinned host memory
template <typename T>
__global__ __launch_bounds__(kThreads) void update_kernel(T *pi, const int32_t *nodes, const int64_t *values, int n_pi,
                                                          int8_t *state, const int32_t *arcs, const int32_t *states,
                                                          int n_st, int base, int count_padded)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_pi) pi[nodes[i]] = (T)values[i];
    if (i < n_st) {
        const int a = arcs[i] - base;
        if ((unsigned)a < (unsigned)count_padded) state[a] = (int8_t)states[i];
    }
}

__global__ void flush_kernel(uint4 *buf, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) { uint4 v = buf[i]; v.x += 1; buf[i] = v; }
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t err__ = (expr);                                                                         \
        if (err__ != hipSuccess) return mcf::fail(MCF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(err__)); \
    } while (0)

// --- RCCL, bound lazily so that the library loads on machines without it
struct Id128 { char b[128]; };
typedef int (*nccl_get_unique_id_t)(Id128 *);
typedef int (*nccl_comm_init_rank_t)(void **, int, Id128, int);
typedef int (*nccl_all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_comm_destroy_t)(void *);
typedef const char *(*nccl_get_error_string_t)(int);
struct RcclApi {
    void *lib = nullptr;
    nccl_get_unique_id_t get_unique_id = nullptr;
    nccl_comm_init_rank_t comm_init_rank = nullptr;
    nccl_all_gather_t all_gather = nullptr;
    nccl_comm_destroy_t comm_destroy = nullptr;
    nccl_get_error_string_t error_string = nullptr;
};
RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) {
            api.get_unique_id = (nccl_get_unique_id_t)dlsym(api.lib, "ncclGetUniqueId");
            api.comm_init_rank = (nccl_comm_init_rank_t)dlsym(api.lib, "ncclCommInitRank");
            api.all_gather = (nccl_all_gather_t)dlsym(api.lib, "ncclAllGather");
            api.comm_destroy = (nccl_comm_destroy_t)dlsym(api.lib, "ncclCommDestroy");
            api.error_string = (nccl_get_error_string_t)dlsym(api.lib, "ncclGetErrorString");
        }
    }
    return (api.lib && api.get_unique_id && api.comm_init_rank && api.all_gather) ? &api : nullptr;
}
constexpr int kNcclChar = 0;   // ncclInt8 / ncclChar

}  // namespace

// ------------------------------------------------------------------------------------------------ engine

struct mcf_engine {
    mcf_engine_desc d{};
    int begin = 0, end = 0;        // shard [begin, end) of the search arcs
    int count_padded = 0;
    int next_arc = 0, block_size = 0;
    hipStream_t stream = nullptr;
    int32_t *d_src = nullptr, *d_tgt = nullptr;
    void *d_cost = nullptr, *d_pi = nullptr;
    int8_t *d_state = nullptr;
    Slot *h_slots = nullptr, *d_slots = nullptr;    // pinned host memory and its device alias
    int grid = 0, unroll = 1;
    uint32_t seq = 0;
    bool uploaded = false;
    // host mirror of pi: patches carry final values
    std::vector<int64_t> pi;
    int64_t max_abs_cost = 0;
    // pending patches of the current pivot
    std::vector<int32_t> pend_node, pend_arc, pend_state;
    std::vector<int64_t> pend_val;
    // staging for lists that do not fit the kernel arguments (pinned, read by update_kernel over PCIe)
    struct Staging {
        int32_t *nodes = nullptr, *arcs = nullptr, *states = nullptr;
        int64_t *values = nullptr;
        void *d_nodes = nullptr, *d_arcs = nullptr, *d_states = nullptr, *d_values = nullptr;
        hipEvent_t done = nullptr;
        bool busy = false;
        int cap_nodes = 0;
    } stage[2];
    int stage_next = 0;
    // kernel timing samples
    static constexpr int kEvRing = 64;
    hipEvent_t ev_start[kEvRing]{}, ev_stop[kEvRing]{};
    int ev_head = 0, ev_tail = 0;
    mcf_engine_stats st{};
    // sharding
    void *comm = nullptr;
    int rank = 0, world = 1;
    mcf_candidate *d_cand_local = nullptr, *d_cand_all = nullptr, *h_cand_all = nullptr;
    // flush buffer for cold micro-benchmarks
    void *d_flush = nullptr;
    size_t flush_bytes = 0;
};

namespace {

bool fits32(int64_t v) { return v >= INT32_MIN && v <= INT32_MAX; }

int drain_events(mcf_engine *e, bool all)
{
    while (e->ev_tail != e->ev_head) {
        const int i = e->ev_tail % mcf_engine::kEvRing;
        if (!all && hipEventQuery(e->ev_stop[i]) != hipSuccess) break;
        HIP_TRY(hipEventSynchronize(e->ev_stop[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev_start[i], e->ev_stop[i]));
        e->st.timed_scans += 1;
        e->st.timed_scan_ns += (double)ms * 1e6;
        e->ev_tail++;
    }
    return MCF_OK;
}

template <typename T>
void fill_params(mcf_engine *e, ScanParams<T> &p, bool with_patches)
{
    p.src = e->d_src;
    p.tgt = e->d_tgt;
    p.cost = (const T *)e->d_cost;
    p.state = e->d_state;
    p.pi = (T *)e->d_pi;
    p.slots = e->d_slots;
    p.base = e->begin;
    p.count_padded = e->count_padded;
    p.m_s = e->d.search_arc_num;
    const int na = e->next_arc >= e->d.search_arc_num ? 0 : e->next_arc;
    p.next_arc = na;
    p.block_size = e->block_size;
    p.rstar = -1;
    if (e->d.rule == MCF_RULE_BLOCK_SEARCH && e->d.semantics == MCF_SEM_OPTIMIZED && e->next_arc < e->d.search_arc_num) {
        const int len1 = e->d.search_arc_num - e->next_arc;   // BSPO.cs:49 first range
        if (len1 % e->block_size != 0) p.rstar = len1 / e->block_size;
    }
    p.seq = e->seq;
    p.n_pi = 0;
    p.n_st = 0;
    if (with_patches) {
        p.n_pi = (int)e->pend_node.size();
        p.n_st = (int)e->pend_arc.size();
        for (int i = 0; i < p.n_pi; ++i) { p.pi_node[i] = e->pend_node[i]; p.pi_val[i] = (T)e->pend_val[i]; }
        for (int i = 0; i < p.n_st; ++i) { p.st_arc[i] = e->pend_arc[i]; p.st_val[i] = e->pend_state[i]; }
    }
}

template <typename T, int RULE, bool OPT>
void launch_scan_u(mcf_engine *e, const ScanParams<T> &p, hipEvent_t start, hipEvent_t stop)
{
    const dim3 grid(e->grid), block(kThreads);
    if (e->unroll == 2) {
        if (start) hipExtLaunchKernelGGL((scan_kernel<T, RULE, OPT, 2>), grid, block, 0, e->stream, start, stop, 0, p);
        else hipLaunchKernelGGL((scan_kernel<T, RULE, OPT, 2>), grid, block, 0, e->stream, p);
    } else {
        if (start) hipExtLaunchKernelGGL((scan_kernel<T, RULE, OPT, 1>), grid, block, 0, e->stream, start, stop, 0, p);
        else hipLaunchKernelGGL((scan_kernel<T, RULE, OPT, 1>), grid, block, 0, e->stream, p);
    }
}

template <typename T>
void dispatch_scan(mcf_engine *e, const ScanParams<T> &p, hipEvent_t start, hipEvent_t stop)
{
    const bool opt = e->d.semantics == MCF_SEM_OPTIMIZED;
    switch (e->d.rule) {
    case MCF_RULE_BEST_ELIGIBLE: launch_scan_u<T, MCF_RULE_BEST_ELIGIBLE, false>(e, p, start, stop); break;
    case MCF_RULE_FIRST_ELIGIBLE: launch_scan_u<T, MCF_RULE_FIRST_ELIGIBLE, false>(e, p, start, stop); break;
    default:
        if (opt) launch_scan_u<T, MCF_RULE_BLOCK_SEARCH, true>(e, p, start, stop);
        else launch_scan_u<T, MCF_RULE_BLOCK_SEARCH, false>(e, p, start, stop);
    }
}

template <typename T>
int launch_scan(mcf_engine *e, bool with_patches, bool timed)
{
    ScanParams<T> p;
    fill_params(e, p, with_patches);
    hipEvent_t start = nullptr, stop = nullptr;
    if (timed) {
        if (e->ev_head - e->ev_tail >= mcf_engine::kEvRing) { int rc = drain_events(e, true); if (rc) return rc; }
        const int i = e->ev_head % mcf_engine::kEvRing;
        start = e->ev_start[i];
        stop = e->ev_stop[i];
        e->ev_head++;
    }
    dispatch_scan<T>(e, p, start, stop);
    HIP_TRY(hipGetLastError());
    e->st.scan_launches += 1;
    e->st.arcs_scanned += e->end - e->begin;
    return MCF_OK;
}

// ship the pending patches with update_kernel (lists too long for the kernel arguments, or explicit flush)
int flush_pending(mcf_engine *e)
{
    const int n_pi = (int)e->pend_node.size(), n_st = (int)e->pend_arc.size();
    if (n_pi == 0 && n_st == 0) return MCF_OK;
    mcf_engine::Staging &s = e->stage[e->stage_next];
    e->stage_next ^= 1;
    if (s.busy) { HIP_TRY(hipEventSynchronize(s.done)); s.busy = false; }
    if (n_pi) { memcpy(s.nodes, e->pend_node.data(), sizeof(int32_t) * n_pi); memcpy(s.values, e->pend_val.data(), sizeof(int64_t) * n_pi); }
    if (n_st) { memcpy(s.arcs, e->pend_arc.data(), sizeof(int32_t) * n_st); memcpy(s.states, e->pend_state.data(), sizeof(int32_t) * n_st); }
    const int blocks = (std::max(n_pi, n_st) + kThreads - 1) / kThreads;
    if (e->d.int_width == 32)
        hipLaunchKernelGGL(update_kernel<int32_t>, dim3(blocks), dim3(kThreads), 0, e->stream, (int32_t *)e->d_pi, (const int32_t *)s.d_nodes,
                           (const int64_t *)s.d_values, n_pi, e->d_state, (const int32_t *)s.d_arcs, (const int32_t *)s.d_states, n_st, e->begin, e->count_padded);
    else
        hipLaunchKernelGGL(update_kernel<int64_t>, dim3(blocks), dim3(kThreads), 0, e->stream, (int64_t *)e->d_pi, (const int32_t *)s.d_nodes,
                           (const int64_t *)s.d_values, n_pi, e->d_state, (const int32_t *)s.d_arcs, (const int32_t *)s.d_states, n_st, e->begin, e->count_padded);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s.done, e->stream));
    s.busy = true;
    e->st.update_launches += 1;
    e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
    return MCF_OK;
}

// wait for the `grid` records of dispatch `seq`, merge them with the rule's ordering
int collect(mcf_engine *e, Key *out)
{
    const double t0 = mcf::now_ns();
    const bool block_rule = e->d.rule == MCF_RULE_BLOCK_SEARCH, best_rule = e->d.rule == MCF_RULE_BEST_ELIGIBLE;
    Key best{0, kNone, kNone};
    const volatile Slot *slots = e->h_slots;
    const uint32_t seq = e->seq;
    const int rstar = [&] {
        if (!(block_rule && e->d.semantics == MCF_SEM_OPTIMIZED) || e->next_arc >= e->d.search_arc_num) return -1;
        const int len1 = e->d.search_arc_num - e->next_arc;
        return len1 % e->block_size ? len1 / e->block_size : -1;
    }();
    const int na = e->next_arc >= e->d.search_arc_num ? 0 : e->next_arc;
    for (int g = 0; g < e->grid; ++g) {
        uint64_t spins = 0;
        while (slots[g].tag != seq) {
            _mm_pause();
            if ((++spins & 0xFFFFF) == 0) {
                const hipError_t q = hipStreamQuery(e->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return mcf::fail(MCF_ERR_HIP, "scan dispatch failed: %s", hipGetErrorString(q));
                if (mcf::now_ns() - t0 > 20e9) return mcf::fail(MCF_ERR_TIMEOUT, "no answer from the device after 20 s (workgroup %d of %d)", g, e->grid);
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        Key k;
        k.c = slots[g].c;
        k.p = slots[g].p;
        k.r = 0;
        if (k.p == kNone) continue;
        bool take;
        if (best_rule) take = best.p == kNone || k.c < best.c || (k.c == best.c && k.p < best.p);
        else if (!block_rule) take = k.p < best.p;
        else {
            uint32_t r = k.p / (uint32_t)e->block_size;
            const int arc = (int)((k.p + (uint32_t)na) % (uint32_t)e->d.search_arc_num);
            k.r = 2 * r + ((rstar >= 0 && (int)r == rstar && arc < e->next_arc) ? 1u : 0u);
            take = best.p == kNone || k.r < best.r || (k.r == best.r && (k.c < best.c || (k.c == best.c && k.p < best.p)));
        }
        if (take) best = k;
    }
    e->st.host_wait_ns += mcf::now_ns() - t0;
    *out = best;
    return MCF_OK;
}

int local_search(mcf_engine *e, Key *k)
{
    if (!e->uploaded) return mcf::fail(MCF_ERR_STATE, "mcf_engine_upload has not been called");
    const double t0 = mcf::now_ns();
    e->seq += 1;
    if (e->seq == 0) e->seq = 1;
    const bool inline_ok = !(e->d.flags & MCF_ENGINE_NO_INLINE_UPDATE) && (int)e->pend_node.size() <= kInlinePi &&
                           (int)e->pend_arc.size() <= kInlineState;
    const bool had = !e->pend_node.empty() || !e->pend_arc.empty();
    if (!inline_ok) { int rc = flush_pending(e); if (rc) return rc; }
    const bool timed = (e->d.flags & MCF_ENGINE_TIME_EVERY_KERNEL) ||
                       ((e->d.flags & MCF_ENGINE_SAMPLE_KERNEL_TIME) && (e->st.scan_launches & 15) == 0);
    int rc = e->d.int_width == 32 ? launch_scan<int32_t>(e, inline_ok, timed) : launch_scan<int64_t>(e, inline_ok, timed);
    if (rc) return rc;
    if (inline_ok) {
        if (had) e->st.inline_updates += 1;
        e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
    }
    e->st.host_launch_ns += mcf::now_ns() - t0;
    e->st.searches += 1;
    rc = collect(e, k);
    if (rc) return rc;
    if (timed) drain_events(e, false);
    return MCF_OK;
}

// entering arc, reduced cost and the rule's next_arc from the winning key (host part of the rules)
void resolve_key_raw(int rule, int semantics, int m_s, int B, int &next_arc, const Key &k, int32_t *found, int32_t *arc, int64_t *rcost)
{
    if (k.p == kNone) { *found = 0; *arc = -1; if (rcost) *rcost = 0; return; }
    *found = 1;
    if (rcost) *rcost = k.c;
    if (rule == MCF_RULE_BEST_ELIGIBLE) { *arc = (int32_t)k.p; return; }          // stateless: NS.cs:1644-1667
    const int na = next_arc >= m_s ? 0 : next_arc;
    const int a = (int)(((int64_t)k.p + na) % m_s);
    *arc = a;
    if (rule == MCF_RULE_FIRST_ELIGIBLE) { next_arc = a + 1; return; }             // NS.cs:1617, BSPO.cs:199
    const int64_t r = k.p / B, boundary = (r + 1) * (int64_t)B - 1;   // scan position of the block's last arc
    if (semantics != MCF_SEM_OPTIMIZED) {
        // NS.cs:1358-1397: stop at the boundary -> next_arc = that arc; cycle exhausted first -> unchanged
        if (boundary <= m_s - 1) next_arc = (int)((boundary + na) % m_s);
        return;
    }
    // BSPO.cs:49-63,98-103: first range [next_arc, m_s), wrapped range [0, next_arc) only if nothing was found
    const int64_t len1 = next_arc >= m_s ? 0 : m_s - next_arc;
    if (a >= next_arc && next_arc < m_s) {
        next_arc = boundary < len1 ? (int)(next_arc + boundary + 1) : m_s;
    } else {
        if (boundary < m_s) next_arc = (int)(boundary - len1 + 1);
        /* else: the range ended first, ProcessArcRange returns `end` = the old next_arc */
    }
}

void resolve_key(mcf_engine *e, const Key &k, int32_t *found, int32_t *arc, int64_t *rcost)
{
    resolve_key_raw(e->d.rule, e->d.semantics, e->d.search_arc_num, e->block_size, e->next_arc, k, found, arc, rcost);
}

// MINLOC over the shards' candidates with the rule's ordering
Key merge_candidates(int rule, int semantics, int m_s, int B, int next_arc, int count, const mcf_candidate *all)
{
    const bool block_rule = rule == MCF_RULE_BLOCK_SEARCH, best_rule = rule == MCF_RULE_BEST_ELIGIBLE;
    int rstar = -1;
    if (block_rule && semantics == MCF_SEM_OPTIMIZED && next_arc < m_s) {
        const int len1 = m_s - next_arc;
        if (len1 % B) rstar = len1 / B;
    }
    Key best{0, kNone, kNone};
    for (int i = 0; i < count; ++i) {
        if (all[i].pos == kNone) continue;
        Key k{all[i].reduced_cost, 0, all[i].pos};
        bool take;
        if (best_rule) take = best.p == kNone || k.c < best.c || (k.c == best.c && k.p < best.p);
        else if (!block_rule) take = k.p < best.p;
        else {
            const uint32_t r = k.p / (uint32_t)B;
            k.r = 2 * r + ((rstar >= 0 && (int)r == rstar && all[i].arc < next_arc) ? 1u : 0u);
            take = best.p == kNone || k.r < best.r || (k.r == best.r && (k.c < best.c || (k.c == best.c && k.p < best.p)));
        }
        if (take) best = k;
    }
    return best;
}

}  // namespace

extern "C" {

int mcf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int mcf_shard_range(int32_t m_s, int32_t rank, int32_t world, int32_t *begin, int32_t *end)
{
    if (world < 1 || rank < 0 || rank >= world || m_s < 0 || !begin || !end) return mcf::fail(MCF_ERR_INVALID, "mcf_shard_range: bad arguments");
    const int64_t groups = ((int64_t)m_s + kArcsPerThread - 1) / kArcsPerThread;
    const int64_t b = groups * rank / world * kArcsPerThread, en = groups * (rank + 1) / world * kArcsPerThread;
    *begin = (int32_t)std::min<int64_t>(b, m_s);
    *end = (int32_t)std::min<int64_t>(en, m_s);
    return MCF_OK;
}

int mcf_engine_create(mcf_engine **out, const mcf_engine_desc *desc)
{
    if (!out || !desc) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_create: null argument");
    *out = nullptr;
    if (desc->node_count < 1 || desc->search_arc_num < 0 || desc->arc_capacity < desc->search_arc_num)
        return mcf::fail(MCF_ERR_INVALID, "mcf_engine_create: bad sizes (nodes %d, arcs %d, search %d)", desc->node_count, desc->arc_capacity, desc->search_arc_num);
    if (desc->int_width != 32 && desc->int_width != 64) return mcf::fail(MCF_ERR_INVALID, "int_width must be 32 or 64");
    if (desc->rule < 0 || desc->rule > 2) return mcf::fail(MCF_ERR_INVALID, "pivot rule %d not implemented (NS.cs:884)", desc->rule);
    if (desc->semantics != MCF_SEM_PLAIN && desc->semantics != MCF_SEM_OPTIMIZED) return mcf::fail(MCF_ERR_INVALID, "bad semantics %d", desc->semantics);
    if (mcf_device_count() <= desc->device || desc->device < 0)
        return mcf::fail(MCF_ERR_NO_DEVICE, "HIP device %d not available (%d visible); this library has no CPU search path", desc->device, mcf_device_count());
    HIP_TRY(hipSetDevice(desc->device));
    mcf_engine *e = new mcf_engine();
    e->d = *desc;
    e->begin = desc->shard_begin;
    e->end = desc->shard_end;
    if (e->begin == 0 && e->end == 0) e->end = desc->search_arc_num;
    if (e->begin < 0 || e->end < e->begin || e->end > desc->search_arc_num || (e->begin % kArcsPerThread) != 0) {
        delete e;
        return mcf::fail(MCF_ERR_INVALID, "bad shard [%d, %d)", desc->shard_begin, desc->shard_end);
    }
    const int count = e->end - e->begin;
    e->count_padded = std::max(kPad, (count + kPad - 1) / kPad * kPad);
    e->block_size = desc->block_size > 0 ? desc->block_size : mcf::default_block_size(desc->search_arc_num, desc->semantics);
    e->unroll = count > (1 << 20) ? 2 : 1;
    const int groups = e->count_padded / (kTile * e->unroll);
    e->grid = desc->scan_workgroups > 0 ? std::min(desc->scan_workgroups, kMaxWorkgroups) : std::min(groups, kMaxWorkgroups);
    e->grid = std::max(1, std::min(e->grid, groups));
    const size_t w = desc->int_width / 8;
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess && x != hipSuccess) err = x; };
    chk(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    chk(hipMalloc((void **)&e->d_src, sizeof(int32_t) * e->count_padded));
    chk(hipMalloc((void **)&e->d_tgt, sizeof(int32_t) * e->count_padded));
    chk(hipMalloc(&e->d_cost, w * e->count_padded));
    chk(hipMalloc((void **)&e->d_state, e->count_padded));
    chk(hipMalloc(&e->d_pi, w * (size_t)desc->node_count));
    chk(hipHostMalloc((void **)&e->h_slots, sizeof(Slot) * kMaxWorkgroups, hipHostMallocMapped | hipHostMallocCoherent));
    if (err == hipSuccess) {
        memset(e->h_slots, 0, sizeof(Slot) * kMaxWorkgroups);
        chk(hipHostGetDevicePointer((void **)&e->d_slots, e->h_slots, 0));
    }
    for (int i = 0; i < 2 && err == hipSuccess; ++i) {
        mcf_engine::Staging &s = e->stage[i];
        s.cap_nodes = desc->node_count;
        chk(hipHostMalloc((void **)&s.nodes, sizeof(int32_t) * s.cap_nodes, hipHostMallocMapped | hipHostMallocCoherent));
        chk(hipHostMalloc((void **)&s.values, sizeof(int64_t) * s.cap_nodes, hipHostMallocMapped | hipHostMallocCoherent));
        chk(hipHostMalloc((void **)&s.arcs, sizeof(int32_t) * 64, hipHostMallocMapped | hipHostMallocCoherent));
        chk(hipHostMalloc((void **)&s.states, sizeof(int32_t) * 64, hipHostMallocMapped | hipHostMallocCoherent));
        if (err == hipSuccess) {
            chk(hipHostGetDevicePointer(&s.d_nodes, s.nodes, 0));
            chk(hipHostGetDevicePointer(&s.d_values, s.values, 0));
            chk(hipHostGetDevicePointer(&s.d_arcs, s.arcs, 0));
            chk(hipHostGetDevicePointer(&s.d_states, s.states, 0));
        }
        chk(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    for (int i = 0; i < mcf_engine::kEvRing && err == hipSuccess; ++i) { chk(hipEventCreate(&e->ev_start[i])); chk(hipEventCreate(&e->ev_stop[i])); }
    if (err != hipSuccess) {
        const int rc = mcf::fail(MCF_ERR_HIP, "mcf_engine_create: %s", hipGetErrorString(err));
        mcf_engine_destroy(e);
        return rc;
    }
    e->st.scan_workgroups = e->grid;
    e->st.scan_threads = kThreads;
    e->st.bytes_per_scan = (int64_t)(desc->int_width == 64 ? 17 : 13) * count + (int64_t)w * desc->node_count;
    *out = e;
    return MCF_OK;
}

void mcf_engine_destroy(mcf_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->d.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->comm && rccl() && rccl()->comm_destroy) rccl()->comm_destroy(e->comm);
    (void)hipFree(e->d_src); (void)hipFree(e->d_tgt); (void)hipFree(e->d_cost); (void)hipFree(e->d_state); (void)hipFree(e->d_pi);
    (void)hipFree(e->d_cand_local); (void)hipFree(e->d_cand_all); (void)hipFree(e->d_flush);
    if (e->h_cand_all) (void)hipHostFree(e->h_cand_all);
    if (e->h_slots) (void)hipHostFree(e->h_slots);
    for (auto &s : e->stage) {
        if (s.nodes) (void)hipHostFree(s.nodes);
        if (s.values) (void)hipHostFree(s.values);
        if (s.arcs) (void)hipHostFree(s.arcs);
        if (s.states) (void)hipHostFree(s.states);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (int i = 0; i < mcf_engine::kEvRing; ++i) {
        if (e->ev_start[i]) (void)hipEventDestroy(e->ev_start[i]);
        if (e->ev_stop[i]) (void)hipEventDestroy(e->ev_stop[i]);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int mcf_engine_upload(mcf_engine *e, const int32_t *source, const int32_t *target, const int64_t *cost, const int8_t *state, const int64_t *pi)
{
    if (!e || !source || !target || !cost || !state || !pi) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_upload: null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    const int n = e->d.node_count, count = e->end - e->begin, cp = e->count_padded;
    for (int i = e->begin; i < e->end; ++i)
        if ((unsigned)source[i] >= (unsigned)n || (unsigned)target[i] >= (unsigned)n)
            return mcf::fail(MCF_ERR_INVALID, "arc %d has an end point outside [0, %d)", i, n);
    int64_t maxc = 0;
    for (int i = e->begin; i < e->end; ++i) maxc = std::max<int64_t>(maxc, cost[i] < 0 ? -cost[i] : cost[i]);
    e->max_abs_cost = maxc;
    std::vector<int32_t> s(cp, 0), t(cp, 0);
    std::vector<int8_t> st(cp, 0);
    memcpy(s.data(), source + e->begin, sizeof(int32_t) * count);
    memcpy(t.data(), target + e->begin, sizeof(int32_t) * count);
    memcpy(st.data(), state + e->begin, count);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->d_src, s.data(), sizeof(int32_t) * cp, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_tgt, t.data(), sizeof(int32_t) * cp, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_state, st.data(), cp, hipMemcpyHostToDevice));
    if (e->d.int_width == 32) {
        // d = cost + pi[s] - pi[t] is formed in 64 bits on the device, so each operand only has to fit int32
        std::vector<int32_t> c(cp, 0), p(n);
        for (int i = 0; i < count; ++i) {
            if (!fits32(cost[e->begin + i])) return mcf::fail(MCF_ERR_OVERFLOW, "cost of arc %d does not fit int32", e->begin + i);
            c[i] = (int32_t)cost[e->begin + i];
        }
        for (int i = 0; i < n; ++i) {
            if (!fits32(pi[i])) return mcf::fail(MCF_ERR_OVERFLOW, "potential of node %d does not fit int32", i);
            p[i] = (int32_t)pi[i];
        }
        HIP_TRY(hipMemcpy(e->d_cost, c.data(), sizeof(int32_t) * cp, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_pi, p.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    } else {
        std::vector<int64_t> c(cp, 0);
        memcpy(c.data(), cost + e->begin, sizeof(int64_t) * count);
        HIP_TRY(hipMemcpy(e->d_cost, c.data(), sizeof(int64_t) * cp, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_pi, pi, sizeof(int64_t) * n, hipMemcpyHostToDevice));
    }
    e->pi.assign(pi, pi + n);
    e->pend_node.clear(); e->pend_val.clear(); e->pend_arc.clear(); e->pend_state.clear();
    e->next_arc = 0;
    e->uploaded = true;
    return MCF_OK;
}

int mcf_engine_patch_state(mcf_engine *e, int32_t count, const int32_t *arcs, const int8_t *states)
{
    if (!e || count < 0 || (count && (!arcs || !states))) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_patch_state: bad arguments");
    for (int i = 0; i < count; ++i) {
        if (arcs[i] < 0 || arcs[i] >= e->d.arc_capacity) return mcf::fail(MCF_ERR_INVALID, "arc %d out of range", arcs[i]);
        if (states[i] < -1 || states[i] > 1) return mcf::fail(MCF_ERR_INVALID, "state %d is not -1/0/1", states[i]);
        if (arcs[i] < e->begin || arcs[i] >= e->end) continue;   // not resident here (outside the search range or another shard)
        bool dup = false;
        for (size_t j = 0; j < e->pend_arc.size(); ++j)
            if (e->pend_arc[j] == arcs[i]) { e->pend_state[j] = states[i]; dup = true; }
        if (dup) continue;
        if ((int)e->pend_arc.size() >= 64) { int rc = flush_pending(e); if (rc) return rc; }
        e->pend_arc.push_back(arcs[i]);
        e->pend_state.push_back(states[i]);
    }
    return MCF_OK;
}

int mcf_engine_update_potential(mcf_engine *e, int32_t count, const int32_t *nodes, int64_t sigma)
{
    if (!e || count < 0 || (count && !nodes)) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_update_potential: bad arguments");
    if (!e->uploaded) return mcf::fail(MCF_ERR_STATE, "mcf_engine_upload has not been called");
    if (count == 0) return MCF_OK;
    // one list per dispatch: a second list may repeat nodes of the first
    if (!e->pend_node.empty()) { int rc = flush_pending(e); if (rc) return rc; }
    if (count > e->d.node_count) return mcf::fail(MCF_ERR_INVALID, "%d nodes in a graph of %d", count, e->d.node_count);
    e->pend_node.resize(count);
    e->pend_val.resize(count);
    for (int i = 0; i < count; ++i) {
        const int u = nodes[i];
        if ((unsigned)u >= (unsigned)e->d.node_count) { e->pend_node.clear(); e->pend_val.clear(); return mcf::fail(MCF_ERR_INVALID, "node %d out of range", u); }
        const int64_t v = e->pi[u] + sigma;
        if (e->d.int_width == 32 && !fits32(v)) { e->pend_node.clear(); e->pend_val.clear(); return mcf::fail(MCF_ERR_OVERFLOW, "potential of node %d leaves int32; create the engine with int_width 64", u); }
        e->pend_node[i] = u;
        e->pend_val[i] = v;
    }
    for (int i = 0; i < count; ++i) e->pi[nodes[i]] = e->pend_val[i];
    e->st.potential_nodes += count;
    return MCF_OK;
}

int mcf_engine_patch_arcs(mcf_engine *e, int32_t count, const int32_t *arcs, const int32_t *source, const int32_t *target, const int64_t *cost)
{
    if (!e || count < 0 || (count && (!arcs || !source || !target || !cost))) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_patch_arcs: bad arguments");
    HIP_TRY(hipSetDevice(e->d.device));
    int rc = flush_pending(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < count; ++i) {
        const int a = arcs[i];
        if (a < 0 || a >= e->d.arc_capacity) return mcf::fail(MCF_ERR_INVALID, "arc %d out of range", a);
        if ((unsigned)source[i] >= (unsigned)e->d.node_count || (unsigned)target[i] >= (unsigned)e->d.node_count) return mcf::fail(MCF_ERR_INVALID, "arc %d: end point out of range", a);
        if (a < e->begin || a >= e->end) continue;
        const int l = a - e->begin;
        HIP_TRY(hipMemcpy(e->d_src + l, &source[i], 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_tgt + l, &target[i], 4, hipMemcpyHostToDevice));
        if (e->d.int_width == 32) {
            if (!fits32(cost[i])) return mcf::fail(MCF_ERR_OVERFLOW, "cost of arc %d does not fit int32", a);
            const int32_t c = (int32_t)cost[i];
            HIP_TRY(hipMemcpy((int32_t *)e->d_cost + l, &c, 4, hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy((int64_t *)e->d_cost + l, &cost[i], 8, hipMemcpyHostToDevice));
        }
    }
    return MCF_OK;
}

int mcf_engine_find_entering(mcf_engine *e, int32_t *found, int32_t *arc, int64_t *reduced_cost)
{
    if (!e || !found || !arc) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_find_entering: null argument");
    if (e->begin != 0 || e->end != e->d.search_arc_num) return mcf::fail(MCF_ERR_STATE, "sharded engine: use mcf_engine_find_entering_local / _sharded");
    Key k;
    int rc = local_search(e, &k);
    if (rc) return rc;
    resolve_key(e, k, found, arc, reduced_cost);
    return MCF_OK;
}

int mcf_engine_find_entering_local(mcf_engine *e, mcf_candidate *out)
{
    if (!e || !out) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_find_entering_local: null argument");
    Key k;
    int rc = local_search(e, &k);
    if (rc) return rc;
    out->reduced_cost = k.p == kNone ? 0 : k.c;
    out->pos = k.p;
    if (k.p == kNone) out->arc = -1;
    else if (e->d.rule == MCF_RULE_BEST_ELIGIBLE) out->arc = (int32_t)k.p;
    else { const int na = e->next_arc >= e->d.search_arc_num ? 0 : e->next_arc; out->arc = (int32_t)(((int64_t)k.p + na) % e->d.search_arc_num); }
    return MCF_OK;
}

int mcf_engine_resolve(mcf_engine *e, int32_t count, const mcf_candidate *all, int32_t *found, int32_t *arc, int64_t *reduced_cost)
{
    if (!e || count < 0 || (count && !all) || !found || !arc) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_resolve: bad arguments");
    const Key best = merge_candidates(e->d.rule, e->d.semantics, e->d.search_arc_num, e->block_size, e->next_arc, count, all);
    resolve_key(e, best, found, arc, reduced_cost);
    return MCF_OK;
}

int mcf_resolve_candidates(int32_t rule, int32_t semantics, int32_t search_arc_num, int32_t block_size, int32_t *next_arc,
                           int32_t count, const mcf_candidate *all, int32_t *found, int32_t *arc, int64_t *reduced_cost)
{
    if (!next_arc || count < 0 || (count && !all) || !found || !arc || rule < 0 || rule > 2 || search_arc_num < 0 ||
        (semantics != MCF_SEM_PLAIN && semantics != MCF_SEM_OPTIMIZED) || *next_arc < 0 || *next_arc > search_arc_num)
        return mcf::fail(MCF_ERR_INVALID, "mcf_resolve_candidates: bad arguments");
    const int B = block_size > 0 ? block_size : mcf::default_block_size(search_arc_num, semantics);
    const Key best = merge_candidates(rule, semantics, search_arc_num, B, *next_arc, count, all);
    int na = *next_arc;
    resolve_key_raw(rule, semantics, search_arc_num, B, na, best, found, arc, reduced_cost);
    *next_arc = na;
    return MCF_OK;
}

int mcf_engine_get_next_arc(mcf_engine *e, int32_t *next_arc) { if (!e || !next_arc) return mcf::fail(MCF_ERR_INVALID, "null argument"); *next_arc = e->next_arc; return MCF_OK; }
int mcf_engine_set_next_arc(mcf_engine *e, int32_t next_arc)
{
    if (!e || next_arc < 0 || next_arc > e->d.search_arc_num) return mcf::fail(MCF_ERR_INVALID, "next_arc out of range");
    e->next_arc = next_arc;
    return MCF_OK;
}
int mcf_engine_get_block_size(mcf_engine *e, int32_t *b) { if (!e || !b) return mcf::fail(MCF_ERR_INVALID, "null argument"); *b = e->block_size; return MCF_OK; }

int mcf_engine_download_pi(mcf_engine *e, int64_t *out)
{
    if (!e || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    int rc = flush_pending(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    const int n = e->d.node_count;
    if (e->d.int_width == 32) {
        std::vector<int32_t> t(n);
        HIP_TRY(hipMemcpy(t.data(), e->d_pi, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) out[i] = t[i];
    } else {
        HIP_TRY(hipMemcpy(out, e->d_pi, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
    }
    return MCF_OK;
}

int mcf_engine_download_state(mcf_engine *e, int8_t *out)
{
    if (!e || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->d.device));
    int rc = flush_pending(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(out + e->begin, e->d_state, e->end - e->begin, hipMemcpyDeviceToHost));
    return MCF_OK;
}

int mcf_engine_get_stats(mcf_engine *e, mcf_engine_stats *out)
{
    if (!e || !out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    (void)hipSetDevice(e->d.device);
    int rc = drain_events(e, true);
    if (rc) return rc;
    *out = e->st;
    return MCF_OK;
}

int mcf_engine_reset_stats(mcf_engine *e)
{
    if (!e) return mcf::fail(MCF_ERR_INVALID, "null argument");
    (void)hipSetDevice(e->d.device);
    drain_events(e, true);
    const mcf_engine_stats keep = e->st;
    e->st = mcf_engine_stats{};
    e->st.scan_workgroups = keep.scan_workgroups;
    e->st.scan_threads = keep.scan_threads;
    e->st.bytes_per_scan = keep.bytes_per_scan;
    return MCF_OK;
}

int mcf_engine_bench_scan(mcf_engine *e, int32_t reps, int32_t cold, int64_t flush_bytes, double *avg_ns, double *min_ns)
{
    if (!e || reps < 1 || !avg_ns || !min_ns) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_bench_scan: bad arguments");
    if (!e->uploaded) return mcf::fail(MCF_ERR_STATE, "mcf_engine_upload has not been called");
    HIP_TRY(hipSetDevice(e->d.device));
    int rc = flush_pending(e);
    if (rc) return rc;
    if (cold) {
        const size_t want = (size_t)std::max<int64_t>(flush_bytes, 1 << 20);
        if (e->flush_bytes < want) {
            (void)hipFree(e->d_flush);
            e->d_flush = nullptr;
            HIP_TRY(hipMalloc(&e->d_flush, want));
            HIP_TRY(hipMemset(e->d_flush, 0, want));
            e->flush_bytes = want;
        }
    }
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    double sum = 0, mn = 1e30;
    for (int r = 0; r < reps; ++r) {
        if (cold) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, e->stream, (uint4 *)e->d_flush, e->flush_bytes / 16);
        e->seq += 1;
        if (e->seq == 0) e->seq = 1;
        // same dispatch as a search, timed on the engine's stream; the records are simply not merged
        if (e->d.int_width == 32) { ScanParams<int32_t> p; fill_params(e, p, false); dispatch_scan<int32_t>(e, p, a, b); }
        else { ScanParams<int64_t> p; fill_params(e, p, false); dispatch_scan<int64_t>(e, p, a, b); }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventSynchronize(b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
        sum += ms * 1e6;
        mn = std::min(mn, (double)ms * 1e6);
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *avg_ns = sum / reps;
    *min_ns = mn;
    return MCF_OK;
}

// ------------------------------------------------------------------------------------------------ RCCL exchange

int mcf_comm_unique_id(uint8_t id_out[128])
{
    RcclApi *r = rccl();
    if (!r) return mcf::fail(MCF_ERR_COMM, "librccl.so could not be loaded");
    Id128 id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return mcf::fail(MCF_ERR_COMM, "ncclGetUniqueId: %s", r->error_string ? r->error_string(rc) : "error");
    memcpy(id_out, id.b, 128);
    return MCF_OK;
}

int mcf_engine_comm_init(mcf_engine *e, const uint8_t id[128], int32_t rank, int32_t world)
{
    if (!e || !id || world < 1 || rank < 0 || rank >= world) return mcf::fail(MCF_ERR_INVALID, "mcf_engine_comm_init: bad arguments");
    RcclApi *r = rccl();
    if (!r) return mcf::fail(MCF_ERR_COMM, "librccl.so could not be loaded");
    HIP_TRY(hipSetDevice(e->d.device));
    Id128 nid;
    memcpy(nid.b, id, 128);
    const int rc = r->comm_init_rank(&e->comm, world, nid, rank);
    if (rc != 0) return mcf::fail(MCF_ERR_COMM, "ncclCommInitRank: %s", r->error_string ? r->error_string(rc) : "error");
    e->rank = rank;
    e->world = world;
    HIP_TRY(hipMalloc((void **)&e->d_cand_local, sizeof(mcf_candidate)));
    HIP_TRY(hipMalloc((void **)&e->d_cand_all, sizeof(mcf_candidate) * world));
    HIP_TRY(hipHostMalloc((void **)&e->h_cand_all, sizeof(mcf_candidate) * world, hipHostMallocDefault));
    return MCF_OK;
}

int mcf_engine_find_entering_sharded(mcf_engine *e, int32_t *found, int32_t *arc, int64_t *reduced_cost)
{
    if (!e || !found || !arc) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (!e->comm) return mcf::fail(MCF_ERR_STATE, "mcf_engine_comm_init has not been called");
    mcf_candidate mine;
    int rc = mcf_engine_find_entering_local(e, &mine);
    if (rc) return rc;
    // MINLOC over (key, arc): RCCL has no MINLOC, so all-gather the 16-byte records and reduce locally
    HIP_TRY(hipMemcpyAsync(e->d_cand_local, &mine, sizeof(mine), hipMemcpyHostToDevice, e->stream));
    const int nrc = rccl()->all_gather(e->d_cand_local, e->d_cand_all, sizeof(mcf_candidate), kNcclChar, e->comm, e->stream);
    if (nrc != 0) return mcf::fail(MCF_ERR_COMM, "ncclAllGather: %s", rccl()->error_string ? rccl()->error_string(nrc) : "error");
    HIP_TRY(hipMemcpyAsync(e->h_cand_all, e->d_cand_all, sizeof(mcf_candidate) * e->world, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return mcf_engine_resolve(e, e->world, e->h_cand_all, found, arc, reduced_cost);
}

}  // extern "C"
