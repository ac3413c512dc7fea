// kernels.hip.h -- device code of the pivot engine (included by engine.hip only).
//
//   scan_kernel / scan_kernel_lds   one dispatch per search: stream the SoA arc arrays, two potential gathers per arc, exact argmin
//   resident_kernel                 one grid per solve: requests arrive through a mailbox in BAR-mapped VRAM; arcs (REG) and the potentials of
//                                   their end points (PIREG) live in registers, or all potentials in LDS (LPI); CAND adds a candidate list
//   update_kernel                   long patch lists in dispatch mode
//   scan_rc_kernel / update_rc_kernel  large sparse instances: reduced costs kept per arc, the scan streams 9 bytes per arc and gathers nothing
// PERM variants read arcs stored in bucketed order (sorted by target-node range) and break ties through the original arc ids.
// No MFMA: there is no contraction on this path; it is bound by memory bandwidth, gather throughput and host <-> device latency.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mcf_hip.h"

namespace {


constexpr int kThreads = 256;              // 4 wavefronts of 64
constexpr int kArcsPerThread = 4;          // 16-byte loads of source/target, 4-byte load of state
constexpr int kTile = kThreads * kArcsPerThread;   // 1024 arcs per workgroup per step
constexpr int kPad = 8 * kTile;            // device arrays are padded to this with state = 0 (two 4096-arc tiles of a 1024-thread workgroup)
constexpr int kInlinePi = 96;              // potentials patched through the kernel arguments
constexpr int kInlineState = 4;
constexpr int kMaxWorkgroups = 8192;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kSlotStride = 4;             // every workgroup answers with one whole 64-byte line (the record four times): partial-line
                                           // writes into host memory cost 1-2.5 us more per search (read-modify-write at the host bridge)
constexpr int kResidentThreads = 1024;     // 16 wavefronts per workgroup: few pollers, few records (resident and LDS-potential kernels)
constexpr int kResidentTile = kResidentThreads * kArcsPerThread;   // 4096 arcs per such workgroup
constexpr int kLdsPiMax = 16384;           // potentials kept in LDS when node_count fits (128 KB of int64): gathers leave the L1/TA path

// 16-byte answer of one workgroup, written with one store into pinned host memory.
// tag is last so that a host that sees the tag sees the payload (one PCIe write, ascending addresses).
struct alignas(16) Slot {
    int64_t c;
    uint32_t p;
    uint32_t tag;
};

// tag of a record = request number mixed with the payload: a host that reads a record whose halves belong to different writes
// (a torn 16-byte store) sees a tag that does not verify and keeps waiting
__host__ __device__ inline uint32_t record_tag(uint32_t seq, int64_t c, uint32_t p)
{
    const uint64_t u = (uint64_t)c;
    return seq ^ (uint32_t)u ^ (uint32_t)(u >> 32) ^ ((p << 13) | (p >> 19));
}

struct Key {
    int64_t c;
    uint32_t r;   // Block Search: rank of the block in scan order (doubled, +1 for the wrapped half in OPTIMIZED)
    uint32_t p;   // Best: arc index; First/Block: position in the cyclic scan that starts at next_arc
};
// OPTIMIZED Block Search answers with TWO keys (kDual): the block key above and the RANGE key (r = 0 for arcs of the first range
// [next_arc, m_s), 1 for the wrapped range [0, next_arc); then c, then the scan position) = the best arc of the first range that holds
// an eligible arc at all.  On a host with Vector.IsHardwareAccelerated the reference's scan does not stop at the block boundary when the
// hit falls into the "SIMD" part of the range (BSPO.cs:84-99 falls through with cnt == 0 and runs to the end of the range): the entering
// arc is then the range key's, and which of the two applies follows from the block key alone (engine.hip: resolve_key_raw).
template <int RULE, bool OPT> constexpr bool kDual = RULE == MCF_RULE_BLOCK_SEARCH && OPT;

template <typename T>
struct ScanParams {
    const int32_t *src;
    const int32_t *tgt;
    const T *cost;
    int8_t *state;
    T *pi;
    Slot *slots;
    const int32_t *orig;   // bucketed layout: global arc id of every local position (nullptr = arcs are stored in their own order)
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

// Cross-lane exchange without LDS traffic: DPP moves inside a row of 16 lanes (xor 1, xor 2, mirror within 8, mirror within 16 -- each an
// involution that pairs lanes holding different partial results), then the four row results are read with v_readlane.  A butterfly of
// ds_bpermute (what __shfl_xor compiles to) costs about five times as many cycles for the 64-lane argmin.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ int64_t dpp_i64(int64_t v)
{
    const uint32_t lo = dpp_u32<CTRL>((uint32_t)(uint64_t)v), hi = dpp_u32<CTRL>((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ int64_t lane_i64(int64_t v, int lane)
{
    const uint32_t lo = lane_u32((uint32_t)(uint64_t)v, lane), hi = lane_u32((uint32_t)((uint64_t)v >> 32), lane);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <int RULE, int CTRL>
__device__ __forceinline__ void wave_min_step(Key &k)
{
    const int64_t oc = dpp_i64<CTRL>(k.c);
    const uint32_t op = dpp_u32<CTRL>(k.p);
    const uint32_t orank = (RULE == MCF_RULE_BLOCK_SEARCH) ? dpp_u32<CTRL>(k.r) : 0u;
    take_if_better<RULE>(k, oc, orank, op);
}

// result valid in every lane (the row results are combined through scalar registers)
template <int RULE>
__device__ __forceinline__ Key wave_min(Key k)
{
    wave_min_step<RULE, kDppXor1>(k);
    wave_min_step<RULE, kDppXor2>(k);
    wave_min_step<RULE, kDppHalfMirror>(k);
    wave_min_step<RULE, kDppMirror>(k);
    Key r;
    r.c = lane_i64(k.c, 0);
    r.p = lane_u32(k.p, 0);
    r.r = (RULE == MCF_RULE_BLOCK_SEARCH) ? lane_u32(k.r, 0) : 0u;
#pragma unroll
    for (int row = 1; row < 4; ++row)
        take_if_better<RULE>(r, lane_i64(k.c, 16 * row), (RULE == MCF_RULE_BLOCK_SEARCH) ? lane_u32(k.r, 16 * row) : 0u, lane_u32(k.p, 16 * row));
    return r;
}

template <typename T> struct Vec4;
template <> struct Vec4<int32_t> {
    int32_t v[4];
    __device__ __forceinline__ void load(const int32_t *p) { const int4 a = *reinterpret_cast<const int4 *>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
    __device__ __forceinline__ void load_nt(const int32_t *p)
    {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i a = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
};
template <> struct Vec4<int64_t> {
    int64_t v[4];
    __device__ __forceinline__ void load(const int64_t *p)
    {
        const longlong2 a = *reinterpret_cast<const longlong2 *>(p), b = *reinterpret_cast<const longlong2 *>(p + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    __device__ __forceinline__ void load_nt(const int64_t *p)
    {
        typedef long v2l __attribute__((ext_vector_type(2)));
        const v2l a = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p + 2));
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
};

// One tile = 1024 consecutive arcs, 4 per thread.  load_tile issues all of a thread's streamed loads; eval_tile gathers the
// eight potentials and folds the four reduced costs into the running key.  (The resident kernel loads once and evaluates
// every pivot.)
template <typename T>
struct TileData {
    uint32_t st4;
    Vec4<int32_t> s, t;
    Vec4<T> c;
};

template <typename T, bool NT = false>
__device__ __forceinline__ void load_tile(const int32_t *src, const int32_t *tgt, const T *cost, const int8_t *state, int i0, TileData<T> &d)
{
    if (NT) {   // streamed once per scan: non-temporal, do not displace the potentials from L2
        d.st4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(state + i0));
        d.s.load_nt(src + i0);
        d.t.load_nt(tgt + i0);
        d.c.load_nt(cost + i0);
    } else {
        d.st4 = *reinterpret_cast<const uint32_t *>(state + i0);
        d.s.load(src + i0);
        d.t.load(tgt + i0);
        d.c.load(cost + i0);
    }
}

// the eight potential gathers of a thread's four arcs
template <typename T>
__device__ __forceinline__ void gather_tile(const TileData<T> &d, const T *pi, T ps[4], T pt[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { ps[j] = pi[d.s.v[j]]; pt[j] = pi[d.t.v[j]]; }
}

// folds the four reduced costs (potentials already in registers) into the running key
// PERM (Best Eligible only): the arcs are stored in a bucketed order (engine.hip: build_layout) and e0 counts POSITIONS; orig[position - base]
// is the arc's id, which is what breaks ties (NS.cs:1653: lowest arc index among equal reduced costs).  Positions keep the arcs' own
// order inside a bucket, so only a tie with the running best costs the two look-ups.
template <typename T, int RULE, bool OPT, bool PERM = false>
__device__ __forceinline__ void fold_tile(const TileData<T> &d, const T ps[4], const T pt[4], int e0, int m_s, int next_arc, int block_size, int rstar,
                                          Key &best, Key &range, const int32_t *orig = nullptr, int base = 0)
{
    uint32_t pos0 = 0;
    if (RULE != MCF_RULE_BEST_ELIGIBLE) {
        int q = e0 - next_arc;
        if (q < 0) q += m_s;
        pos0 = (uint32_t)q;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int st = (int)(int8_t)(d.st4 >> (8 * j));
        // 64-bit arithmetic in both widths: int32 inputs cannot overflow it
        const int64_t dd = (int64_t)d.c.v[j] + (int64_t)ps[j] - (int64_t)pt[j];
        const int64_t rc = st > 0 ? dd : (st < 0 ? -dd : 0);
        if (RULE == MCF_RULE_BEST_ELIGIBLE) {
            if (rc < best.c) { best.c = rc; best.p = (uint32_t)(e0 + j); }   // strict <: lowest arc wins ties
            else if (PERM && rc == best.c && rc < 0 && orig[e0 + j - base] < orig[(int)best.p - base]) best.p = (uint32_t)(e0 + j);
        } else {
            uint32_t pos = pos0 + j;   // the group may straddle the wrap point
            if (pos >= (uint32_t)m_s) pos -= (uint32_t)m_s;
            if (RULE == MCF_RULE_FIRST_ELIGIBLE) {
                if (rc < 0) take_if_better<RULE>(best, rc, 0u, pos);
            } else {
                uint32_t r = pos / (uint32_t)block_size;
                r = 2 * r + ((OPT && (int)r == rstar && e0 + j < next_arc) ? 1u : 0u);
                if (rc < 0) {
                    take_if_better<RULE>(best, rc, r, pos);
                    if (OPT) take_if_better<RULE>(range, rc, e0 + j < next_arc ? 1u : 0u, pos);
                }
            }
        }
    }
}

template <typename T, int RULE, bool OPT, bool PERM = false>
__device__ __forceinline__ void eval_tile(const TileData<T> &d, const T *pi, int e0, int m_s, int next_arc, int block_size, int rstar, Key &best, Key &range,
                                          int sub_node = -1, T sub_val = 0, const int32_t *orig = nullptr, int base = 0)
{
    T ps[4], pt[4];
    gather_tile<T>(d, pi, ps, pt);
    // a single patched potential whose store may still be in flight (resident fast path): take its value from the request
#pragma unroll
    for (int j = 0; j < 4; ++j) { ps[j] = d.s.v[j] == sub_node ? sub_val : ps[j]; pt[j] = d.t.v[j] == sub_node ? sub_val : pt[j]; }
    fold_tile<T, RULE, OPT, PERM>(d, ps, pt, e0, m_s, next_arc, block_size, rstar, best, range, orig, base);
}

template <typename T, int RULE, bool OPT, bool NT = false, bool PERM = false>
__device__ __forceinline__ void scan_tile(const ScanParams<T> &p, int i0, Key &best, Key &range)
{
    TileData<T> d;
    load_tile<T, NT>(p.src, p.tgt, p.cost, p.state, i0, d);
    eval_tile<T, RULE, OPT, PERM>(d, p.pi, p.base + i0, p.m_s, p.next_arc, p.block_size, p.rstar, best, range, -1, (T)0, p.orig, p.base);
}

// workgroup-level finish: wave butterfly -> LDS -> one 16-byte record.  SYSTEM = write-through store for kernels that keep running.
// DUAL (OPTIMIZED Block Search): the range key is folded the same way and fills the second half of the line (records 2 and 3).
template <int RULE, bool SYSTEM, int NT = kThreads, bool DUAL = false>
__device__ __forceinline__ void publish_best(Key best, Slot *slot, uint32_t tag, bool full_line = false, Key range = Key{0, kNone, kNone})
{
    const int tid = threadIdx.x;
    best = wave_min<RULE>(best);
    if (DUAL) range = wave_min<RULE>(range);
    __shared__ Key wave_best[NT / 64];
    __shared__ Key wave_range[DUAL ? NT / 64 : 1];
    if ((tid & 63) == 0) { wave_best[tid >> 6] = best; if (DUAL) wave_range[tid >> 6] = range; }
    __syncthreads();
    if (tid < 64) {
        // second stage: lane w takes wave w's result (at most 16 waves: one row of lanes), the same four row steps fold them
        const int waves = (int)(blockDim.x >> 6);          // NT is the largest block the kernel is launched with
        const int w = tid < waves ? tid : 0;
        Key k;
        k.c = tid < waves ? wave_best[w].c : 0;
        k.r = tid < waves ? wave_best[w].r : kNone;
        k.p = tid < waves ? wave_best[w].p : kNone;
        wave_min_step<RULE, kDppXor1>(k);
        wave_min_step<RULE, kDppXor2>(k);
        wave_min_step<RULE, kDppHalfMirror>(k);
        wave_min_step<RULE, kDppMirror>(k);
        k.c = lane_i64(k.c, 0);
        k.p = lane_u32(k.p, 0);
        k.r = lane_u32(k.r, 0);
        if (DUAL) {
            Key q;
            q.c = tid < waves ? wave_range[w].c : 0;
            q.r = tid < waves ? wave_range[w].r : kNone;
            q.p = tid < waves ? wave_range[w].p : kNone;
            wave_min_step<RULE, kDppXor1>(q);
            wave_min_step<RULE, kDppXor2>(q);
            wave_min_step<RULE, kDppHalfMirror>(q);
            wave_min_step<RULE, kDppMirror>(q);
            q.c = lane_i64(q.c, 0);
            q.p = lane_u32(q.p, 0);
            if (tid >= 2) { k.c = q.c; k.p = q.p; }        // lanes 2 and 3 carry the range key
        }
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        v4u out;
        out.x = (uint32_t)(uint64_t)k.c;
        out.y = (uint32_t)((uint64_t)k.c >> 32);
        out.z = k.p;
        out.w = record_tag(tag, k.c, k.p);
        // full_line: lanes 0..3 write four records = one whole 64-byte line (no partial-line write on the host side): the record four
        // times, or (DUAL) the block key twice and the range key twice
        const int lanes = (full_line || DUAL) ? 4 : 1;
        if (tid < lanes) {
            if (SYSTEM) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot + tid), "v"(out) : "memory");
            else *reinterpret_cast<v4u *>(slot + tid) = out;
        }
    }
}

template <typename T, int RULE, bool OPT, int UNROLL, bool NT = false, bool PERM = false>
__global__ __launch_bounds__(kThreads) void scan_kernel(const ScanParams<T> p)
{
    static_assert(!PERM || RULE == MCF_RULE_BEST_ELIGIBLE, "the bucketed layout serves Best Eligible only");
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

    Key best{0, kNone, kNone}, range{0, kNone, kNone};
    const int step = gridDim.x * kTile * UNROLL;
    for (int i0 = blockIdx.x * kTile * UNROLL + tid * kArcsPerThread; i0 < p.count_padded; i0 += step) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) scan_tile<T, RULE, OPT, NT, PERM>(p, i0 + u * kTile, best, range);
    }
    if (PERM && best.p != kNone) best.p = (uint32_t)p.orig[(int)best.p - p.base];     // position -> arc id; from here on everything is as without PERM
    publish_best<RULE, false, kThreads, kDual<RULE, OPT>>(best, p.slots + (size_t)blockIdx.x * kSlotStride, p.seq, true, range);
}

// ------------------------------------------------------------------------------------------------ reduced costs kept per arc (RC layout)
// Large sparse instances (arcs in neither registers nor LDS: config 5) spend their scan on the two potential gathers per arc: 16 useful
// bytes fetched as two whole cache lines from an 8 MB table, L2 -> L1 bandwidth bound (27-33 % of the HBM rate for the 17 streamed bytes).
// In this layout the engine keeps d[e] = cost[e] + pi[source[e]] - pi[target[e]] per arc and the scan streams state (1 B) + d (8 B): no
// gather at all.  The gathers move to the potential update, where they are scatters over the MOVED nodes' arcs only: update_rc_kernel
// adds the node's change to d of its out-arcs and subtracts it from d of its in-arcs (integer atomics: exact in any order; an arc inside
// the moved subtree gets +delta and -delta).  Same reduced costs, bit for bit, same keys, same tie-breaks (arcs keep their own order).
template <int RULE, bool OPT>
__device__ __forceinline__ void fold_rc(uint32_t st4, const int64_t d[4], int e0, int m_s, int next_arc, int block_size, int rstar, Key &best, Key &range)
{
    uint32_t pos0 = 0;
    if (RULE != MCF_RULE_BEST_ELIGIBLE) {
        int q = e0 - next_arc;
        if (q < 0) q += m_s;
        pos0 = (uint32_t)q;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int st = (int)(int8_t)(st4 >> (8 * j));
        const int64_t rc = st > 0 ? d[j] : (st < 0 ? -d[j] : 0);
        if (RULE == MCF_RULE_BEST_ELIGIBLE) {
            if (rc < best.c) { best.c = rc; best.p = (uint32_t)(e0 + j); }   // strict <: lowest arc wins ties
        } else {
            uint32_t pos = pos0 + j;
            if (pos >= (uint32_t)m_s) pos -= (uint32_t)m_s;
            if (RULE == MCF_RULE_FIRST_ELIGIBLE) {
                if (rc < 0) take_if_better<RULE>(best, rc, 0u, pos);
            } else {
                uint32_t r = pos / (uint32_t)block_size;
                r = 2 * r + ((OPT && (int)r == rstar && e0 + j < next_arc) ? 1u : 0u);
                if (rc < 0) {
                    take_if_better<RULE>(best, rc, r, pos);
                    if (OPT) take_if_better<RULE>(range, rc, e0 + j < next_arc ? 1u : 0u, pos);
                }
            }
        }
    }
}

constexpr int kRcInlineNodes = 64;          // moved nodes whose change rides in the scan's arguments ...
constexpr int kRcInlineEntries = 4096;      // ... when their arc lists together are no longer than this

struct RcParams {
    const int8_t *state_ro;
    int8_t *state;
    int64_t *rc;
    Slot *slots;
    void *pi;                               // device potentials (kept for download / later list updates): workgroup 0 adds the change
    const uint32_t *adj;
    int32_t base, count_padded, m_s, next_arc, block_size, rstar;
    uint32_t seq;
    int32_t n_st, n_pi, narrow;             // narrow: pi is int32
    int32_t st_arc[kInlineState];
    int32_t st_val[kInlineState];
    int64_t sigma;                          // every moved node's potential changes by this (NS.cs:1185-1209: one sigma per pivot)
    int32_t pi_node[kRcInlineNodes];
    int32_t adj_lo[kRcInlineNodes];         // first entry of the node's arc list
    int32_t prefix[kRcInlineNodes + 1];     // entries before node k's list in the concatenation of the lists
};

template <int RULE, bool OPT, int UNROLL>
__global__ __launch_bounds__(kResidentThreads) void scan_rc_kernel(const RcParams p)
{
    const int tid = threadIdx.x, nt = (int)blockDim.x, tile = nt * kArcsPerThread;      // 256 .. 1024 threads per workgroup (the host's choice)
    if (p.n_st | p.n_pi) {
        // the State[] writes of the previous pivot: final values, applied by EVERY workgroup before it reads
        if (tid < p.n_st) {
            const int a = p.st_arc[tid] - p.base;
            if ((unsigned)a < (unsigned)p.count_padded) p.state[a] = (int8_t)p.st_val[tid];
        }
        // A short potential list: every workgroup walks the moved nodes' arc lists and shifts the reduced costs of the arcs IT scans
        // (tile t belongs to workgroup t mod grid), so every arc is shifted exactly once, by its only reader, before that reader reads
        // it -- no ordering between workgroups is needed.  An arc between two moved nodes appears twice (+sigma, -sigma): atomics.
        const int total = p.n_pi ? p.prefix[p.n_pi] : 0;
        for (int f = tid; f < total; f += nt) {
            int k = 0;
            for (int step = kRcInlineNodes / 2; step > 0; step >>= 1)        // largest k with prefix[k] <= f
                if (k + step < p.n_pi && p.prefix[k + step] <= f) k += step;
            const uint32_t x = p.adj[p.adj_lo[k] + (f - p.prefix[k])];
            const uint32_t pos = x & 0x7FFFFFFFu;
            if ((int)((pos / (uint32_t)(tile * UNROLL)) % gridDim.x) == (int)blockIdx.x) {
                const long long add = (x >> 31) ? -p.sigma : p.sigma;
                atomicAdd(reinterpret_cast<unsigned long long *>(p.rc + pos), (unsigned long long)add);
            }
        }
        if (blockIdx.x == 0 && tid < p.n_pi) {
            if (p.narrow) reinterpret_cast<int32_t *>(p.pi)[p.pi_node[tid]] += (int32_t)p.sigma;
            else reinterpret_cast<int64_t *>(p.pi)[p.pi_node[tid]] += p.sigma;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    Key best{0, kNone, kNone}, range{0, kNone, kNone};
    typedef long v2l __attribute__((ext_vector_type(2)));
    const int step = gridDim.x * tile * UNROLL;
    for (int i0 = blockIdx.x * tile * UNROLL + tid * kArcsPerThread; i0 < p.count_padded; i0 += step) {
        uint32_t st4[UNROLL];
        v2l a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {       // streamed once per scan: non-temporal, all loads of the trip in flight together
            const int i = i0 + u * tile;
            st4[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p.state_ro + i));
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p.rc + i));
            b[u] = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p.rc + i + 2));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t d[4] = {a[u].x, a[u].y, b[u].x, b[u].y};
            fold_rc<RULE, OPT>(st4[u], d, p.base + i0 + u * tile, p.m_s, p.next_arc, p.block_size, p.rstar, best, range);
        }
    }
    publish_best<RULE, false, kResidentThreads, kDual<RULE, OPT>>(best, p.slots + (size_t)blockIdx.x * kSlotStride, p.seq, true, range);
}

// d[e] = cost[e] + pi[source[e]] - pi[target[e]] for every stored arc (upload, mcf_engine_patch_arcs); padding arcs have state 0
template <typename T>
__global__ __launch_bounds__(kThreads) void rc_init_kernel(const int32_t *src, const int32_t *tgt, const T *cost, const T *pi, int64_t *rc, int count_padded)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < count_padded) rc[i] = (int64_t)cost[i] + (int64_t)pi[src[i]] - (int64_t)pi[tgt[i]];
}

// pi[node[i]] = value[i] and the change carried over to the node's arcs; state[arc[j]] = s[j].  One thread per node (a hub's arc list is
// walked by its whole workgroup): every node's change is computed and applied exactly once (a node named twice carries the same value both times).
// adj: the shard's arcs incident to each node as local positions, bit 31 set when the node is the arc's TARGET.
constexpr int kRcHeavyDegree = 512;         // a node with a longer arc list is shifted by its whole workgroup, not by one thread

template <typename T>
__global__ __launch_bounds__(kThreads) void update_rc_kernel(T *pi, const int32_t *nodes, const int64_t *values, int n_pi, int8_t *state, const int32_t *arcs,
                                                             const int32_t *states, int n_st, int base, int count_padded, int64_t *rc,
                                                             const int32_t *adj_start, const uint32_t *adj)
{
    __shared__ int32_t heavy_lo[kThreads], heavy_hi[kThreads];
    __shared__ int64_t heavy_delta[kThreads];
    __shared__ int heavy_n;
    const int tid = threadIdx.x, i = blockIdx.x * kThreads + tid;
    if (tid == 0) heavy_n = 0;
    __syncthreads();
    if (i < n_pi) {
        const int u = nodes[i];
        // exchange, not read + write: a list of the candidate cache may name a node twice (with the same value) -- the second entry then
        // finds the value in place and shifts nothing
        int64_t old;
        if (sizeof(T) == 4) old = (int64_t)atomicExch(reinterpret_cast<int *>(pi) + u, (int)values[i]);
        else old = (int64_t)atomicExch(reinterpret_cast<unsigned long long *>(pi) + u, (unsigned long long)values[i]);
        const int64_t delta = values[i] - old;
        if (delta != 0) {
            const int lo = adj_start[u], hi = adj_start[u + 1];
            if (hi - lo > kRcHeavyDegree) {                 // a hub: left to the whole workgroup below
                const int q = atomicAdd(&heavy_n, 1);
                heavy_lo[q] = lo; heavy_hi[q] = hi; heavy_delta[q] = delta;
            } else {
                for (int k = lo; k < hi; ++k) {
                    const uint32_t x = adj[k];
                    const long long add = (x >> 31) ? -delta : delta;
                    atomicAdd(reinterpret_cast<unsigned long long *>(rc + (x & 0x7FFFFFFFu)), (unsigned long long)add);
                }
            }
        }
    }
    if (i < n_st) {
        const int a = arcs[i] - base;
        if ((unsigned)a < (unsigned)count_padded) state[a] = (int8_t)states[i];
    }
    __syncthreads();
    for (int q = 0; q < heavy_n; ++q) {
        const int64_t delta = heavy_delta[q];
        for (int k = heavy_lo[q] + tid; k < heavy_hi[q]; k += kThreads) {
            const uint32_t x = adj[k];
            const long long add = (x >> 31) ? -delta : delta;
            atomicAdd(reinterpret_cast<unsigned long long *>(rc + (x & 0x7FFFFFFFu)), (unsigned long long)add);
        }
    }
}

// Same scan for graphs whose potential vector fits LDS (node_count <= kLdsPiMax): 1024-thread workgroups copy pi into LDS once
// and loop over 4096-arc tiles; the two gathers per arc become ds_read instead of divergent vector-memory loads, which
// otherwise cap the scan at about one lane per clock per CU even when every gather hits L1.
template <typename T, int RULE, bool OPT, int UNROLL>
__global__ __launch_bounds__(kResidentThreads) void scan_kernel_lds(const ScanParams<T> p, int n_nodes)
{
    __shared__ __attribute__((aligned(16))) T lpi[kLdsPiMax];
    const int tid = threadIdx.x;
    if (p.n_pi | p.n_st) {
        if (tid < p.n_pi) p.pi[p.pi_node[tid]] = p.pi_val[tid];
        if (tid >= kResidentThreads - kInlineState && tid - (kResidentThreads - kInlineState) < p.n_st) {
            const int k = tid - (kResidentThreads - kInlineState);
            const int a = p.st_arc[k] - p.base;
            if ((unsigned)a < (unsigned)p.count_padded) p.state[a] = (int8_t)p.st_val[k];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    for (int i = tid; i < n_nodes; i += kResidentThreads) lpi[i] = p.pi[i];
    __syncthreads();
    Key best{0, kNone, kNone}, range{0, kNone, kNone};
    const int step = gridDim.x * UNROLL * kResidentTile;     // UNROLL tiles per step: all their streamed loads are in flight together
    for (int i0 = blockIdx.x * UNROLL * kResidentTile + tid * kArcsPerThread; i0 < p.count_padded; i0 += step) {
        TileData<T> d[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) load_tile<T>(p.src, p.tgt, p.cost, p.state, i0 + u * kResidentTile, d[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) eval_tile<T, RULE, OPT>(d[u], lpi, p.base + i0 + u * kResidentTile, p.m_s, p.next_arc, p.block_size, p.rstar, best, range);
    }
    publish_best<RULE, false, kResidentThreads, kDual<RULE, OPT>>(best, p.slots + (size_t)blockIdx.x * kSlotStride, p.seq, true, range);
}

// ------------------------------------------------------------------------------------------------ resident mode
// One dispatch per pivot costs ~6.5 us of launch + dispatch latency before any arc is read (profiles/r01_dispatch_floor_kfloor.txt).
// resident_kernel is launched once and then serves one request per pivot through a MAILBOX in fine-grained VRAM that the host
// writes through the PCIe BAR: the host posts {seq, next_arc, patches}, every workgroup sees the new seq by polling device
// memory, applies the patches, scans the tile(s) it owns and answers with its 16-byte record in pinned host memory.
// Workgroups that own a single tile keep its arcs in registers (REG), so a request costs two potential gathers per arc and nothing else.
//
constexpr int kMailboxLines = 256;                // lines staged in LDS at a time: 16 KB = line 0 + a chunk of 255 patch lines (1275 patches)
constexpr int kMailboxPatchesPerLine = 5;
constexpr int kMaxReplicas = 16;                  // copies of the poll unit (lines 0 and 1), 4 KB apart, so that 256 pollers do not hammer one address
constexpr int kReplicaStride = 1024;              // dwords
constexpr int kMailboxTail = kMaxReplicas * kReplicaStride;   // dword offset of line 2

template <typename T>
struct ResidentParams {
    const int32_t *src;
    const int32_t *tgt;
    const T *cost;
    int8_t *state;
    T *pi;
    Slot *slots;
    const int32_t *orig;        // bucketed layout (see ScanParams), only read by the PERM variant
    const uint32_t *mailbox;    // fine-grained VRAM, written by the host through the BAR
    uint32_t *exit_word;        // pinned host memory: [0] exit code, [1] requests served, [2..3] scan ticks of workgroup 0
    int32_t base, count_padded, m_s;
    uint32_t start_seq, idle_ticks;
    int32_t n_nodes;            // length of pi
    int32_t max_pi;             // potential patches the mailbox can hold
    int32_t max_st;             // state patches it can hold
    int32_t poll_replicas, poll_sleep;
    const int64_t *host_pi;     // resident_cand_kernel: the caller's bound potentials in mapped host memory (cmd 3 reloads from them), or null
    uint32_t *barrier;          // ... and the arrival counter of its grid-wide barrier (zero at launch)
};

__device__ __forceinline__ void resident_exit(uint32_t *exit_word, uint32_t code, uint32_t served, uint64_t scan_ticks)
{
    __hip_atomic_store(exit_word + 1, served, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(exit_word + 2, (uint32_t)scan_ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(exit_word + 3, (uint32_t)(scan_ticks >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(exit_word, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (c, p) lexicographic; {0, kNone} = "none" and loses against every eligible arc
struct Cand { int64_t c; uint32_t p; };
__device__ __forceinline__ bool cand_less(int64_t c1, uint32_t p1, int64_t c2, uint32_t p2) { return c1 < c2 || (c1 == c2 && p1 < p2); }
template <int CTRL>
__device__ __forceinline__ void cand_min_step(int64_t &c, uint32_t &p)
{
    const int64_t oc = dpp_i64<CTRL>(c);
    const uint32_t op = dpp_u32<CTRL>(p);
    const bool take = cand_less(oc, op, c, p);
    c = take ? oc : c;
    p = take ? op : p;
}
__device__ __forceinline__ void cand_wave_min(int64_t &c, uint32_t &p)
{
    cand_min_step<kDppXor1>(c, p);
    cand_min_step<kDppXor2>(c, p);
    cand_min_step<kDppHalfMirror>(c, p);
    cand_min_step<kDppMirror>(c, p);
    int64_t rc = lane_i64(c, 0);
    uint32_t rp = lane_u32(p, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const int64_t oc = lane_i64(c, 16 * row);
        const uint32_t op = lane_u32(p, 16 * row);
        const bool take = cand_less(oc, op, rc, rp);
        rc = take ? oc : rc;
        rp = take ? op : rp;
    }
    c = rc;
    p = rp;
}
__device__ __forceinline__ void store_record_system(Slot *slot, int64_t c, uint32_t p, uint32_t tag)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u out;
    out.x = (uint32_t)(uint64_t)c;
    out.y = (uint32_t)((uint64_t)c >> 32);
    out.z = p;
    out.w = record_tag(tag, c, p);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot), "v"(out) : "memory");
}

// Best Eligible with candidates (CAND): besides its best arc a workgroup reports up to two more candidates and a THRESHOLD
// = the smallest key among all of its eligible arcs that it does NOT report.  The host can then serve the following pivots
// from the candidate list as long as the answer provably is on it (mcf_engine.cand_* in the host code) -- same pivots, fewer
// round trips.  Per thread: best and second best of its 4 arcs; per wave: two butterflies give the wave's best and its exact
// second best; wave 0 (one lane per wave) takes the kCandPerGroup best of the wave winners out with three more butterflies and folds
// everything else into the threshold with a fourth.
constexpr int kCandPerGroup = 3;
constexpr int kCandRecords = 4;        // records per workgroup in candidate mode: 3 candidates + the threshold = one whole 64-byte line

// best and second best of a tile's four arcs, potentials already in registers
template <typename T>
__device__ __forceinline__ void fold_tile_best2(const TileData<T> &d, const T ps[4], const T pt[4], int e0, int64_t &c1, uint32_t &p1, int64_t &c2, uint32_t &p2)
{
    c1 = 0; p1 = kNone; c2 = 0; p2 = kNone;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int st = (int)(int8_t)(d.st4 >> (8 * j));
        const int64_t dd = (int64_t)d.c.v[j] + (int64_t)ps[j] - (int64_t)pt[j];
        const int64_t rc = st > 0 ? dd : (st < 0 ? -dd : 0);
        const uint32_t e = (uint32_t)(e0 + j);
        // the arcs come in increasing id and c1, c2 start at 0 ("none"): an arc is better exactly when its key is strictly smaller
        // (eligible <=> rc < 0, and an equal key keeps the lower id that is already there) -- two 64-bit compares per arc
        const bool b1 = rc < c1;
        const bool b2 = !b1 && rc < c2;
        // new best pushes the old best down to second
        c2 = b1 ? c1 : (b2 ? rc : c2);
        p2 = b1 ? p1 : (b2 ? e : p2);
        c1 = b1 ? rc : c1;
        p1 = b1 ? e : p1;
    }
}

template <typename T>
__device__ __forceinline__ void eval_tile_best2(const TileData<T> &d, const T *pi, int e0, int64_t &c1, uint32_t &p1, int64_t &c2, uint32_t &p2,
                                                int sub_node, T sub_val)
{
    T ps[4], pt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ps[j] = pi[d.s.v[j]]; pt[j] = pi[d.t.v[j]]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { ps[j] = d.s.v[j] == sub_node ? sub_val : ps[j]; pt[j] = d.t.v[j] == sub_node ? sub_val : pt[j]; }
    fold_tile_best2<T>(d, ps, pt, e0, c1, p1, c2, p2);
}

__device__ __forceinline__ void publish_candidates(int64_t c1, uint32_t p1, int64_t c2, uint32_t p2, Slot *slots, uint32_t tag)
{
    const int tid = threadIdx.x;
    __shared__ Cand wave_first[kResidentThreads / 64], wave_second[kResidentThreads / 64];
    int64_t wc = c1;
    uint32_t wp = p1;
    cand_wave_min(wc, wp);                                   // the wave's best
    int64_t sc = (p1 == wp && p1 != kNone) ? c2 : c1;        // the winner lane now offers its second best
    uint32_t sp = (p1 == wp && p1 != kNone) ? p2 : p1;
    cand_wave_min(sc, sp);                                   // the wave's exact second best
    if ((tid & 63) == 0) { wave_first[tid >> 6] = Cand{wc, wp}; wave_second[tid >> 6] = Cand{sc, sp}; }
    __syncthreads();
    // Wave 0 folds the waves' records, one lane per wave: three butterflies take the kCandPerGroup best of the wave winners out one after the
    // other, a fourth gives the threshold = the smallest key that is not reported (the winners left over and every wave's second best).
    // (One thread walking the 16 waves' records took 4-5 us of every device search: a dependent chain of a thousand instructions on one lane.)
    if (tid < 64) {
        const int waves = (int)(blockDim.x >> 6);
        int64_t fc = 0, tc = 0;
        uint32_t fp = kNone, tp = kNone;
        if (tid < waves) { fc = wave_first[tid].c; fp = wave_first[tid].p; tc = wave_second[tid].c; tp = wave_second[tid].p; }
        int64_t kc[kCandPerGroup];
        uint32_t kp[kCandPerGroup];
#pragma unroll
        for (int k = 0; k < kCandPerGroup; ++k) {
            int64_t mc = fc;
            uint32_t mp = fp;
            cand_wave_min(mc, mp);
            kc[k] = mc;
            kp[k] = mp;
            const bool mine = mp != kNone && fp == mp;       // arc ids are unique: exactly one lane holds the winner
            fc = mine ? 0 : fc;
            fp = mine ? kNone : fp;
        }
        if (cand_less(fc, fp, tc, tp)) { tc = fc; tp = fp; }
        cand_wave_min(tc, tp);
        // kCandRecords lanes write the records = one whole 64-byte line; the threshold fills the rest of the line
        if (tid < kCandRecords) {
            int64_t oc = tc;
            uint32_t op = tp;
#pragma unroll
            for (int k = 0; k < kCandPerGroup; ++k) { oc = tid == k ? kc[k] : oc; op = tid == k ? kp[k] : op; }
            store_record_system(slots + (size_t)blockIdx.x * kCandRecords + tid, oc, op, tag);
        }
    }
}

// Mailbox: lines of 64 bytes; dword 15 of EVERY line repeats seq, so a torn read of any line is detected and retried.
//   line 0      [0] seq [1] cmd (0 scan, 1 quit, 2 apply) [2] next_arc [3] rstar [4] n_pi [5] n_st [6..9] state patches 0,1 {arc, value}
//               [10..12] potential patch 0 {node, lo, hi} [13] scan: block size of THIS search (Block Search; the reference's adaptive
//               rule changes it between searches, NS.cs:1400-1438) / apply: entry lines valid so far [14] apply: post counter
// cmd 2 ("apply") streams a long potential list while the host is still walking the subtree: the entry lines 1..[13] of the NEXT scan
// request are in place, every workgroup applies those it has not applied yet and goes back to polling (no answer).  The posts are
// cumulative, so one that is overwritten before a workgroup saw it loses nothing; the scan request finishes the list.
//   line 1..    five entries {a, b, c} each: potential patches 1..n_pi-1 {node, lo, hi}, then state patches 2..n_st-1 {arc, value, 0}
// The poll reads line 0 only (one 64-byte read per workgroup per poll); the other lines are fetched when there are more entries.
//
// PIREG (register-resident arcs, potentials not in LDS, at most kPiRegThreads threads; with or without the candidate list): a thread also keeps the potentials of its arcs' end points in
// registers and PATCHES them instead of gathering them again for every request -- the eight divergent gathers per thread cost
// ~1.5 us per request on a CU (one lane per clock) while three out of four pivots move five nodes or fewer.  Lists of up to
// kPiRegCompare entries are compared against the eight end points directly (LDS broadcast reads); up to one chunk goes through a
// bitmap in LDS and only the lanes that hit gather again (up to kPiRegBitmapMax entries); longer lists gather everything again.
constexpr int kPiRegCompare = 20;                 // entries beyond the header patch that are matched by direct comparison (4 lines)
constexpr int kPiRegBitmapMax = 4096;             // longer lists: most lanes would hit anyway
constexpr int kPiRegBitmapWords = 4096;           // 131072 bits, indexed by node mod 131072 (aliases only cost a spurious gather)

constexpr int kPiRegThreads = 512;                // PIREG workgroups: at most 8 waves, so a thread may use up to 256 registers (it needs ~170)

template <typename T, int RULE, bool OPT, bool REG, bool LPI, bool CAND, bool PIREG = false, bool PERM = false>
__global__ __launch_bounds__(PIREG ? kPiRegThreads : kResidentThreads) void resident_kernel(const ResidentParams<T> p)
{
    static_assert(!PERM || (RULE == MCF_RULE_BEST_ELIGIBLE && !REG && !LPI && !CAND), "the bucketed layout serves the tile loop of Best Eligible");
    static_assert(!PIREG || (REG && !LPI), "PIREG needs register-resident arcs and global potentials");
    // mailbox staging: line 0 + one chunk of patch lines: 16 KB next to LDS-resident potentials, else 32 KB (2555 entries per chunk);
    // kept small so that several resident grids (independent solves) can share a CU
    constexpr int kLines = LPI ? kMailboxLines : 2 * kMailboxLines, kChunk = kLines - 1;
    __shared__ __attribute__((aligned(16))) uint32_t lm[kLines * 16];
    __shared__ __attribute__((aligned(16))) T lpi[LPI ? kLdsPiMax : 2];      // LPI: the whole potential vector lives here
    __shared__ uint32_t s_timeout;
    __shared__ uint32_t bitmap[PIREG ? kPiRegBitmapWords : 1];
    const int tid = threadIdx.x, nt = (int)blockDim.x;        // 64..1024 threads: the host sizes the grid so that every CU gets a workgroup
    const int my_i0 = blockIdx.x * nt * kArcsPerThread + tid * kArcsPerThread;
    TileData<T> mine;
    if (REG) load_tile<T>(p.src, p.tgt, p.cost, p.state, my_i0, mine);
    T ps[4], pt[4];                                           // PIREG: pi[source], pi[target] of my four arcs, kept current by the patches
    if (PIREG) {
        gather_tile<T>(mine, p.pi, ps, pt);
        for (int i = tid; i < kPiRegBitmapWords; i += nt) bitmap[i] = 0u;
        __syncthreads();
    }
    if (LPI) {
        for (int i = tid; i < p.n_nodes; i += nt) lpi[i] = p.pi[i];
        __syncthreads();
    }
    const T *const pi_view = LPI ? lpi : p.pi;
    uint32_t last = p.start_seq, served = 0, last_sub = 0;
    int applied = 0;                                          // entry lines of the coming scan request already applied (cmd 2)
    uint64_t scan_ticks = 0;
    uint64_t idle_since = __builtin_amdgcn_s_memrealtime();
    const uint32_t *const my_unit = p.mailbox + (size_t)(blockIdx.x % p.poll_replicas) * kReplicaStride;   // this workgroup's copy of lines 0, 1
    for (;;) {
        // ---- wait for a request: wave 0 polls lines 0 and 1 of the mailbox (128 bytes, system-scope reads of device memory) in a tight
        // loop, the other waves sleep at the barrier
        if (tid < 64) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            v4u x = v4u{0u, 0u, 0u, 0u};
            uint32_t flag;
            for (;;) {
                if (tid < 8) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(my_unit + tid * 4) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                const uint32_t seq0 = lane_u32(x[0], 0), tag0 = lane_u32(x[3], 3), cmd0 = lane_u32(x[1], 0), sub0 = lane_u32(x[2], 3);
                if (seq0 != last && tag0 == seq0 && (cmd0 != 2u || sub0 != last_sub)) { flag = 1u; break; }
                if (__builtin_amdgcn_s_memrealtime() - idle_since > p.idle_ticks) { flag = 2u; break; }
                for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
            }
            if (tid < 8) *reinterpret_cast<v4u *>(lm + tid * 4) = x;
            if (tid == 0) s_timeout = flag;
        }
        __syncthreads();
        const uint32_t seq = lm[0];
        int n_pi = (int)lm[4], n_st = (int)lm[5];
        n_pi = n_pi < 0 ? 0 : (n_pi > p.max_pi ? p.max_pi : n_pi);
        n_st = n_st < 0 ? 0 : (n_st > p.max_st ? p.max_st : n_st);
        const int extra_pi = n_pi > 1 ? n_pi - 1 : 0, extra_st = n_st > 2 ? n_st - 2 : 0;
        const int entries = extra_pi + extra_st;
        const uint32_t cmd = lm[1];
        const bool apply_only = cmd == 2u;
        const uint32_t sub = lm[14];
        int upto = (int)lm[13];                              // apply: the last entry line in place
        upto = upto < 0 ? 0 : (upto > (p.max_pi + p.max_st) / kMailboxPatchesPerLine ? (p.max_pi + p.max_st) / kMailboxPatchesPerLine : upto);
        const int lines = apply_only ? 1 + upto : 1 + (entries + kMailboxPatchesPerLine - 1) / kMailboxPatchesPerLine;
        const bool timed_out = s_timeout == 2u;
        const bool line1_staged = lm[31] == seq;           // line 1 came along with the poll and is complete
        const int next_arc = (int)lm[2], rstar = (int)lm[3];
        const int block_size = (int)lm[13] > 0 ? (int)lm[13] : 1;      // scan requests only (an apply post keeps its line count there)
        const int st_arc0 = (int)lm[6], st_arc1 = (int)lm[8];
        const uint32_t st_val0 = lm[7], st_val1 = lm[9];
        const uint32_t p0_node = lm[10], p0_lo = lm[11], p0_hi = lm[12];
        if (lines > 1) __syncthreads();                    // everybody has read lines 0 and 1 before the patch lines land in lm (uniform condition)
        if (timed_out) {
            if (tid == 0 && blockIdx.x == 0) resident_exit(p.exit_word, 2u, served, scan_ticks);
            return;
        }
        const uint64_t t_seen = blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0;     // only workgroup 0's clock is reported
        if (cmd == 1u) {                                   // quit
            if (tid == 0 && blockIdx.x == 0) resident_exit(p.exit_word, 1u, served, scan_ticks);
            return;
        }
        // ---- patches: final values, applied by EVERY workgroup before it reads (same argument as scan_kernel).
        // The entries beyond the header come in chunks of 255 lines (1275 entries), each line verified by its tag before use.
        // PIREG: 0 = nothing beyond the header, 1 = compare directly, 2 = bitmap + gather the hits, 3 = gather everything again
        // a streamed list (apply posts seen, or this is one) only stores; the scan request then gathers everything again (mode 3)
        const int pr_mode = !PIREG ? 0 : ((apply_only || applied > 0) ? 3 : (entries == 0 ? 0 : (entries <= kPiRegCompare ? 1 : (entries <= kPiRegBitmapMax ? 2 : 3))));
        const int entries_here = apply_only ? upto * kMailboxPatchesPerLine : entries;      // entries that lines 1 .. lines-1 hold
        const int pi_here = apply_only ? entries_here : extra_pi;                          // ... of which potential patches come first
        bool torn = false;
        for (int first = applied + 1; first < lines; first += kChunk) {
            const int chunk = lines - first < kChunk ? lines - first : kChunk;
            const bool staged = first == 1 && chunk == 1 && line1_staged;
            for (int base = 0; base < chunk * 4 && !staged; base += nt * 4) {      // up to four 16-byte reads per thread in flight, one wait
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                v4u x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[q] = v4u{0u, 0u, 0u, 0u};
                    const int c = base + q * nt + tid;
                    if (c < chunk * 4) {
                        const int line = first + (c >> 2);      // line 1 sits in the poll unit, lines 2.. in the tail
                        const uint32_t *src = line == 1 ? my_unit + 16 + (c & 3) * 4 : p.mailbox + (kMailboxTail + (size_t)(line - 2) * 16 + (c & 3) * 4);
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x[q]) : "v"(src) : "memory");
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])::"memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = base + q * nt + tid;
                    if (c < chunk * 4) *reinterpret_cast<v4u *>(lm + 16 + c * 4) = x[q];
                }
            }
            __syncthreads();
            int bad = 0;
            for (int l = tid; l < chunk; l += nt) bad |= (lm[(1 + l) * 16 + 15] != seq);
            if (__syncthreads_or(bad)) { torn = true; break; }
            const int i_lo = (first - 1) * kMailboxPatchesPerLine;                    // entry index of the chunk's first entry
            const int i_hi = entries_here < i_lo + chunk * kMailboxPatchesPerLine ? entries_here : i_lo + chunk * kMailboxPatchesPerLine;
            for (int i = i_lo + tid; i < i_hi; i += nt) {
                const int rel = i - i_lo;
                const uint32_t *q = lm + (1 + rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                if (i < pi_here) {
                    const int64_t v = (int64_t)(((uint64_t)q[2] << 32) | q[1]);
                    p.pi[q[0]] = (T)v;
                    if (LPI) lpi[q[0]] = (T)v;
                    if (pr_mode == 2) atomicOr(&bitmap[(q[0] >> 5) & (kPiRegBitmapWords - 1)], 1u << (q[0] & 31));
                } else {
                    const int a = (int)q[0] - p.base;
                    if ((unsigned)a < (unsigned)p.count_padded) p.state[a] = (int8_t)q[1];
                }
            }
            if (pr_mode == 1) {                             // every thread matches the chunk's potential patches against its eight end points
                const int hi_pi = i_hi < pi_here ? i_hi : pi_here;
                for (int i = i_lo; i < hi_pi; ++i) {
                    const int rel = i - i_lo;
                    const uint32_t *q = lm + (1 + rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                    const int node = (int)q[0];
                    const T v = (T)(int64_t)(((uint64_t)q[2] << 32) | q[1]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ps[j] = mine.s.v[j] == node ? v : ps[j]; pt[j] = mine.t.v[j] == node ? v : pt[j]; }
                }
            }
            if (REG && extra_st > 0 && !apply_only) {       // every thread checks the chunk's state patches against its four arcs
                const int s_lo = i_lo > extra_pi ? i_lo : extra_pi;
                for (int i = s_lo; i < i_hi; ++i) {
                    const int rel = i - i_lo;
                    const uint32_t *q = lm + (1 + rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                    const int a = (int)q[0] - p.base - my_i0;
                    if ((unsigned)a < 4u) mine.st4 = (mine.st4 & ~(0xFFu << (8 * a))) | ((q[1] & 0xFFu) << (8 * a));
                }
            }
            __syncthreads();                               // the chunk has been consumed before the next one lands in lm
        }
        if (torn) continue;                                // a line was still in flight: poll again (re-applying final values is harmless)
        if (apply_only) {                                  // part of a list: remember how far it got, no answer
            applied = upto > applied ? upto : applied;
            last_sub = sub;
            idle_since = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            continue;
        }
        // header patches.  With nothing but them (n_pi <= 1, n_st <= 2) and register-resident arcs nobody has to wait for the stores:
        // the potential is substituted from the request while its store retires behind the scan.
        const bool fast = REG && (entries == 0 || pr_mode == 1) && applied == 0;
        const T v0 = (T)(int64_t)(((uint64_t)p0_hi << 32) | p0_lo);
        if (n_pi | n_st) {
            if (tid == 0 && n_pi > 0) { p.pi[p0_node] = v0; if (LPI) lpi[p0_node] = v0; }
            if (tid == 1 && n_st > 0) { const int a = st_arc0 - p.base; if ((unsigned)a < (unsigned)p.count_padded) p.state[a] = (int8_t)st_val0; }
            if (tid == 2 && n_st > 1) { const int a = st_arc1 - p.base; if ((unsigned)a < (unsigned)p.count_padded) p.state[a] = (int8_t)st_val1; }
            if (REG) {
                if (n_st > 0) { const int a = st_arc0 - p.base - my_i0; if ((unsigned)a < 4u) mine.st4 = (mine.st4 & ~(0xFFu << (8 * a))) | ((st_val0 & 0xFFu) << (8 * a)); }
                if (n_st > 1) { const int a = st_arc1 - p.base - my_i0; if ((unsigned)a < 4u) mine.st4 = (mine.st4 & ~(0xFFu << (8 * a))) | ((st_val1 & 0xFFu) << (8 * a)); }
            }
            if (!fast) {
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
            }
        }
        const int sub_node = (fast && n_pi > 0) ? (int)p0_node : -1;
        if (PIREG) {
            if (pr_mode == 2) {                            // the stores have retired (barrier above): lanes whose end points are on the list gather again
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = (uint32_t)mine.s.v[j], b = (uint32_t)mine.t.v[j];
                    if ((bitmap[(a >> 5) & (kPiRegBitmapWords - 1)] >> (a & 31)) & 1u) ps[j] = p.pi[a];
                    if ((bitmap[(b >> 5) & (kPiRegBitmapWords - 1)] >> (b & 31)) & 1u) pt[j] = p.pi[b];
                }
                __syncthreads();                           // everybody has tested before the bitmap is wiped
                for (int i = tid; i < kPiRegBitmapWords; i += nt) bitmap[i] = 0u;
            } else if (pr_mode == 3) {
                gather_tile<T>(mine, p.pi, ps, pt);
            }
            if (n_pi > 0) {                                // the header patch is on no list
#pragma unroll
                for (int j = 0; j < 4; ++j) { ps[j] = mine.s.v[j] == (int)p0_node ? v0 : ps[j]; pt[j] = mine.t.v[j] == (int)p0_node ? v0 : pt[j]; }
            }
        }
        // ---- scan
        if (CAND) {
            int64_t c1, c2;
            uint32_t p1, p2;
            if (PIREG) fold_tile_best2<T>(mine, ps, pt, p.base + my_i0, c1, p1, c2, p2);
            else eval_tile_best2<T>(mine, pi_view, p.base + my_i0, c1, p1, c2, p2, sub_node, v0);
            publish_candidates(c1, p1, c2, p2, p.slots, seq);
        } else {
            Key best{0, kNone, kNone}, range{0, kNone, kNone};
            if (PIREG) {
                fold_tile<T, RULE, OPT>(mine, ps, pt, p.base + my_i0, p.m_s, next_arc, block_size, rstar, best, range);
            } else if (REG) {
                eval_tile<T, RULE, OPT>(mine, pi_view, p.base + my_i0, p.m_s, next_arc, block_size, rstar, best, range, sub_node, v0);
            } else {
                for (int i0 = my_i0; i0 < p.count_padded; i0 += gridDim.x * nt * kArcsPerThread) {
                    TileData<T> d;
                    load_tile<T>(p.src, p.tgt, p.cost, p.state, i0, d);
                    eval_tile<T, RULE, OPT, PERM>(d, pi_view, p.base + i0, p.m_s, next_arc, block_size, rstar, best, range, -1, (T)0, p.orig, p.base);
                }
                if (PERM && best.p != kNone) best.p = (uint32_t)p.orig[(int)best.p - p.base];
            }
            publish_best<RULE, true, kResidentThreads, kDual<RULE, OPT>>(best, p.slots + (size_t)blockIdx.x * kSlotStride, seq, true, range);
        }
        last = seq;
        applied = 0;
        served += 1;
        idle_since = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) scan_ticks += idle_since - t_seen;
        if (fast) __builtin_amdgcn_s_waitcnt(0);           // the patches' stores have retired before anybody gathers again
        __syncthreads();                                   // lm and the reduction scratch are reused by the next request
    }
}

// ------------------------------------------------------------------------------------------------ the candidate cache's grid, register resident
// resident_cand_kernel<T>: Best Eligible with candidates for instances whose arcs AND end-point potentials live in registers (the headline
// workload).  What distinguishes it from resident_kernel<..., CAND, PIREG>: a request is applied to the registers straight from the request,
// and nothing else is touched -- no store into the potential / state arrays in memory, no wait for such stores, no gather after them.
//   * SHIFT LIST: the one big subtree a pivot moved, as bare node ids (15 per line) + the pivot's sigma (NS.cs:1187-1190: one sigma for
//     the whole subtree).  Every workgroup sets the nodes' bits in an exact bitmap in LDS (one bit per node: node_count <= kShiftBits) and
//     every thread adds sigma to those of its eight end points whose bit is set.  The lines travel while the host is still walking the
//     subtree (cmd 2 posts: set bits, no answer); 4 bytes per node instead of 12, and the host never forms the values.
//   * VALUE ENTRIES {node, value}: the nodes the host-side cache touched since the last request, with their final values (they override
//     the shift: a value is read by the host when the request is built).  Few: compared directly; many: rounds of up to kCandRoundLines
//     lines through a hash table in LDS (node -> value) that every thread probes for its eight end points.
//   * STATE WRITES {arc, state}: compared against the thread's four arcs.
//   * RELOAD (cmd 3): a walk so long that its node list would cost more than it is worth names no nodes: the host only moves its own
//     potentials, the workgroups copy the bound array (mapped host memory; each its slice, coalesced, over PCIe) into the device array, meet
//     at a grid-wide barrier and every thread gathers its end points' potentials again.  The one place where this grid touches memory.
// The arrays in memory are only read when the grid starts: the host, whose mirrors are authoritative in candidate mode, writes them again
// before every launch (resident_start) -- so a grid that left on its idle timeout comes back with current values and the request it finds
// waiting is re-posted without patches.
constexpr int kShiftBits = 131072;                // nodes the exact bitmap covers (16 KB of LDS)
constexpr int kShiftNodesPerLine = 15;
constexpr int kShiftPairsPerLine = 7;             // ... or seven {first id, length} pairs: runs of consecutive ids (the host relabels the nodes in thread order)
constexpr int kCandLines = 512;                   // staging: line 0 + 511 lines (32 KB)
constexpr int kCandRoundLines = 480;              // value-entry lines per hash round: 2400 entries in ...
constexpr int kCandHash = 4096;                   // ... this many slots
constexpr uint32_t kHashEmpty = 0xFFFFFFFFu;
constexpr int kCandCompare = 3;                   // value entries beyond the header's that are matched by direct comparison (~40 instructions each)

// Mailbox of this grid.  Poll unit (replicated): line 0 = header, line 1 = first entry line.
//   line 0   [0] seq [1] cmd (0 scan, 1 quit, 2 shift lines in place, 3 reload the potentials, then scan) [2] n_val [3] scan: n_shift nodes / cmd 2: shift lines in place so far
//            [4] cmd 2: post counter [5] n_st [6..9] state writes 0, 1 {arc, state} [10..12] value entry 0 {node, lo, hi} [13..14] sigma [15] seq
//   entry lines (tail, from kMailboxTail): five {a, b, c} each: value entries 1.., then state writes 2..; [15] = seq
//   shift lines (from shift_base): fifteen node ids each, or (scan: line 0 [4] == 1; in-place posts: cmd 4 instead of 2) seven {first id,
//            length} pairs; [15] = seq of the scan request they belong to
// TILES register tiles of four arcs per thread (tile t of a thread lies gridDim * blockDim * 4 arcs behind tile t - 1), at most kCandThreads
// threads: four waves, one per SIMD of the CU.  A wave instruction takes four cycles and two waves on one SIMD take turns, so the reductions of
// seven waves of 448 threads (one tile each) ran at half the rate of these four.
constexpr int kCandThreads = 256;
template <typename T, int TILES>
__global__ __launch_bounds__(kCandThreads) void resident_cand_kernel(const ResidentParams<T> p, const uint32_t shift_base, const int max_shift_lines)
{
    __shared__ __attribute__((aligned(16))) uint32_t lm[kCandLines * 16];
    __shared__ uint32_t bitmap[kShiftBits / 32];
    __shared__ uint32_t vmap[kShiftBits / 32];           // marks of the value entries of the round under way
    __shared__ uint32_t hkey[kCandHash];
    __shared__ __attribute__((aligned(8))) int64_t hval[kCandHash];
    __shared__ uint32_t s_timeout;
    // state writes are POSTED to the thread that holds the arc: whoever reads an entry marks byte `slot` of the owner's word pair, and every
    // thread looks at its own words once per request (a loop over the entries in every thread cost 60+ cycles per entry and thread)
    __shared__ uint32_t sp_mask[TILES * kCandThreads], sp_val[TILES * kCandThreads];
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    int my_i[TILES];
    TileData<T> mine[TILES];
    T ps[TILES][4], pt[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        my_i[t] = (t * (int)gridDim.x + (int)blockIdx.x) * nt * kArcsPerThread + tid * kArcsPerThread;
        const int at = my_i[t] < p.count_padded ? my_i[t] : 0;          // behind the arrays: a tile of nothing (state 0 = never eligible)
        load_tile<T>(p.src, p.tgt, p.cost, p.state, at, mine[t]);
        if (my_i[t] >= p.count_padded) mine[t].st4 = 0u;
        gather_tile<T>(mine[t], p.pi, ps[t], pt[t]);
    }
    for (int i = tid; i < kShiftBits / 32; i += nt) { bitmap[i] = 0u; vmap[i] = 0u; }
    for (int i = tid; i < kCandHash; i += nt) hkey[i] = kHashEmpty;
    for (int i = tid; i < TILES * kCandThreads; i += nt) { sp_mask[i] = 0u; sp_val[i] = 0u; }
    __syncthreads();
    auto post_state = [&](int arc, uint32_t val) {
        const int local = arc - p.base;
        if ((unsigned)local >= (unsigned)p.count_padded) return;
        const int tile = local / (nt * kArcsPerThread), t = tile / (int)gridDim.x;
        if (tile % (int)gridDim.x != (int)blockIdx.x || t >= TILES) return;
        const int r = local % (nt * kArcsPerThread), owner = r / kArcsPerThread, slot = r % kArcsPerThread;
        atomicOr(&sp_mask[t * kCandThreads + owner], 0xFFu << (8 * slot));
        atomicOr(&sp_val[t * kCandThreads + owner], (val & 0xFFu) << (8 * slot));
    };
    uint32_t last = p.start_seq, served = 0, last_sub = 0, shifted_for = p.start_seq;
    int shift_done = 0;                                   // shift lines of the coming scan request whose bits are set
    bool bits_set = false;
    uint32_t reloaded_for = p.start_seq, barriers = 0;    // a request's reload happens once (a retry after a torn line must not meet the others at the barrier again)
    // grid-wide barrier (all workgroups are resident: one per CU); false when it gave up -- every spin of this kernel is bounded
    auto grid_barrier = [&]() -> bool {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        barriers += 1;
        if (tid == 0) {
            __hip_atomic_fetch_add(p.barrier, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = barriers * gridDim.x;
            const uint64_t t_bar = __builtin_amdgcn_s_memrealtime();
            uint32_t gave_up = 0u;
            while (__hip_atomic_load(p.barrier, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t_bar > 8ull * p.idle_ticks) { gave_up = 1u; break; }
            }
            s_timeout = gave_up ? 3u : 0u;
        }
        __syncthreads();
        if (s_timeout == 3u) return false;
        // what the other XCDs wrote through to memory may still sit in this CU's L1 / this XCD's L2 in its old form: forget it
        if (tid < 64) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        return true;
    };
    uint64_t scan_ticks = 0;
    uint64_t ph_shift = 0, ph_values = 0, ph_scan = 0, n_shift_req = 0;       // workgroup 0's clock by phase (exit record words 4..11)
    const uint64_t born_rt = __builtin_amdgcn_s_memrealtime(), born_clk = __builtin_amdgcn_s_memtime();
    uint64_t idle_since = __builtin_amdgcn_s_memrealtime();
    const uint32_t *const my_unit = p.mailbox + (size_t)(blockIdx.x % p.poll_replicas) * kReplicaStride;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int kStage = 32;                            // dword offset of the staging area in lm: lines 0 and 1 keep the poll unit
    constexpr int kChunk = kCandLines - 2;
    // stages n_a shift lines (from shift line first_a) followed by n_v entry lines (from entry line first_v; entry line 0 sits in the poll
    // unit, the others in the tail) into lm[kStage..] with ONE round of loads; true when every line carries the tag `seq`
    auto fetch = [&](int n_a, int first_a, int n_v, int first_v, uint32_t seq) -> bool {
        const int count = n_a + n_v;
        for (int base = 0; base < count * 4; base += nt * 4) {
            v4u x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x[q] = v4u{0u, 0u, 0u, 0u};
                const int c = base + q * nt + tid;
                if (c < count * 4) {
                    const int l = c >> 2;
                    const uint32_t *src;
                    if (l < n_a) src = p.mailbox + (shift_base + (size_t)(first_a + l) * 16 + (c & 3) * 4);
                    else {
                        const int v = first_v + l - n_a;
                        src = v == 0 ? my_unit + 16 + (c & 3) * 4 : p.mailbox + (kMailboxTail + (size_t)(v - 1) * 16 + (c & 3) * 4);
                    }
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x[q]) : "v"(src) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])::"memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = base + q * nt + tid;
                if (c < count * 4) *reinterpret_cast<v4u *>(lm + kStage + c * 4) = x[q];
            }
        }
        __syncthreads();
        int bad = 0;
        for (int l = tid; l < count; l += nt) bad |= (lm[kStage + l * 16 + 15] != seq);
        return __syncthreads_or(bad) == 0;
    };
    for (;;) {
        if (tid < 64) {
            v4u x = v4u{0u, 0u, 0u, 0u};
            uint32_t flag;
            for (;;) {
                if (tid < 8) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(my_unit + tid * 4) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                const uint32_t seq0 = lane_u32(x[0], 0), tag0 = lane_u32(x[3], 3), cmd0 = lane_u32(x[1], 0), sub0 = lane_u32(x[0], 1);
                if (seq0 != last && tag0 == seq0 && ((cmd0 != 2u && cmd0 != 4u) || sub0 != last_sub)) { flag = 1u; break; }
                if (__builtin_amdgcn_s_memrealtime() - idle_since > p.idle_ticks) { flag = 2u; break; }
                for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
            }
            if (tid < 8) *reinterpret_cast<v4u *>(lm + tid * 4) = x;
            if (tid == 0) s_timeout = flag;
        }
        __syncthreads();
        const uint32_t seq = lm[0], cmd = lm[1], sub = lm[4];
        int n_val = (int)lm[2], n_st = (int)lm[5];
        n_val = n_val < 0 ? 0 : (n_val > p.max_pi ? p.max_pi : n_val);
        n_st = n_st < 0 ? 0 : (n_st > p.max_st ? p.max_st : n_st);
        const bool apply_only = cmd == 2u || cmd == 4u;
        // the shift list comes as node ids (15 per line) or as {first, length} pairs (7 per line: runs of consecutive ids)
        const bool ranges = apply_only ? cmd == 4u : lm[4] == 1u;
        const int per_line = ranges ? kShiftPairsPerLine : kShiftNodesPerLine;
        int n_shift = apply_only ? 0 : (int)lm[3];                      // entries: nodes or pairs
        n_shift = n_shift < 0 ? 0 : (n_shift > max_shift_lines * per_line ? max_shift_lines * per_line : n_shift);
        int shift_lines = apply_only ? (int)lm[3] : (n_shift + per_line - 1) / per_line;
        shift_lines = shift_lines < 0 ? 0 : (shift_lines > max_shift_lines ? max_shift_lines : shift_lines);
        const int64_t sigma = (int64_t)(((uint64_t)lm[14] << 32) | lm[13]);
        const bool timed_out = s_timeout == 2u;
        const bool line1_staged = lm[31] == seq;
        const int st_arc0 = (int)lm[6], st_arc1 = (int)lm[8];
        const uint32_t st_val0 = lm[7], st_val1 = lm[9];
        const uint32_t v0_node = lm[10];
        const T v0 = (T)(int64_t)(((uint64_t)lm[12] << 32) | lm[11]);
        if (timed_out || cmd == 1u) {
            if (tid == 0 && blockIdx.x == 0) {
                // shader clock over this launch in MHz (s_memtime counts shader cycles, s_memrealtime 100 MHz): word 10
                const uint64_t d_rt = __builtin_amdgcn_s_memrealtime() - born_rt, d_clk = __builtin_amdgcn_s_memtime() - born_clk;
                n_shift_req = d_rt ? d_clk * 100 / d_rt : 0;
                const uint64_t ph[4] = {ph_shift, ph_values, ph_scan, n_shift_req};
                for (int q = 0; q < 4; ++q) {
                    __hip_atomic_store(p.exit_word + 4 + 2 * q, (uint32_t)ph[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(p.exit_word + 5 + 2 * q, (uint32_t)(ph[q] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                resident_exit(p.exit_word, timed_out ? 2u : 1u, served, scan_ticks);
            }
            return;
        }
        const uint64_t t_seen = blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0;
        // ---- reload: every potential anew from the caller's array, my end points' potentials anew from them (NS.cs:1196-1208 moved so many
        // nodes that the host sends no list: it only moved its own potentials)
        if (cmd == 3u && p.host_pi && reloaded_for != seq) {
            const int per = (p.n_nodes + (int)gridDim.x - 1) / (int)gridDim.x;
            const int lo = (int)blockIdx.x * per, hi = lo + per < p.n_nodes ? lo + per : p.n_nodes;
            // agent scope: written through to memory, where the workgroups of the other XCDs (each with an L2 of its own) will find it
            for (int i = lo + tid; i < hi; i += nt) __hip_atomic_store(p.pi + i, (T)p.host_pi[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!grid_barrier()) {
                // the grid leaves with code 3: the host writes the arrays from its mirrors, starts it again and posts the request as a plain scan
                if (tid == 0) resident_exit(p.exit_word, 3u, served, scan_ticks);
                return;
            }
#pragma unroll
            for (int t = 0; t < TILES; ++t) gather_tile<T>(mine[t], p.pi, ps[t], pt[t]);
            if (bits_set) {                                   // lines of a list that is part of the array now
                for (int i = tid; i < kShiftBits / 32; i += nt) bitmap[i] = 0u;
                bits_set = false;
                __syncthreads();
            }
            shift_done = 0;
            reloaded_for = seq;
        }
        // ---- what has to be fetched: shift lines not seen yet, entry lines (value entries 1.., then state writes 2..) unless the one line
        // there is came with the poll
        const int extra_val = n_val > 1 ? n_val - 1 : 0, extra_st = n_st > 2 ? n_st - 2 : 0, entries = apply_only ? 0 : extra_val + extra_st;
        const int lines = (entries + kMailboxPatchesPerLine - 1) / kMailboxPatchesPerLine;
        const bool inline_entries = lines == 1 && line1_staged;
        const int need_v = inline_entries ? 0 : lines;
        // the nodes of `chunk` staged shift lines (staged from lm line `off`; `first` = index of the first of them in the list): set their bits
        auto set_bits = [&](int off, int first, int chunk) {
            if (ranges) {
                // one thread per pair: a pair covers at most 256 ids = nine words of the bitmap
                const int left = n_shift - first * kShiftPairsPerLine;
                const int pairs_here = apply_only ? chunk * kShiftPairsPerLine : (left < chunk * kShiftPairsPerLine ? left : chunk * kShiftPairsPerLine);
                for (int i = tid; i < pairs_here; i += nt) {
                    const uint32_t *q = lm + kStage + (off + i / kShiftPairsPerLine) * 16 + 2 * (i % kShiftPairsPerLine);
                    const uint32_t a = q[0];
                    uint32_t len = q[1];
                    if (a >= (uint32_t)kShiftBits || len == 0u) continue;
                    if (len > (uint32_t)kShiftBits - a) len = (uint32_t)kShiftBits - a;
                    const uint32_t z = a + len - 1u;                  // last id
                    for (uint32_t w = a >> 5; w <= (z >> 5); ++w) {
                        uint32_t mask = 0xFFFFFFFFu;
                        if (w == (a >> 5)) mask &= 0xFFFFFFFFu << (a & 31u);
                        if (w == (z >> 5)) mask &= 0xFFFFFFFFu >> (31u - (z & 31u));
                        atomicOr(&bitmap[w], mask);
                    }
                }
                return;
            }
            const int left = n_shift - first * kShiftNodesPerLine;
            const int nodes_here = apply_only ? chunk * kShiftNodesPerLine : (left < chunk * kShiftNodesPerLine ? left : chunk * kShiftNodesPerLine);
            // A subtree's nodes come as runs of consecutive ids (the host relabels the nodes in thread order), and 32 consecutive ids share a
            // word of the bitmap: neighbouring lanes take entries 37 apart so that one atomic instruction does not hit one word 32 times
            int p2 = 64;
            while (p2 < nodes_here) p2 <<= 1;
            for (int x = tid; x < p2; x += nt) {
                const int i = (x * 37) & (p2 - 1);
                if (i < nodes_here) {
                    const uint32_t u = lm[kStage + (off + i / kShiftNodesPerLine) * 16 + i % kShiftNodesPerLine];
                    if (u < (uint32_t)kShiftBits) atomicOr(&bitmap[u >> 5], 1u << (u & 31));
                }
            }
        };
        // the shift: once per request (a retry after a torn line must not repeat it), before the value entries, which override it
        auto shift_once = [&]() {
            if (n_shift > 0 && shifted_for != seq) {
                // all the words first (independent LDS reads, one wait), then the adds as selects: with one wave per SIMD nothing hides a
                // dependent LDS round trip (64 cycles each)
                uint32_t wa[TILES][4], wb[TILES][4];
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wa[t][j] = bitmap[((uint32_t)mine[t].s.v[j] >> 5) & (kShiftBits / 32 - 1)];
                        wb[t][j] = bitmap[((uint32_t)mine[t].t.v[j] >> 5) & (kShiftBits / 32 - 1)];
                    }
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t da = ((wa[t][j] >> ((uint32_t)mine[t].s.v[j] & 31)) & 1u) ? sigma : 0;
                        const int64_t db = ((wb[t][j] >> ((uint32_t)mine[t].t.v[j] & 31)) & 1u) ? sigma : 0;
                        ps[t][j] = (T)((int64_t)ps[t][j] + da);
                        pt[t][j] = (T)((int64_t)pt[t][j] + db);
                    }
                shifted_for = seq;
            }
            if (n_val > 0) {
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ps[t][j] = mine[t].s.v[j] == (int)v0_node ? v0 : ps[t][j]; pt[t][j] = mine[t].t.v[j] == (int)v0_node ? v0 : pt[t][j]; }
            }
            if (tid == 0 && n_st > 0) post_state(st_arc0, st_val0);
            if (tid == 1 && n_st > 1) post_state(st_arc1, st_val1);
        };
        // `chunk` entry lines starting with entry line `first`, found at dword offset `at` of lm: value entries into the registers (few:
        // compared directly; many: through the hash table), state writes against my four arcs
        auto apply_entries = [&](const uint32_t *at, int first, int chunk) {
            const int i_lo = first * kMailboxPatchesPerLine;
            const int i_hi = entries < i_lo + chunk * kMailboxPatchesPerLine ? entries : i_lo + chunk * kMailboxPatchesPerLine;
            const int v_hi = i_hi < extra_val ? i_hi : extra_val;                       // value entries here: [i_lo, v_hi)
            const int n_here = v_hi > i_lo ? v_hi - i_lo : 0;
            if (n_here <= kCandCompare) {
                for (int i = i_lo; i < v_hi; ++i) {
                    const int rel = i - i_lo;
                    const uint32_t *q = at + (rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                    const int node = (int)q[0];
                    const T v = (T)(int64_t)(((uint64_t)q[2] << 32) | q[1]);
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { ps[t][j] = mine[t].s.v[j] == node ? v : ps[t][j]; pt[t][j] = mine[t].t.v[j] == node ? v : pt[t][j]; }
                }
            } else {
                // Many entries: every entry is inserted by one thread into a hash table (node -> value; a node may come twice, with the same
                // value: the first claim of the slot wins) and marked in a second exact bitmap; a thread then tests its eight end points
                // against the bitmap (one LDS read each) and probes the table only for the few that are marked.  A wave instruction costs four
                // cycles here, so what counts is instructions per thread: comparing 20 entries against 8 end points directly is ~800 of them.
                int p2 = 64;
                while (p2 < n_here) p2 <<= 1;
                for (int x = tid; x < p2; x += nt) {
                    const int rel = (x * 37) & (p2 - 1);                                // neighbouring lanes: entries 37 apart (see set_bits)
                    if (rel >= n_here) continue;
                    const uint32_t *q = at + (rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                    const uint32_t node = q[0];
                    atomicOr(&vmap[(node >> 5) & (kShiftBits / 32 - 1)], 1u << (node & 31));
                    uint32_t h = (node * 2654435761u) >> 20;                            // 12 bits
                    for (int step = 0; step < kCandHash; ++step) {
                        const uint32_t old = atomicCAS(&hkey[h], kHashEmpty, node);
                        if (old == kHashEmpty || old == node) { hval[h] = (int64_t)(((uint64_t)q[2] << 32) | q[1]); break; }
                        h = (h + 1) & (kCandHash - 1);
                    }
                }
                __syncthreads();
                uint32_t marks = 0;                                                     // which of my end points are marked: all words first, one wait
                {
                    uint32_t w[TILES][8];
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[t][j] = vmap[((uint32_t)(j < 4 ? mine[t].s.v[j & 3] : mine[t].t.v[j & 3]) >> 5) & (kShiftBits / 32 - 1)];
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int j = 0; j < 8; ++j) marks |= ((w[t][j] >> ((uint32_t)(j < 4 ? mine[t].s.v[j & 3] : mine[t].t.v[j & 3]) & 31)) & 1u) << (8 * t + j);
                }
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t node = (uint32_t)(j < 4 ? mine[t].s.v[j & 3] : mine[t].t.v[j & 3]);
                        if ((marks >> (8 * t + j)) & 1u) {
                            uint32_t h = (node * 2654435761u) >> 20;
                            for (int step = 0; step < kCandHash; ++step) {
                                const uint32_t k = hkey[h];
                                if (k == kHashEmpty) break;
                                if (k == node) { const T v = (T)hval[h]; if (j < 4) ps[t][j & 3] = v; else pt[t][j & 3] = v; break; }
                                h = (h + 1) & (kCandHash - 1);
                            }
                        }
                    }
                __syncthreads();
                // wipe: the marks by walking the entries again (whole words: no atomics needed, everybody writes 0), the table wholesale
                for (int i = i_lo + tid; i < v_hi; i += nt) {
                    const int rel = i - i_lo;
                    const uint32_t node = at[(rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine)];
                    vmap[(node >> 5) & (kShiftBits / 32 - 1)] = 0u;
                }
                for (int i = tid; i < kCandHash; i += nt) hkey[i] = kHashEmpty;
            }
            if (extra_st > 0 && i_hi > extra_val) {
                const int s_lo = i_lo > extra_val ? i_lo : extra_val;
                for (int i = s_lo + tid; i < i_hi; i += nt) {
                    const int rel = i - i_lo;
                    const uint32_t *q = at + (rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                    post_state((int)q[0], q[1]);
                }
            }
        };
        bool torn = false;
        uint64_t t_shifted = 0;
        const int pend_a = shift_lines - shift_done;
        if (!apply_only && pend_a + need_v <= kChunk && need_v <= kCandRoundLines) {
            // the usual case: everything this request still needs arrives with ONE round of loads
            if (pend_a + need_v > 0) {
                if (!fetch(pend_a, shift_done, need_v, 0, seq)) continue;
                if (pend_a > 0) { set_bits(0, shift_done, pend_a); bits_set = true; shift_done += pend_a; __syncthreads(); }
            }
            t_shifted = blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0;
            shift_once();
            if (inline_entries) apply_entries(lm + 16, 0, 1);
            else if (need_v > 0) apply_entries(lm + kStage + pend_a * 16, 0, need_v);
        } else {
            for (int first = shift_done; first < shift_lines; first += kChunk) {
                const int chunk = shift_lines - first < kChunk ? shift_lines - first : kChunk;
                if (!fetch(chunk, first, 0, 0, seq)) { torn = true; break; }
                set_bits(0, first, chunk);
                bits_set = true;
                shift_done = first + chunk;
                __syncthreads();
            }
            if (torn) continue;
            if (apply_only) {
                last_sub = sub;
                idle_since = __builtin_amdgcn_s_memrealtime();
                continue;
            }
            t_shifted = blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0;
            shift_once();
            if (inline_entries) apply_entries(lm + 16, 0, 1);
            else {
                for (int first = 0; first < lines; first += kCandRoundLines) {
                    const int chunk = lines - first < kCandRoundLines ? lines - first : kCandRoundLines;
                    if (!fetch(0, 0, chunk, first, seq)) { torn = true; break; }
                    apply_entries(lm + kStage, first, chunk);
                    __syncthreads();                                                    // the round has been consumed before the next one lands in lm
                }
                if (torn) continue;                                                     // value entries are final values: repeating them is harmless; the shift is guarded
            }
        }
        // ---- the state writes that were posted to me
        if (n_st > 0) {
            __syncthreads();
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const uint32_t m = sp_mask[t * kCandThreads + tid];
                if (m) {
                    mine[t].st4 = (mine[t].st4 & ~m) | (sp_val[t * kCandThreads + tid] & m);
                    sp_mask[t * kCandThreads + tid] = 0u;
                    sp_val[t * kCandThreads + tid] = 0u;
                }
            }
        }
        // ---- scan
        const uint64_t t_patched = blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0;
        int64_t c1 = 0, c2 = 0;                         // best and second best of this thread's arcs (its tiles come in increasing arc id)
        uint32_t p1 = kNone, p2 = kNone;
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int st = (int)(int8_t)(mine[t].st4 >> (8 * j));
                const int64_t dd = (int64_t)mine[t].c.v[j] + (int64_t)ps[t][j] - (int64_t)pt[t][j];
                const int64_t rc = st > 0 ? dd : (st < 0 ? -dd : 0);
                const uint32_t e = (uint32_t)(p.base + my_i[t] + j);
                const bool b1 = rc < c1;
                const bool b2 = !b1 && rc < c2;
                c2 = b1 ? c1 : (b2 ? rc : c2);
                p2 = b1 ? p1 : (b2 ? e : p2);
                c1 = b1 ? rc : c1;
                p1 = b1 ? e : p1;
            }
        publish_candidates(c1, p1, c2, p2, p.slots, seq);
        if (bits_set) {
            for (int i = tid; i < kShiftBits / 32; i += nt) bitmap[i] = 0u;
            bits_set = false;
        }
        shift_done = 0;
        last = seq;
        served += 1;
        idle_since = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) {
            scan_ticks += idle_since - t_seen;
            ph_shift += t_shifted - t_seen; ph_values += t_patched - t_shifted; ph_scan += idle_since - t_patched;
            n_shift_req += n_shift > 0 ? 1 : 0;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ resident grid over the RC layout
// The RC layout's scan (scan_rc_kernel) still pays one dispatch per search (about 6.5 us + 3 us of host launch on a 3-15 us scan).  This grid
// serves the searches through the same mailbox as resident_kernel:
//   LD = true   the workgroup's window of arcs (up to kRcWindow: state + reduced cost, 9 B per arc) lives in LDS -- an arc shard of a large
//               instance (config 5 over 8 GPUs: 1.125 M arcs) is scanned without touching memory;
//   LD = false  the windows are streamed from memory for every request (a whole large instance on one GPU).
// A request carries the State[] writes and the MOVED NODES WITH THEIR SHIFT ({node, delta} where resident_kernel gets {node, value}); every
// workgroup walks the moved nodes' arc lists and shifts the arcs of its own window (LDS copy and memory), so every arc is shifted exactly once,
// by its only reader.  A request fits one staging chunk (kRcResidentNodes nodes); the host sends longer lists through update_rc_kernel with the
// grid stopped.
constexpr int kRcWindow = 8192;                    // arcs per workgroup in LDS: 64 KB of reduced costs + 8 KB of states
constexpr int kRcResidentNodes = 1 + 255 * kMailboxPatchesPerLine - 8;   // moved nodes that fit ONE staging chunk of entry lines (longer lists take several)

// running best and second best (c, arc) of a thread's arcs in the RC layout (candidate variant of resident_rc_kernel, Best Eligible only)
__device__ __forceinline__ void fold_rc_best2(uint32_t st4, const int64_t d[4], int e0, int64_t &c1, uint32_t &p1, int64_t &c2, uint32_t &p2)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int st = (int)(int8_t)(st4 >> (8 * j));
        const int64_t rc = st > 0 ? d[j] : (st < 0 ? -d[j] : 0);
        const uint32_t e = (uint32_t)(e0 + j);
        // a thread's arcs come in increasing id (within a tile and from trip to trip) and c1, c2 start at 0 ("none"): better <=> strictly smaller key
        const bool b1 = rc < c1;
        const bool b2 = !b1 && rc < c2;
        c2 = b1 ? c1 : (b2 ? rc : c2);
        p2 = b1 ? p1 : (b2 ? e : p2);
        c1 = b1 ? rc : c1;
        p1 = b1 ? e : p1;
    }
}

struct ResidentRcParams {
    int8_t *state;
    int64_t *rc;
    void *pi;
    const int32_t *adj_start;
    const uint32_t *adj;
    Slot *slots;
    const uint32_t *mailbox;
    uint32_t *exit_word;
    // in-grid reload (cmd 3): the arcs' end points and costs, the caller's bound potentials as the device sees them (mapped host memory; nullptr:
    // the host never posts a reload), and a counter in device memory for the grid-wide barrier (zeroed by the host before every launch)
    const int32_t *src, *tgt;
    const void *cost;
    const int64_t *host_pi;
    uint32_t *barrier;
    int32_t base, count_padded, m_s, window;       // window: arcs per workgroup (LD), multiple of 4 * blockDim
    uint32_t start_seq, idle_ticks;
    int32_t narrow, max_pi, max_st, poll_replicas, poll_sleep, n_nodes;
};

// A request of this grid: the State[] writes and the moved nodes WITH THEIR SHIFT ({node, delta}), any number of them: the entry lines are
// staged and worked off in chunks of one staging area (255 lines).  Shifts are atomics -- applying a chunk twice would be wrong -- so the
// grid remembers how far it got with the request it is working on (a torn line makes it poll again and resume there).
//   cmd 3 (reload): "the bound potentials in host memory are current, every potential may have changed": the workgroups copy the array
//   (each its share, over PCIe), meet at a grid-wide barrier, and every workgroup computes the reduced costs of ITS arcs again -- what
//   mcf_engine_reload_potentials used to stop the grid for (memcpy + rc_init_kernel + a new launch).
template <int RULE, bool OPT, bool LD, bool CAND = false>
__global__ __launch_bounds__(kResidentThreads) void resident_rc_kernel(const ResidentRcParams p)
{
    static_assert(!CAND || RULE == MCF_RULE_BEST_ELIGIBLE, "candidate lists serve Best Eligible");
    constexpr int kLines = kMailboxLines;                      // line 0 + one chunk of 255 entry lines
    constexpr int kChunk = kLines - 1;
    constexpr int kNodesMax = 1280;
    __shared__ __attribute__((aligned(16))) uint32_t lm[kLines * 16];
    __shared__ __attribute__((aligned(16))) int64_t ld[LD ? kRcWindow : 2];      // reduced costs of my window
    __shared__ __attribute__((aligned(16))) int8_t ls[LD ? kRcWindow : 4];       // their states
    __shared__ int32_t s_node[kNodesMax], s_lo[kNodesMax], s_pre[kNodesMax + 1];
    __shared__ int64_t s_delta[kNodesMax];
    __shared__ uint32_t s_timeout;
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    const int w_lo = LD ? (int)blockIdx.x * p.window : 0;                       // first position of my window
    typedef long v2l __attribute__((ext_vector_type(2)));
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    auto load_window = [&]() {
        for (int i = tid * 4; i < p.window; i += nt * 4) {
            const int g = w_lo + i;
            if (g < p.count_padded) {
                *reinterpret_cast<v2l *>(ld + i) = *reinterpret_cast<const v2l *>(p.rc + g);
                *reinterpret_cast<v2l *>(ld + i + 2) = *reinterpret_cast<const v2l *>(p.rc + g + 2);
                *reinterpret_cast<uint32_t *>(ls + i) = *reinterpret_cast<const uint32_t *>(p.state + g);
            } else {
                ld[i] = ld[i + 1] = ld[i + 2] = ld[i + 3] = 0;
                *reinterpret_cast<uint32_t *>(ls + i) = 0u;
            }
        }
        __syncthreads();
    };
    if (LD) load_window();
    uint32_t last = p.start_seq, served = 0, prog_seq = p.start_seq, barriers = 0;
    int prog_lines = 0;                                        // entry lines of request prog_seq already worked off
    bool prog_header = false;                                  // ... and its header entries
    uint64_t scan_ticks = 0;
    uint64_t idle_since = __builtin_amdgcn_s_memrealtime();
    const uint32_t *const my_unit = p.mailbox + (size_t)(blockIdx.x % p.poll_replicas) * kReplicaStride;
    // one moved node's arcs: shift those of MY window (LDS copy and memory), each arc exactly once, by its only reader
    auto shift_lists = [&](int n_here) {
        if (tid == 0) s_pre[0] = 0;
        __syncthreads();
        // inclusive scan of the list lengths (Hillis-Steele over at most kNodesMax entries; every thread owns up to two)
        for (int off = 1; off < n_here; off <<= 1) {
            int v0 = 0, v1 = 0;
            const int i0 = tid + 1, i1 = tid + 1 + nt;
            if (i0 <= n_here && i0 - off >= 1) v0 = s_pre[i0 - off];
            if (i1 <= n_here && i1 - off >= 1) v1 = s_pre[i1 - off];
            __syncthreads();
            if (i0 <= n_here) s_pre[i0] += v0;
            if (i1 <= n_here) s_pre[i1] += v1;
            __syncthreads();
        }
        const int total = s_pre[n_here];
        for (int f = tid; f < total; f += nt) {
            int k = 0;
            for (int step = 1024; step > 0; step >>= 1)           // largest k < n_here with s_pre[k] <= f
                if (k + step < n_here && s_pre[k + step] <= f) k += step;
            const uint32_t x = p.adj[s_lo[k] + (f - s_pre[k])];
            const int pos = (int)(x & 0x7FFFFFFFu);
            const bool mine = LD ? (pos >= w_lo && pos < w_lo + p.window) : ((int)(((uint32_t)pos / (uint32_t)(nt * kArcsPerThread)) % gridDim.x) == (int)blockIdx.x);
            if (mine) {
                const long long add = (x >> 31) ? -s_delta[k] : s_delta[k];
                atomicAdd(reinterpret_cast<unsigned long long *>(p.rc + pos), (unsigned long long)add);
                if (LD) atomicAdd(reinterpret_cast<unsigned long long *>(ld + (pos - w_lo)), (unsigned long long)add);
            }
        }
        if (blockIdx.x == 0) {
            for (int i = tid; i < n_here; i += nt) {
                // atomics: the candidate cache's requests carry several pivots' lists, a node may come more than once
                if (p.narrow) atomicAdd(reinterpret_cast<int32_t *>(p.pi) + s_node[i], (int32_t)s_delta[i]);
                else atomicAdd(reinterpret_cast<unsigned long long *>(p.pi) + s_node[i], (unsigned long long)s_delta[i]);
            }
        }
        __syncthreads();                                       // s_node .. s_pre are reused by the next chunk
    };
    auto state_write = [&](int arc, uint32_t val) {            // final values (memory: every workgroup, idempotent; LDS copy: the window's owner)
        const int a = arc - p.base;
        if ((unsigned)a < (unsigned)p.count_padded) {
            p.state[a] = (int8_t)val;
            if (LD && a >= w_lo && a < w_lo + p.window) ls[a - w_lo] = (int8_t)val;
        }
    };
    // grid-wide barrier (all workgroups are resident: one per CU).  Returns false when it gave up: a workgroup that never got a CU
    // (somebody else's grid holds them) must not hang the others -- every spin of this kernel is bounded.
    auto grid_barrier = [&]() -> bool {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        barriers += 1;
        if (tid == 0) {
            __hip_atomic_fetch_add(p.barrier, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = barriers * gridDim.x;
            const uint64_t t_bar = __builtin_amdgcn_s_memrealtime();
            uint32_t gave_up = 0u;
            while (__hip_atomic_load(p.barrier, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t_bar > 8ull * p.idle_ticks) { gave_up = 1u; break; }
            }
            s_timeout = gave_up ? 3u : 0u;
        }
        __syncthreads();
        if (s_timeout == 3u) return false;
        // what other XCDs wrote through to memory may still sit in this CU's L1 / this XCD's L2 in its old form: forget it (once per workgroup)
        if (tid < 64) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        return true;
    };
    for (;;) {
        if (tid < 64) {                                        // wave 0 polls lines 0 and 1 (see resident_kernel)
            v4u x = v4u{0u, 0u, 0u, 0u};
            uint32_t flag;
            for (;;) {
                if (tid < 8) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(my_unit + tid * 4) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                const uint32_t seq0 = lane_u32(x[0], 0), tag0 = lane_u32(x[3], 3);
                if (seq0 != last && tag0 == seq0) { flag = 1u; break; }
                if (__builtin_amdgcn_s_memrealtime() - idle_since > p.idle_ticks) { flag = 2u; break; }
                for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
            }
            if (tid < 8) *reinterpret_cast<v4u *>(lm + tid * 4) = x;
            if (tid == 0) s_timeout = flag;
        }
        __syncthreads();
        const uint32_t seq = lm[0], cmd = lm[1];
        int n_pi = (int)lm[4], n_st = (int)lm[5];
        n_pi = n_pi < 0 ? 0 : (n_pi > p.max_pi ? p.max_pi : n_pi);
        n_st = n_st < 0 ? 0 : (n_st > p.max_st ? p.max_st : n_st);
        const int extra_pi = n_pi > 1 ? n_pi - 1 : 0, extra_st = n_st > 2 ? n_st - 2 : 0, entries = extra_pi + extra_st;
        const int lines = (entries + kMailboxPatchesPerLine - 1) / kMailboxPatchesPerLine;
        const bool timed_out = s_timeout == 2u;
        const bool line1_staged = lm[31] == seq;
        const int next_arc = (int)lm[2], rstar = (int)lm[3];
        const int block_size = (int)lm[13] > 0 ? (int)lm[13] : 1;
        const int st_arc0 = (int)lm[6], st_arc1 = (int)lm[8];
        const uint32_t st_val0 = lm[7], st_val1 = lm[9];
        const uint32_t p0_node = lm[10], p0_lo = lm[11], p0_hi = lm[12];
        __syncthreads();                                       // everybody has read lines 0 and 1 before entry lines land in lm
        if (timed_out) {
            if (tid == 0 && blockIdx.x == 0) resident_exit(p.exit_word, 2u, served, scan_ticks);
            return;
        }
        const uint64_t t_seen = blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0;
        if (cmd == 1u) {
            if (tid == 0 && blockIdx.x == 0) resident_exit(p.exit_word, 1u, served, scan_ticks);
            return;
        }
        if (prog_seq != seq) { prog_seq = seq; prog_lines = 0; prog_header = false; }
        // ---- reload: every potential anew from the caller's array, every reduced cost of my arcs anew from them
        if (cmd == 3u && !prog_header && p.host_pi) {
            const int per = (p.n_nodes + (int)gridDim.x - 1) / (int)gridDim.x;
            const int lo = (int)blockIdx.x * per, hi = lo + per < p.n_nodes ? lo + per : p.n_nodes;
            for (int i = lo + tid; i < hi; i += nt) {
                const int64_t v = p.host_pi[i];
                // agent scope: written through to memory, where the workgroups of the other XCDs (each with an L2 of its own) will find it
                if (p.narrow) __hip_atomic_store(reinterpret_cast<int32_t *>(p.pi) + i, (int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(reinterpret_cast<int64_t *>(p.pi) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!grid_barrier()) {
                // the grid leaves with code 3; the host starts it again and the reload, which overwrites everything, runs from the start
                if (tid == 0) resident_exit(p.exit_word, 3u, served, scan_ticks);
                return;
            }
            const int step = LD ? nt * kArcsPerThread : (int)gridDim.x * nt * kArcsPerThread;
            const int begin = LD ? w_lo + tid * kArcsPerThread : (int)blockIdx.x * nt * kArcsPerThread + tid * kArcsPerThread;
            const int end = LD ? (w_lo + p.window < p.count_padded ? w_lo + p.window : p.count_padded) : p.count_padded;
            for (int i0 = begin; i0 < end; i0 += step) {
                const int4 s4 = *reinterpret_cast<const int4 *>(p.src + i0), t4 = *reinterpret_cast<const int4 *>(p.tgt + i0);
                const int sv[4] = {s4.x, s4.y, s4.z, s4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
                int64_t d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (p.narrow) d[j] = (int64_t)reinterpret_cast<const int32_t *>(p.cost)[i0 + j] + (int64_t)reinterpret_cast<const int32_t *>(p.pi)[sv[j]] - (int64_t)reinterpret_cast<const int32_t *>(p.pi)[tv[j]];
                    else d[j] = reinterpret_cast<const int64_t *>(p.cost)[i0 + j] + reinterpret_cast<const int64_t *>(p.pi)[sv[j]] - reinterpret_cast<const int64_t *>(p.pi)[tv[j]];
                }
                *reinterpret_cast<v2l *>(p.rc + i0) = v2l{d[0], d[1]};
                *reinterpret_cast<v2l *>(p.rc + i0 + 2) = v2l{d[2], d[3]};
                if (LD) { ld[i0 - w_lo] = d[0]; ld[i0 - w_lo + 1] = d[1]; ld[i0 - w_lo + 2] = d[2]; ld[i0 - w_lo + 3] = d[3]; }
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            // These were plain stores: the new reduced costs sit as DIRTY lines in this XCD's L2.  A later long list is dealt out -- workgroups
            // of other XCDs shift my arcs with atomics performed at memory -- and a dirty line written back afterwards would bury their shifts
            // (found by mcf_engine_check_reduced_costs on config 5: 206 arcs off after 1.5 M pivots).  Write the L2 back now.
            // ... and drop the copies: the lines stay in the L2 as clean copies otherwise, the shifts of the next requests are performed
            // at memory, and a scan that hits such a copy picks a wrong entering arc (seen as four pivots fewer over the whole of config 5).
            if (tid < 64) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- a LONG list (more moved nodes than one staging chunk holds): dealt out.  Every wave of the grid takes entry lines of its own,
        // walks the arc lists of the five nodes a line names (lanes over the arcs) and shifts the arcs wherever they are stored -- atomics at
        // agent scope, performed where every XCD sees them; then the workgroups meet at a grid-wide barrier, forget what their caches hold of
        // the reduced costs, and (LD) read their windows again.  The short lists of almost every request keep the barrier-free scheme below
        // (every workgroup walks every list and shifts only what it reads itself); that scheme read the arc lists 256 times over, which for a
        // list of 50 000 nodes was 0.9 ms per request.  Not repeatable (shifts are not idempotent): nothing here is retried, a line that has
        // not arrived yet is waited for.
        if (cmd != 3u && n_pi > kRcResidentNodes && p.barrier && !prog_header) {
            const int wave = tid >> 6, lane = tid & 63, waves = nt >> 6;
            uint32_t failed = 0u;
            if (blockIdx.x == 0 && wave == 0) {                // the header's moved node
                const int64_t delta = (int64_t)(((uint64_t)p0_hi << 32) | p0_lo);
                const int a0 = p.adj_start[p0_node], a1 = p.adj_start[p0_node + 1];
                for (int f = a0 + lane; f < a1; f += 64) {
                    const uint32_t x = p.adj[f];
                    atomicAdd(reinterpret_cast<unsigned long long *>(p.rc + (x & 0x7FFFFFFFu)), (unsigned long long)((x >> 31) ? -delta : delta));
                }
                if (lane == 0) {
                    if (p.narrow) atomicAdd(reinterpret_cast<int32_t *>(p.pi) + p0_node, (int32_t)delta);
                    else atomicAdd(reinterpret_cast<unsigned long long *>(p.pi) + p0_node, (unsigned long long)delta);
                }
            }
            for (int l = (int)blockIdx.x + wave * (int)gridDim.x; l < lines; l += (int)gridDim.x * waves) {
                const uint32_t *src = l == 0 ? my_unit + 16 : p.mailbox + (kMailboxTail + (size_t)(l - 1) * 16);
                v4u x = v4u{0u, 0u, 0u, 0u};
                const uint64_t t_line = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    if (lane < 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(src + lane * 4) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                    if (lane_u32(x[3], 3) == seq) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_line > p.idle_ticks) { failed = 1u; break; }
                }
                if (failed) break;
#pragma unroll
                for (int k = 0; k < kMailboxPatchesPerLine; ++k) {
                    const int e = l * kMailboxPatchesPerLine + k;
                    if (e >= entries) break;
                    const uint32_t q0 = lane_u32(x[(3 * k) & 3], (3 * k) >> 2), q1 = lane_u32(x[(3 * k + 1) & 3], (3 * k + 1) >> 2), q2 = lane_u32(x[(3 * k + 2) & 3], (3 * k + 2) >> 2);
                    if (e < extra_pi) {
                        const int64_t delta = (int64_t)(((uint64_t)q2 << 32) | q1);
                        const int a0 = p.adj_start[q0], a1 = p.adj_start[q0 + 1];
                        for (int f = a0 + lane; f < a1; f += 64) {
                            const uint32_t y = p.adj[f];
                            atomicAdd(reinterpret_cast<unsigned long long *>(p.rc + (y & 0x7FFFFFFFu)), (unsigned long long)((y >> 31) ? -delta : delta));
                        }
                        if (lane == 0) {
                            if (p.narrow) atomicAdd(reinterpret_cast<int32_t *>(p.pi) + q0, (int32_t)delta);
                            else atomicAdd(reinterpret_cast<unsigned long long *>(p.pi) + q0, (unsigned long long)delta);
                        }
                    }
                }
            }
            // The State[] writes are NOT dealt: like everywhere in this kernel every workgroup applies all of them with plain stores, so that
            // every XCD's L2 holds every write (one workgroup writing through to memory would leave the other XCDs' copies of the line stale).
            // They sit in the entry lines behind the moved nodes.
            for (int l = extra_pi / kMailboxPatchesPerLine + wave; l < lines && !failed; l += waves) {
                const uint32_t *src = l == 0 ? my_unit + 16 : p.mailbox + (kMailboxTail + (size_t)(l - 1) * 16);
                v4u x = v4u{0u, 0u, 0u, 0u};
                const uint64_t t_line = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    if (lane < 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(src + lane * 4) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                    if (lane_u32(x[3], 3) == seq) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_line > p.idle_ticks) { failed = 1u; break; }
                }
                if (failed) break;
#pragma unroll
                for (int k = 0; k < kMailboxPatchesPerLine; ++k) {
                    const int e = l * kMailboxPatchesPerLine + k;
                    if (e >= extra_pi && e < entries && lane == 0) state_write((int)lane_u32(x[(3 * k) & 3], (3 * k) >> 2), lane_u32(x[(3 * k + 1) & 3], (3 * k + 1) >> 2));
                }
            }
            if (tid == 0 && n_st > 0) state_write(st_arc0, st_val0);
            if (tid == 1 && n_st > 1) state_write(st_arc1, st_val1);
            const bool met = grid_barrier();
            if (__syncthreads_or((int)failed) || !met) {
                // part of the list is applied and part is not: there is no way back.  The host gets an error, not a wrong answer
                if (tid == 0 && blockIdx.x == 0) resident_exit(p.exit_word, 4u, served, scan_ticks);
                return;
            }
            if (LD) load_window();
            prog_header = true;
            prog_lines = lines;
        }
        // ---- header entries: moved node 0 {node, delta}, state writes 0 and 1
        if (!prog_header) {
            if (n_pi > 0 && cmd != 3u) {
                if (tid == 0) {
                    s_node[0] = (int)p0_node;
                    s_delta[0] = (int64_t)(((uint64_t)p0_hi << 32) | p0_lo);
                    const int a0 = p.adj_start[p0_node], a1 = p.adj_start[p0_node + 1];
                    s_lo[0] = a0;
                    s_pre[1] = a1 - a0;
                }
                shift_lists(1);
            }
            if (tid == 0 && n_st > 0) state_write(st_arc0, st_val0);
            if (tid == 1 && n_st > 1) state_write(st_arc1, st_val1);
            prog_header = true;
        }
        // ---- the entry lines in chunks, each line verified by its tag: moved nodes 1.., then state writes 2..
        bool torn = false;
        for (int first = prog_lines; first < lines; first += kChunk) {
            const int chunk = lines - first < kChunk ? lines - first : kChunk;
            const bool staged = first == 0 && chunk == 1 && line1_staged;
            if (!staged) {                                     // (a single line that came with the poll is in place already)
                for (int base = 0; base < chunk * 4; base += nt) {
                    const int c = base + tid;
                    if (c < chunk * 4) {
                        const int line = first + (c >> 2);          // entry line 0 sits in the poll unit, the others in the tail
                        const uint32_t *src = line == 0 ? my_unit + 16 + (c & 3) * 4 : p.mailbox + (kMailboxTail + (size_t)(line - 1) * 16 + (c & 3) * 4);
                        v4u x;
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(src) : "memory");
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                        *reinterpret_cast<v4u *>(lm + 16 + c * 4) = x;
                    }
                }
            }
            __syncthreads();
            int bad = 0;
            for (int l = tid; l < chunk; l += nt) bad |= (lm[(1 + l) * 16 + 15] != seq);
            if (__syncthreads_or(bad)) { torn = true; break; }
            const int i_lo = first * kMailboxPatchesPerLine;
            const int i_hi = entries < i_lo + chunk * kMailboxPatchesPerLine ? entries : i_lo + chunk * kMailboxPatchesPerLine;
            const int pi_hi = i_hi < extra_pi ? i_hi : extra_pi;                     // moved nodes of this chunk: [i_lo, pi_hi)
            const int n_here = pi_hi > i_lo ? pi_hi - i_lo : 0;
            for (int i = tid; i < n_here; i += nt) {
                const uint32_t *q = lm + (1 + i / kMailboxPatchesPerLine) * 16 + 3 * (i % kMailboxPatchesPerLine);
                const uint32_t node = q[0];
                const int a0 = p.adj_start[node], a1 = p.adj_start[node + 1];
                s_node[i] = (int)node;
                s_delta[i] = (int64_t)(((uint64_t)q[2] << 32) | q[1]);
                s_lo[i] = a0;
                s_pre[i + 1] = a1 - a0;
            }
            if (n_here > 0 && cmd != 3u) shift_lists(n_here);
            for (int i = (i_lo > extra_pi ? i_lo : extra_pi) + tid; i < i_hi; i += nt) {
                const int rel = i - i_lo;
                const uint32_t *q = lm + (1 + rel / kMailboxPatchesPerLine) * 16 + 3 * (rel % kMailboxPatchesPerLine);
                state_write((int)q[0], q[1]);
            }
            prog_lines = first + chunk;
            __syncthreads();                                   // the chunk has been consumed before the next one lands in lm
        }
        if (torn) continue;                                    // a line was still in flight: poll again and resume behind what has been applied
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        // The shifts above are atomics performed in this XCD's L2; a line of d that this CU's L1 still holds from an earlier request would be
        // stale.  Only the L1 has to forget it (buffer_inv sc0: the L1 alone -- an agent-scope invalidate also walks the L2 and cost 30 us
        // per request when every wave issued one).  The LDS variant never reads d from memory again.
        if (!LD && (n_pi || cmd == 3u)) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
        // ---- scan
        Key best{0, kNone, kNone}, range{0, kNone, kNone};
        int64_t c1 = 0, c2 = 0;              // CAND: best and second best of this thread's arcs
        uint32_t p1 = kNone, p2 = kNone;
        if (LD) {
            for (int i = tid * kArcsPerThread; i < p.window; i += nt * kArcsPerThread) {
                const uint32_t st4 = *reinterpret_cast<const uint32_t *>(ls + i);
                const int64_t d[4] = {ld[i], ld[i + 1], ld[i + 2], ld[i + 3]};
                if (CAND) fold_rc_best2(st4, d, p.base + w_lo + i, c1, p1, c2, p2);
                else fold_rc<RULE, OPT>(st4, d, p.base + w_lo + i, p.m_s, next_arc, block_size, rstar, best, range);
            }
        } else {
            const int step = gridDim.x * nt * kArcsPerThread;
            for (int i0 = blockIdx.x * nt * kArcsPerThread + tid * kArcsPerThread; i0 < p.count_padded; i0 += step) {
                const uint32_t st4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p.state + i0));
                const v2l a = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p.rc + i0));
                const v2l b = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p.rc + i0 + 2));
                const int64_t d[4] = {a.x, a.y, b.x, b.y};
                if (CAND) fold_rc_best2(st4, d, p.base + i0, c1, p1, c2, p2);
                else fold_rc<RULE, OPT>(st4, d, p.base + i0, p.m_s, next_arc, block_size, rstar, best, range);
            }
        }
        if (CAND) publish_candidates(c1, p1, c2, p2, p.slots, seq);
        else publish_best<RULE, true, kResidentThreads, kDual<RULE, OPT>>(best, p.slots + (size_t)blockIdx.x * kSlotStride, seq, true, range);
        last = seq;
        served += 1;
        idle_since = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) scan_ticks += idle_since - t_seen;
        __syncthreads();
    }
}

// The RCCL exchange of sharded engines wants this shard's candidate in DEVICE memory (the all-gather's send buffer): one workgroup folds the
// scan's per-workgroup records -- written to device memory in that mode -- with the rule's ordering, exactly as the host's collect() does.
// dual: the records' second half holds the range keys of OPTIMIZED Block Search; they are folded into the candidate's range_* fields.
template <int RULE>
__device__ __forceinline__ Key fold_records(const Slot *slots, int grid, int m_s, int na, int next_arc, int block_size, int rstar, bool range_keys, Key *wave_best)
{
    const int tid = threadIdx.x;
    Key best{0, kNone, kNone};
    for (int g = tid; g < grid; g += kThreads) {
        const Slot s = slots[(size_t)g * kSlotStride + (range_keys ? 2 : 0)];
        if (s.p == kNone) continue;
        uint32_t r = 0;
        if (RULE == MCF_RULE_BLOCK_SEARCH) {
            const int arc = (int)(((uint64_t)s.p + (uint32_t)na) % (uint32_t)m_s);
            if (range_keys) r = arc < na ? 1u : 0u;
            else {
                r = s.p / (uint32_t)block_size;
                r = 2 * r + ((rstar >= 0 && (int)r == rstar && arc < next_arc) ? 1u : 0u);
            }
        }
        if (best.p == kNone) { best.c = s.c; best.r = r; best.p = s.p; }
        else take_if_better<RULE>(best, s.c, r, s.p);
    }
    // "none" must lose against every record: give it the largest key of the rule's ordering before the butterflies
    if (best.p == kNone) { best.c = INT64_MAX; best.r = kNone; }
    best = wave_min<RULE>(best);
    __syncthreads();                                  // wave_best may still be read from the previous fold
    if ((tid & 63) == 0) wave_best[tid >> 6] = best;
    __syncthreads();
    Key k = wave_best[0];
    for (int w = 1; w < kThreads / 64; ++w) {
        const Key o = wave_best[w];
        if (o.p == kNone) continue;
        if (k.p == kNone) k = o;
        else take_if_better<RULE>(k, o.c, o.r, o.p);
    }
    return k;
}

template <int RULE>
__global__ __launch_bounds__(kThreads) void reduce_records_kernel(const Slot *slots, int grid, int m_s, int na, int next_arc, int block_size, int rstar, int dual, mcf_candidate *out)
{
    __shared__ Key wave_best[kThreads / 64];
    const Key k = fold_records<RULE>(slots, grid, m_s, na, next_arc, block_size, rstar, false, wave_best);
    Key q{0, kNone, kNone};
    if (dual) q = fold_records<RULE>(slots, grid, m_s, na, next_arc, block_size, rstar, true, wave_best);
    if (threadIdx.x == 0) {
        mcf_candidate c;
        c.reduced_cost = k.p == kNone ? 0 : k.c;
        c.pos = k.p;
        c.arc = k.p == kNone ? -1 : (RULE == MCF_RULE_BEST_ELIGIBLE ? (int32_t)k.p : (int32_t)(((uint64_t)k.p + (uint32_t)na) % (uint32_t)m_s));
        c.range_cost = q.p == kNone ? 0 : q.c;
        c.range_pos = q.p;
        c.range_arc = q.p == kNone ? -1 : (int32_t)(((uint64_t)q.p + (uint32_t)na) % (uint32_t)m_s);
        *out = c;
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

// RC layout, test aid: how many stored arcs' reduced cost differs from cost + pi[source] - pi[target]?
template <typename T>
__global__ __launch_bounds__(kThreads) void rc_check_kernel(const int32_t *src, const int32_t *tgt, const T *cost, const T *pi, const int64_t *rc, int count, unsigned long long *bad, int *first)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < count && rc[i] != (int64_t)cost[i] + (int64_t)pi[src[i]] - (int64_t)pi[tgt[i]]) { atomicAdd(bad, 1ull); atomicMin(first, i); }
}

// node ids relabelled by the host (mcf_engine_renumber_nodes): the arcs' end points follow; padding arcs (state 0) just get some valid id
__global__ __launch_bounds__(kThreads) void renumber_kernel(int32_t *src, int32_t *tgt, const int32_t *new_of, int count_padded)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < count_padded) { src[i] = new_of[src[i]]; tgt[i] = new_of[tgt[i]]; }
}

// evicts the caches before a "cold" measurement: reads a large buffer once (no stores, so no dirty lines are left to write back
// during the measured scan); the sum is kept alive through a store that practically never happens
__global__ void flush_kernel(uint4 *buf, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n16; i += stride) { const uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9E3779B9u) buf[0].x = acc;
}


}  // namespace
