// Solution validator on the device: the checks of the reference's SolutionValidator
// (src/MinCostFlow.Core/Lemon/Validation/SolutionValidator.cs) as reductions over the arcs and the nodes.
//
//   validate_arcs   one pass over the arcs (40 B per arc streamed: source, target, lower, upper, cost, flow; two potential
//                   gathers): bound checks (:104-124), arc complementary slackness (:146-177), sum flow*cost (:232-255), the arc
//                   terms of the dual cost (:287-302, :311-326) and the scatter of the net flow / the supply adjustment into
//                   per-node accumulators (:62-72, :296-301).  A thread owns four CONSECUTIVE arcs (16-byte loads, the layout of
//                   the scan kernels); arcs with flow 0 (all but ~n of them in a basic solution) and lower bound 0 do not
//                   scatter, and consecutive arcs with the same source (NETGEN order) scatter once.
//   validate_nodes  one pass over the nodes (40 B per node): conservation under the supply type (:75-99), node dual
//                   feasibility / slackness (:193-227), the node term of the dual cost (:305-308).
//   validate_fold   one workgroup adds up the per-workgroup partial results.
//
// All of it is HBM-streaming integer work; sums wrap like C# `long` in an unchecked context (unsigned 64-bit adds).
// A workgroup folds its findings with wave shuffles + LDS and writes ONE 64-byte partial record: atomics on a shared result
// block serialise (4096 workgroups x 6 atomics on one line cost 250 us at 8 M arcs, which is how this layout came about).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "common.h"

namespace {

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t err__ = (expr);                                                                         \
        if (err__ != hipSuccess) return mcf::fail(MCF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(err__)); \
    } while (0)

constexpr int kValThreads = 256;
constexpr int kValArcsPerThread = 4;
constexpr int kValTiles = 1;                // tiles of 1024 arcs in flight per trip (2 measured no faster: the gathers and the scatter bound it)
constexpr int kValMaxGroups = 1024;         // grid-stride beyond that: 4 workgroups per CU
constexpr uint32_t kNoId = 0xFFFFFFFFu;

// one per workgroup; cnt / first are per check in the order the kernel uses
struct alignas(64) ValPartial {
    unsigned long long sum0, sum1;
    uint32_t cnt[4];
    uint32_t first[4];
    uint32_t pad[4];
};

// the folded result
struct ValResult {
    unsigned long long count[MCF_VAL_KINDS];
    uint32_t first[MCF_VAL_KINDS];
    unsigned long long objective, dual_arcs, dual_nodes;
};

struct ValAcc {
    uint64_t sum0 = 0, sum1 = 0;
    uint32_t cnt[4] = {0, 0, 0, 0};
    uint32_t first[4] = {kNoId, kNoId, kNoId, kNoId};
    __device__ __forceinline__ void hit(int k, bool bad, uint32_t id)
    {
        cnt[k] += bad ? 1u : 0u;
        first[k] = (bad && id < first[k]) ? id : first[k];
    }
};

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}

// workgroup-wide fold of an accumulator into this workgroup's partial record
__device__ __forceinline__ void fold_to_partial(ValAcc a, ValPartial *out)
{
    __shared__ uint64_t s_sum[2][kValThreads / 64];
    __shared__ uint32_t s_cnt[4][kValThreads / 64], s_first[4][kValThreads / 64];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        a.sum0 += shfl_xor_u64(a.sum0, m);
        a.sum1 += shfl_xor_u64(a.sum1, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a.cnt[k] += (uint32_t)__shfl_xor((int)a.cnt[k], m, 64);
            const uint32_t o = (uint32_t)__shfl_xor((int)a.first[k], m, 64);
            a.first[k] = o < a.first[k] ? o : a.first[k];
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_sum[0][wave] = a.sum0;
        s_sum[1][wave] = a.sum1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_cnt[k][wave] = a.cnt[k]; s_first[k][wave] = a.first[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ValPartial p;
        p.sum0 = 0;
        p.sum1 = 0;
        for (int w = 0; w < kValThreads / 64; ++w) { p.sum0 += s_sum[0][w]; p.sum1 += s_sum[1][w]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t c = 0, f = kNoId;
            for (int w = 0; w < kValThreads / 64; ++w) { c += s_cnt[k][w]; f = s_first[k][w] < f ? s_first[k][w] : f; }
            p.cnt[k] = c;
            p.first[k] = f;
            p.pad[k] = 0;
        }
        *out = p;
    }
}

struct I32x4 { int32_t v[4]; };
struct I64x4 { int64_t v[4]; };
__device__ __forceinline__ I32x4 load4(const int32_t *p, int i0, int m)
{
    I32x4 r;
    if (i0 + 3 < m) {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i x = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + i0));
        r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[j] = i0 + j < m ? p[i0 + j] : 0;
    }
    return r;
}
__device__ __forceinline__ I64x4 load4(const int64_t *p, int i0, int m)
{
    I64x4 r;
    if (i0 + 3 < m) {
        typedef long v2l __attribute__((ext_vector_type(2)));
        const v2l x = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p + i0));
        const v2l y = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(p + i0 + 2));
        r.v[0] = x.x; r.v[1] = x.y; r.v[2] = y.x; r.v[3] = y.y;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[j] = i0 + j < m ? p[i0 + j] : 0;
    }
    return r;
}

// SKIP (measurement only, MCF_VAL_SKIP; results are wrong with it): 1 = no scatter, 2 = no potential gathers
template <int SKIP>
__global__ __launch_bounds__(kValThreads) void validate_arcs(const int32_t *__restrict__ src, const int32_t *__restrict__ tgt,
                                                             const int64_t *__restrict__ lower, const int64_t *__restrict__ upper,
                                                             const int64_t *__restrict__ cost, const int64_t *__restrict__ flow,
                                                             const int64_t *__restrict__ pi, unsigned long long *net,
                                                             unsigned long long *adj, int m, ValPartial *partials)
{
    ValAcc a;
    const int step = (int)(gridDim.x * kValThreads * kValArcsPerThread);
    // per trip: all streamed loads of a thread, then all of its gathers, are in flight together
    for (int base = (int)(blockIdx.x * kValThreads + threadIdx.x) * kValArcsPerThread; base < m; base += kValTiles * step) {
        I32x4 s[kValTiles], t[kValTiles];
        I64x4 lo[kValTiles], up[kValTiles], c[kValTiles], f[kValTiles];
        int64_t ps[kValTiles][4], pt[kValTiles][4];
#pragma unroll
        for (int u = 0; u < kValTiles; ++u) {
            const int i0 = base + u * step;
            s[u] = load4(src, i0, m); t[u] = load4(tgt, i0, m);
            lo[u] = load4(lower, i0, m); up[u] = load4(upper, i0, m); c[u] = load4(cost, i0, m); f[u] = load4(flow, i0, m);
        }
#pragma unroll
        for (int u = 0; u < kValTiles; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { ps[u][j] = (SKIP & 2) ? (int64_t)s[u].v[j] : pi[s[u].v[j]]; pt[u][j] = (SKIP & 2) ? (int64_t)t[u].v[j] : pi[t[u].v[j]]; }
#pragma unroll
        for (int u = 0; u < kValTiles; ++u) {
            const int i0 = base + u * step;
            uint64_t run_flow = 0, run_low = 0;      // pending scatter for the current run of equal sources
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = i0 + j < m;          // lanes past the end carry zeros: they fail nothing and add nothing
                const uint32_t id = (uint32_t)(i0 + j);
                const int64_t fj = f[u].v[j], lj = lo[u].v[j], uj = up[u].v[j], cj = c[u].v[j];
                const int64_t rc = (int64_t)((uint64_t)cj + (uint64_t)ps[u][j] - (uint64_t)pt[u][j]);
                a.hit(0, in && fj < lj, id);
                a.hit(1, in && fj > uj, id);
                a.hit(2, in && rc > 0 && fj != lj, id);
                a.hit(3, in && rc < 0 && fj != uj, id);
                a.sum0 += (uint64_t)fj * (uint64_t)cj;
                a.sum1 += (uint64_t)lj * (uint64_t)cj;
                if (in && rc < 0) a.sum1 -= ((uint64_t)uj - (uint64_t)lj) * (0 - (uint64_t)rc);
                run_flow += (uint64_t)fj;
                run_low += (uint64_t)lj;
                if (SKIP & 1) { a.sum1 += run_flow ^ run_low; continue; }
                if (fj != 0) atomicAdd(net + t[u].v[j], (unsigned long long)(0 - (uint64_t)fj));
                if (lj != 0) atomicAdd(adj + t[u].v[j], (unsigned long long)lj);
                if (j == 3 || s[u].v[j + 1 < 4 ? j + 1 : 3] != s[u].v[j]) {
                    if (run_flow) atomicAdd(net + s[u].v[j], (unsigned long long)run_flow);
                    if (run_low) atomicAdd(adj + s[u].v[j], (unsigned long long)(0 - run_low));
                    run_flow = 0;
                    run_low = 0;
                }
            }
        }
    }
    fold_to_partial(a, partials + blockIdx.x);
}

__global__ __launch_bounds__(kValThreads) void validate_nodes(const int64_t *__restrict__ supply, const int64_t *__restrict__ pi,
                                                              const unsigned long long *__restrict__ net,
                                                              const unsigned long long *__restrict__ adj, int n, int supply_type,
                                                              ValPartial *partials)
{
    ValAcc a;
    const int stride = (int)(gridDim.x * kValThreads);
    for (int i = (int)(blockIdx.x * kValThreads + threadIdx.x); i < n; i += stride) {
        const int64_t sp = supply[i], p = pi[i], nf = (int64_t)net[i];
        const uint64_t adjusted = (uint64_t)sp + (uint64_t)adj[i];
        const uint32_t id = (uint32_t)i;
        const bool ok = supply_type == MCF_SUPPLY_GEQ ? nf >= sp : (supply_type == MCF_SUPPLY_LEQ ? nf <= sp : nf == sp);
        a.hit(0, !ok, id);
        if (supply_type == MCF_SUPPLY_GEQ) {
            a.hit(1, p > 0, id);
            a.hit(2, p < 0 && nf != sp, id);
        } else if (supply_type == MCF_SUPPLY_LEQ) {
            a.hit(1, p < 0, id);
            a.hit(2, p > 0 && nf != sp, id);
        }
        a.sum0 -= adjusted * (uint64_t)p;
    }
    fold_to_partial(a, partials + blockIdx.x);
}

// partials [0, arc_groups) come from validate_arcs, [arc_groups, arc_groups + node_groups) from validate_nodes
__global__ __launch_bounds__(kValThreads) void validate_fold(const ValPartial *__restrict__ partials, int arc_groups, int node_groups, ValResult *res)
{
    ValAcc arcs, nodes;
    for (int i = (int)threadIdx.x; i < arc_groups + node_groups; i += kValThreads) {
        const ValPartial p = partials[i];
        ValAcc &a = i < arc_groups ? arcs : nodes;
        a.sum0 += p.sum0;
        a.sum1 += p.sum1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a.cnt[k] += p.cnt[k]; a.first[k] = p.first[k] < a.first[k] ? p.first[k] : a.first[k]; }
    }
    __shared__ ValPartial folded[2];
    fold_to_partial(arcs, &folded[0]);
    __syncthreads();
    fold_to_partial(nodes, &folded[1]);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int arc_kinds[4] = {MCF_VAL_LOWER, MCF_VAL_UPPER, MCF_VAL_SLACK_POS, MCF_VAL_SLACK_NEG};
        const int node_kinds[3] = {MCF_VAL_CONSERVATION, MCF_VAL_NODE_DUAL, MCF_VAL_NODE_SLACK};
        ValResult r;
        for (int k = 0; k < MCF_VAL_KINDS; ++k) { r.count[k] = 0; r.first[k] = kNoId; }
        for (int k = 0; k < 4; ++k) { r.count[arc_kinds[k]] = folded[0].cnt[k]; r.first[arc_kinds[k]] = folded[0].first[k]; }
        for (int k = 0; k < 3; ++k) { r.count[node_kinds[k]] = folded[1].cnt[k]; r.first[node_kinds[k]] = folded[1].first[k]; }
        r.objective = folded[0].sum0;
        r.dual_arcs = folded[0].sum1;
        r.dual_nodes = folded[1].sum0;
        *res = r;
    }
}

}  // namespace

struct mcf_validator {
    int device = 0, n = 0, m = 0;
    bool have_network = false, have_solution = false;
    int32_t *src = nullptr, *tgt = nullptr;
    int64_t *lower = nullptr, *upper = nullptr, *cost = nullptr, *supply = nullptr, *flow = nullptr, *pi = nullptr;
    unsigned long long *net = nullptr, *adj = nullptr;     // net and adj are one allocation of 2n words
    ValResult *res = nullptr;
    ValPartial *partials = nullptr;                          // 2 * kValMaxGroups records
    ValResult *h_res = nullptr;                              // pinned
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" {

int mcf_validator_create(mcf_validator **out, int32_t device, int32_t node_count, int32_t arc_count)
{
    if (!out) return mcf::fail(MCF_ERR_INVALID, "mcf_validator_create: null argument");
    *out = nullptr;
    if (node_count < 0 || arc_count < 0) return mcf::fail(MCF_ERR_INVALID, "mcf_validator_create: negative size");
    const int devs = mcf_device_count();
    if (devs <= 0) return mcf::fail(MCF_ERR_NO_DEVICE, "no HIP device: the validator runs on the device only");
    if (device < 0 || device >= devs) return mcf::fail(MCF_ERR_INVALID, "device %d of %d", device, devs);
    HIP_TRY(hipSetDevice(device));
    mcf_validator *v = new mcf_validator();
    v->device = device;
    v->n = node_count;
    v->m = arc_count;
    const size_t m = (size_t)arc_count + 1, n = (size_t)node_count + 1;    // never a zero-byte allocation
    hipError_t err = hipSuccess;
    auto take = [&](void **p, size_t bytes) { if (err == hipSuccess) err = hipMalloc(p, bytes); };
    take((void **)&v->src, m * 4); take((void **)&v->tgt, m * 4);
    take((void **)&v->lower, m * 8); take((void **)&v->upper, m * 8); take((void **)&v->cost, m * 8); take((void **)&v->flow, m * 8);
    take((void **)&v->supply, n * 8); take((void **)&v->pi, n * 8); take((void **)&v->net, 2 * n * 8);
    take((void **)&v->res, sizeof(ValResult));
    take((void **)&v->partials, 2 * kValMaxGroups * sizeof(ValPartial));
    if (err == hipSuccess) err = hipHostMalloc((void **)&v->h_res, sizeof(ValResult), hipHostMallocDefault);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreate(&v->ev0);
    if (err == hipSuccess) err = hipEventCreate(&v->ev1);
    if (err != hipSuccess) {
        mcf_validator_destroy(v);
        return mcf::fail(MCF_ERR_HIP, "mcf_validator_create: %s", hipGetErrorString(err));
    }
    v->adj = v->net + n;
    *out = v;
    return MCF_OK;
}

void mcf_validator_destroy(mcf_validator *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    void *dev[] = {v->src, v->tgt, v->lower, v->upper, v->cost, v->flow, v->supply, v->pi, v->net, v->res, v->partials};
    for (void *p : dev) if (p) (void)hipFree(p);
    if (v->h_res) (void)hipHostFree(v->h_res);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

int mcf_validator_upload(mcf_validator *v, const int32_t *source, const int32_t *target, const int64_t *lower, const int64_t *upper,
                         const int64_t *cost, const int64_t *supply, const int64_t *flow, const int64_t *pi)
{
    if (!v) return mcf::fail(MCF_ERR_INVALID, "mcf_validator_upload: null validator");
    const int net_args = (source != nullptr) + (target != nullptr) + (lower != nullptr) + (upper != nullptr) + (cost != nullptr) + (supply != nullptr);
    const int sol_args = (flow != nullptr) + (pi != nullptr);
    if ((net_args != 0 && net_args != 6) || (sol_args != 0 && sol_args != 2))
        return mcf::fail(MCF_ERR_INVALID, "mcf_validator_upload: pass the whole network (six arrays) and / or the whole solution (flow, pi)");
    HIP_TRY(hipSetDevice(v->device));
    if (net_args) {
        for (int e = 0; e < v->m; ++e)       // the kernels index the node arrays with these
            if ((unsigned)source[e] >= (unsigned)v->n || (unsigned)target[e] >= (unsigned)v->n)
                return mcf::fail(MCF_ERR_INVALID, "arc %d: end point out of range", e);
        HIP_TRY(hipMemcpyAsync(v->src, source, (size_t)v->m * 4, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipMemcpyAsync(v->tgt, target, (size_t)v->m * 4, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipMemcpyAsync(v->lower, lower, (size_t)v->m * 8, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipMemcpyAsync(v->upper, upper, (size_t)v->m * 8, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipMemcpyAsync(v->cost, cost, (size_t)v->m * 8, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipMemcpyAsync(v->supply, supply, (size_t)v->n * 8, hipMemcpyHostToDevice, v->stream));
    }
    if (sol_args) {
        HIP_TRY(hipMemcpyAsync(v->flow, flow, (size_t)v->m * 8, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipMemcpyAsync(v->pi, pi, (size_t)v->n * 8, hipMemcpyHostToDevice, v->stream));
    }
    HIP_TRY(hipStreamSynchronize(v->stream));     // the host arrays are only borrowed for the call
    v->have_network |= net_args != 0;
    v->have_solution |= sol_args != 0;
    return MCF_OK;
}

int mcf_validator_run(mcf_validator *v, int32_t supply_type, int64_t reported_cost, mcf_validation *out)
{
    if (!v || !out) return mcf::fail(MCF_ERR_INVALID, "mcf_validator_run: null argument");
    if (supply_type < MCF_SUPPLY_GEQ || supply_type > MCF_SUPPLY_EQ) return mcf::fail(MCF_ERR_INVALID, "supply type %d", supply_type);
    if (!v->have_network || !v->have_solution) return mcf::fail(MCF_ERR_STATE, "mcf_validator_upload has not supplied the network and the solution");
    HIP_TRY(hipSetDevice(v->device));
    const int per_group = kValThreads * kValArcsPerThread;
    const int env_groups = getenv("MCF_VAL_G") ? atoi(getenv("MCF_VAL_G")) : 0;      // experiments only
    const int max_groups = env_groups > 0 && env_groups <= kValMaxGroups ? env_groups : kValMaxGroups;
    const int arc_groups = std::max(1, std::min(max_groups, (v->m + per_group - 1) / per_group));
    const int node_groups = std::max(1, std::min(max_groups, (v->n + kValThreads - 1) / kValThreads));
    HIP_TRY(hipEventRecord(v->ev0, v->stream));
    HIP_TRY(hipMemsetAsync(v->net, 0, 2 * ((size_t)v->n + 1) * 8, v->stream));
    const int skip = getenv("MCF_VAL_SKIP") ? atoi(getenv("MCF_VAL_SKIP")) & 3 : 0;     // experiments only
    auto arcs_kernel = skip == 0 ? validate_arcs<0> : (skip == 1 ? validate_arcs<1> : (skip == 2 ? validate_arcs<2> : validate_arcs<3>));
    hipLaunchKernelGGL(arcs_kernel, dim3(arc_groups), dim3(kValThreads), 0, v->stream, v->src, v->tgt, v->lower, v->upper, v->cost, v->flow,
                       v->pi, v->net, v->adj, v->m, v->partials);
    hipLaunchKernelGGL(validate_nodes, dim3(node_groups), dim3(kValThreads), 0, v->stream, v->supply, v->pi, v->net, v->adj, v->n, supply_type,
                       v->partials + arc_groups);
    hipLaunchKernelGGL(validate_fold, dim3(1), dim3(kValThreads), 0, v->stream, v->partials, arc_groups, node_groups, v->res);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(v->ev1, v->stream));
    HIP_TRY(hipMemcpyAsync(v->h_res, v->res, sizeof(ValResult), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, v->ev0, v->ev1));
    const ValResult &r = *v->h_res;
    memset(out, 0, sizeof(*out));
    out->supply_type = supply_type;
    out->objective = (int64_t)r.objective;
    out->dual_cost = (int64_t)(r.dual_arcs + r.dual_nodes);
    for (int k = 0; k < MCF_VAL_KINDS; ++k) {
        out->errors[k] = (int64_t)r.count[k];
        out->first[k] = r.first[k] == kNoId ? -1 : (int64_t)r.first[k];
    }
    if (out->objective != reported_cost) { out->errors[MCF_VAL_OBJECTIVE] = 1; out->first[MCF_VAL_OBJECTIVE] = 0; }
    if (out->dual_cost != reported_cost) { out->errors[MCF_VAL_DUAL_COST] = 1; out->first[MCF_VAL_DUAL_COST] = 0; }
    out->valid = 1;
    for (int k = 0; k < MCF_VAL_KINDS; ++k) if (out->errors[k]) out->valid = 0;
    out->kernel_us = (double)ms * 1e3;
    out->algorithmic_bytes = 40ll * v->m + 40ll * v->n;
    return MCF_OK;
}

}  // extern "C"
