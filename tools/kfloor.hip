// tools/kfloor.hip -- exploration only (not part of the library): where do the microseconds of a tiny dispatch go?
// Times kernel variants with hipExtLaunchKernelGGL start/stop events (dispatch begin -> end, like rocprof).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct alignas(16) Slot { long long c; unsigned p, tag; };
struct Big { int pad[320]; };   // ~1.3 KB of kernel arguments

__global__ void k_empty() {}
__global__ void k_bigarg(Big b) { if (b.pad[0] == 12345) __builtin_trap(); }
__global__ void k_slot(Slot *s, unsigned tag) { if (threadIdx.x == 0) { uint4 o; o.x = 1; o.y = 0; o.z = blockIdx.x; o.w = tag; *(uint4 *)(s + blockIdx.x) = o; } }

// stream 17 B/arc + gathers, reduce with shuffles, write slot
template <bool GATHER, bool SHFL>
__global__ __launch_bounds__(256) void k_scan(const int *src, const int *tgt, const long long *cost, const signed char *state,
                                               const long long *pi, int count, Slot *s, unsigned tag)
{
    long long bc = 0; unsigned bp = 0xFFFFFFFFu;
    for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < count; i0 += gridDim.x * 1024) {
        const unsigned st4 = *(const unsigned *)(state + i0);
        const int4 s4 = *(const int4 *)(src + i0), t4 = *(const int4 *)(tgt + i0);
        const longlong2 c0 = *(const longlong2 *)(cost + i0), c1 = *(const longlong2 *)(cost + i0 + 2);
        const int sv[4] = {s4.x, s4.y, s4.z, s4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
        const long long cv[4] = {c0.x, c0.y, c1.x, c1.y};
        long long ps[4], pt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ps[j] = GATHER ? pi[sv[j]] : sv[j]; pt[j] = GATHER ? pi[tv[j]] : tv[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int st = (int)(signed char)(st4 >> (8 * j));
            const long long d = cv[j] + ps[j] - pt[j];
            const long long rc = st > 0 ? d : (st < 0 ? -d : 0);
            if (rc < bc) { bc = rc; bp = i0 + j; }
        }
    }
    if (SHFL) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const long long oc = __shfl_xor(bc, off, 64); const unsigned op = __shfl_xor(bp, off, 64);
            const bool take = oc < bc || (oc == bc && op < bp);
            bc = take ? oc : bc; bp = take ? op : bp;
        }
    }
    __shared__ long long wc[4]; __shared__ unsigned wp[4];
    if ((threadIdx.x & 63) == 0) { wc[threadIdx.x >> 6] = bc; wp[threadIdx.x >> 6] = bp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { const bool take = wc[w] < bc || (wc[w] == bc && wp[w] < bp); bc = take ? wc[w] : bc; bp = take ? wp[w] : bp; }
        uint4 o; o.x = (unsigned)bc; o.y = (unsigned)((unsigned long long)bc >> 32); o.z = bp; o.w = tag;
        *(uint4 *)(s + blockIdx.x) = o;
    }
}

// two-level: workgroups publish partials in device memory, the last arriver of each of 8 groups reduces its group, the last
// group leader reduces the 8 group results and writes ONE host slot
__global__ __launch_bounds__(256) void k_scan2(const int *src, const int *tgt, const long long *cost, const signed char *state,
                                               const long long *pi, int count, Slot *host_slot, unsigned tag,
                                               unsigned long long *part_c, unsigned *part_p, unsigned *counters)
{
    long long bc = 0; unsigned bp = 0xFFFFFFFFu;
    for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < count; i0 += gridDim.x * 1024) {
        const unsigned st4 = *(const unsigned *)(state + i0);
        const int4 s4 = *(const int4 *)(src + i0), t4 = *(const int4 *)(tgt + i0);
        const longlong2 c0 = *(const longlong2 *)(cost + i0), c1 = *(const longlong2 *)(cost + i0 + 2);
        const int sv[4] = {s4.x, s4.y, s4.z, s4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
        const long long cv[4] = {c0.x, c0.y, c1.x, c1.y};
        long long ps[4], pt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ps[j] = pi[sv[j]]; pt[j] = pi[tv[j]]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int st = (int)(signed char)(st4 >> (8 * j));
            const long long d = cv[j] + ps[j] - pt[j];
            const long long rc = st > 0 ? d : (st < 0 ? -d : 0);
            if (rc < bc) { bc = rc; bp = i0 + j; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long oc = __shfl_xor(bc, off, 64); const unsigned op = __shfl_xor(bp, off, 64);
        const bool take = oc < bc || (oc == bc && op < bp);
        bc = take ? oc : bc; bp = take ? op : bp;
    }
    __shared__ long long wc[4]; __shared__ unsigned wp[4]; __shared__ int role;
    if ((threadIdx.x & 63) == 0) { wc[threadIdx.x >> 6] = bc; wp[threadIdx.x >> 6] = bp; }
    __syncthreads();
    const int G = 8, grp = blockIdx.x % G, per = (gridDim.x + G - 1 - grp) / G;   // members of my group
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { const bool take = wc[w] < bc || (wc[w] == bc && wp[w] < bp); bc = take ? wc[w] : bc; bp = take ? wp[w] : bp; }
        __hip_atomic_store(part_c + blockIdx.x, (unsigned long long)bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part_p + blockIdx.x, bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(counters + 16 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = (t == (unsigned)per - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!role) return;
    // group reducer: one wave reads the group's partials (sc1 loads)
    if (threadIdx.x < 64) {
        long long c = 0; unsigned p = 0xFFFFFFFFu;
        for (int k = threadIdx.x; k < per; k += 64) {
            const int b = grp + k * G;
            const long long oc = (long long)__hip_atomic_load(part_c + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned op = __hip_atomic_load(part_p + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool take = oc < c || (oc == c && op < p); c = take ? oc : c; p = take ? op : p;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const long long oc = __shfl_xor(c, off, 64); const unsigned op = __shfl_xor(p, off, 64);
            const bool take = oc < c || (oc == c && op < p); c = take ? oc : c; p = take ? op : p;
        }
        if (threadIdx.x == 0) {
            __hip_atomic_store(part_c + 4096 + grp, (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part_p + 4096 + grp, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            counters[16 * grp] = 0;   // reset for the next dispatch (kernel boundary orders it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(counters + 16 * G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == G - 1) {
                long long fc = 0; unsigned fp = 0xFFFFFFFFu;
                for (int g = 0; g < G; ++g) {
                    const long long oc = (long long)__hip_atomic_load(part_c + 4096 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned op = __hip_atomic_load(part_p + 4096 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool take = oc < fc || (oc == fc && op < fp); fc = take ? oc : fc; fp = take ? op : fp;
                }
                counters[16 * G] = 0;
                uint4 o; o.x = (unsigned)fc; o.y = (unsigned)((unsigned long long)fc >> 32); o.z = fp; o.w = tag;
                *(uint4 *)host_slot = o;
            }
        }
    }
}

template <typename F>
static void timeit(const char *name, hipStream_t st, F launch, volatile Slot *poll, int npoll)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double sum = 0, mn = 1e9, rt = 0;
    const int reps = 300;
    for (int r = 0; r < reps + 20; ++r) {
        const unsigned tag = 1000 + r;
        auto t0 = std::chrono::steady_clock::now();
        launch(a, b, tag);
        if (poll) { for (int g = 0; g < npoll; ++g) while (poll[g].tag != tag) _mm_pause(); }
        else CK(hipStreamSynchronize(st));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r >= 20) { sum += ms * 1e3; mn = ms * 1e3 < mn ? ms * 1e3 : mn; rt += std::chrono::duration<double, std::micro>(t1 - t0).count(); }
    }
    printf("%-44s kernel avg %7.2f us  min %7.2f us   host round trip %7.2f us\n", name, sum / reps, mn, rt / reps);
}

int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int m = 401408, n = 100001;   // 392 tiles of 1024
    std::vector<int> src(m), tgt(m); std::vector<long long> cost(m), pi(n); std::vector<signed char> state(m);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int i = 0; i < m; ++i) { src[i] = rnd() % n; tgt[i] = rnd() % n; cost[i] = (long long)(rnd() % 20001) - 10000; state[i] = (signed char)(rnd() % 3) - 1; }
    for (int i = 0; i < n; ++i) pi[i] = -(long long)(rnd() % 1000000000);
    int *dsrc, *dtgt; long long *dcost, *dpi; signed char *dstate; Slot *hslot, *dslot_host, *dslot_dev;
    CK(hipMalloc(&dsrc, m * 4)); CK(hipMalloc(&dtgt, m * 4)); CK(hipMalloc(&dcost, m * 8)); CK(hipMalloc(&dpi, n * 8)); CK(hipMalloc(&dstate, m));
    CK(hipMemcpy(dsrc, src.data(), m * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtgt, tgt.data(), m * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcost, cost.data(), m * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dpi, pi.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dstate, state.data(), m, hipMemcpyHostToDevice));
    CK(hipHostMalloc(&hslot, sizeof(Slot) * 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&dslot_host, hslot, 0));
    CK(hipMalloc(&dslot_dev, sizeof(Slot) * 4096));
    Big big{};

    timeit("empty 1 WG (sync)", st, [&](hipEvent_t a, hipEvent_t b, unsigned) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, a, b, 0); }, nullptr, 0);
    timeit("empty 392 WG (sync)", st, [&](hipEvent_t a, hipEvent_t b, unsigned) { hipExtLaunchKernelGGL(k_empty, dim3(392), dim3(256), 0, st, a, b, 0); }, nullptr, 0);
    timeit("1.3KB args 392 WG (sync)", st, [&](hipEvent_t a, hipEvent_t b, unsigned) { hipExtLaunchKernelGGL(k_bigarg, dim3(392), dim3(256), 0, st, a, b, 0, big); }, nullptr, 0);
    timeit("slot->host 1 WG (poll)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_slot, dim3(1), dim3(256), 0, st, a, b, 0, dslot_host, t); }, hslot, 1);
    timeit("slot->host 392 WG (poll)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_slot, dim3(392), dim3(256), 0, st, a, b, 0, dslot_host, t); }, hslot, 392);
    timeit("slot->device 392 WG (sync)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_slot, dim3(392), dim3(256), 0, st, a, b, 0, dslot_dev, t); }, nullptr, 0);
    for (int grid : {392, 196, 98}) {
        char nm[96];
        snprintf(nm, sizeof nm, "scan gather+shfl ->host, %d WG (poll)", grid);
        timeit(nm, st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL((k_scan<true, true>), dim3(grid), dim3(256), 0, st, a, b, 0, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, t); }, hslot, grid);
        snprintf(nm, sizeof nm, "scan gather+shfl ->device, %d WG (sync)", grid);
        timeit(nm, st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL((k_scan<true, true>), dim3(grid), dim3(256), 0, st, a, b, 0, dsrc, dtgt, dcost, dstate, dpi, m, dslot_dev, t); }, nullptr, 0);
        snprintf(nm, sizeof nm, "scan nogather+shfl ->host, %d WG (poll)", grid);
        timeit(nm, st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL((k_scan<false, true>), dim3(grid), dim3(256), 0, st, a, b, 0, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, t); }, hslot, grid);
        snprintf(nm, sizeof nm, "scan gather, no shfl ->host, %d WG (poll)", grid);
        timeit(nm, st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL((k_scan<true, false>), dim3(grid), dim3(256), 0, st, a, b, 0, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, t); }, hslot, grid);
    }
    timeit("slot->host 392 WG (sync, no poll)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_slot, dim3(392), dim3(256), 0, st, a, b, 0, dslot_host, t); }, nullptr, 0);
    timeit("slot->host 98 WG (poll)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_slot, dim3(98), dim3(256), 0, st, a, b, 0, dslot_host, t); }, hslot, 98);
    timeit("slot->host 32 WG (poll)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_slot, dim3(32), dim3(256), 0, st, a, b, 0, dslot_host, t); }, hslot, 32);
    {
        unsigned long long *part_c; unsigned *part_p, *counters;
        CK(hipMalloc(&part_c, 8 * 8192)); CK(hipMalloc(&part_p, 4 * 8192)); CK(hipMalloc(&counters, 4 * 1024)); CK(hipMemset(counters, 0, 4096));
        timeit("scan two-level device reduce, 392 WG, 1 slot (poll)", st, [&](hipEvent_t a, hipEvent_t b, unsigned t) { hipExtLaunchKernelGGL(k_scan2, dim3(392), dim3(256), 0, st, a, b, 0, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, t, part_c, part_p, counters); }, hslot, 1);
        // check the answer against the plain version
        hipLaunchKernelGGL((k_scan<true, true>), dim3(392), dim3(256), 0, st, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host + 1024, 7u);
        CK(hipStreamSynchronize(st));
        long long bc = 0; unsigned bp = 0xFFFFFFFFu;
        for (int g = 0; g < 392; ++g) { const Slot &q = hslot[1024 + g]; if (q.c < bc || (q.c == bc && q.p < bp)) { bc = q.c; bp = q.p; } }
        hipLaunchKernelGGL(k_scan2, dim3(392), dim3(256), 0, st, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, 9u, part_c, part_p, counters);
        CK(hipStreamSynchronize(st));
        printf("two-level answer (%lld, %u) vs flat (%lld, %u) %s\n", hslot[0].c, hslot[0].p, bc, bp, (hslot[0].c == bc && hslot[0].p == bp) ? "OK" : "MISMATCH");
        double rt = 0; const int reps = 2000;
        for (int r = 0; r < reps + 50; ++r) {
            const unsigned tag = 150000 + r;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_scan2, dim3(392), dim3(256), 0, st, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, tag, part_c, part_p, counters);
            while (((volatile Slot *)hslot)[0].tag != tag) _mm_pause();
            auto t2 = std::chrono::steady_clock::now();
            if (r >= 50) rt += std::chrono::duration<double, std::micro>(t2 - t0).count();
        }
        printf("plain launch two-level scan 392 WG + poll 1 slot: host round trip %.2f us\n", rt / reps);
    }
    // MLP-friendly polling of 392 slots: sweep all lines, no per-slot spinning
    {
        double rt = 0; const int reps = 2000;
        for (int r = 0; r < reps + 50; ++r) {
            const unsigned tag = 250000 + r;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL((k_scan<true, true>), dim3(392), dim3(256), 0, st, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, tag);
            volatile Slot *q = hslot;
            for (;;) {
                unsigned ok = 1;
                for (int g = 0; g < 392; ++g) ok &= (q[g].tag == tag);
                if (ok) break;
                _mm_pause();
            }
            auto t2 = std::chrono::steady_clock::now();
            if (r >= 50) rt += std::chrono::duration<double, std::micro>(t2 - t0).count();
        }
        printf("plain launch scan 392 WG + sweep-poll: host round trip %.2f us\n", rt / reps);
    }
    // the same launches WITHOUT events: host round trip only
    {
        double rt = 0; const int reps = 2000;
        for (int r = 0; r < reps + 50; ++r) {
            const unsigned tag = 50000 + r;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL((k_scan<true, true>), dim3(392), dim3(256), 0, st, dsrc, dtgt, dcost, dstate, dpi, m, dslot_host, tag);
            auto t1 = std::chrono::steady_clock::now();
            for (int g = 0; g < 392; ++g) while (((volatile Slot *)hslot)[g].tag != tag) _mm_pause();
            auto t2 = std::chrono::steady_clock::now();
            if (r >= 50) { rt += std::chrono::duration<double, std::micro>(t2 - t0).count(); }
            (void)t1;
        }
        printf("plain launch scan 392 WG + poll: host round trip %.2f us\n", rt / reps);
        rt = 0;
        for (int r = 0; r < reps + 50; ++r) {
            const unsigned tag = 90000 + r;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_slot, dim3(1), dim3(64), 0, st, dslot_host, tag);
            while (((volatile Slot *)hslot)[0].tag != tag) _mm_pause();
            auto t2 = std::chrono::steady_clock::now();
            if (r >= 50) rt += std::chrono::duration<double, std::micro>(t2 - t0).count();
        }
        printf("plain launch 1-wave slot kernel + poll: host round trip %.2f us\n", rt / reps);
        double lt = 0;
        for (int r = 0; r < reps; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
            auto t1 = std::chrono::steady_clock::now();
            lt += std::chrono::duration<double, std::micro>(t1 - t0).count();
            if ((r & 63) == 63) CK(hipStreamSynchronize(st));
        }
        printf("hipLaunchKernelGGL call itself (empty kernel, queue not full): %.2f us\n", lt / reps);
    }
    return 0;
}
