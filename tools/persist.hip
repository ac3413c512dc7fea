// tools/persist.hip -- exploration only: round trip of a RESIDENT kernel fed through a mailbox, versus one dispatch per request.
//   mailbox A: pinned host memory, polled by the GPU over PCIe
//   mailbox B: fine-grained VRAM made CPU-writable through the BAR (HSA memory pool + hsa_amd_agents_allow_access), polled on-device
// Every workgroup waits for a new sequence number, (optionally) scans its share of a 400k-arc instance, and answers with a 16-byte record
// in pinned host memory.  All spins are bounded by s_memrealtime so the grid always drains.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m_ = nullptr; hsa_status_string(s_, &m_); printf("%s: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)

struct alignas(16) Slot { long long c; unsigned p, tag; };
struct alignas(64) Mailbox { unsigned seq; unsigned quit; unsigned pad[14]; };

__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

template <bool SCAN>
__global__ __launch_bounds__(256) void resident(const Mailbox *mb, Slot *slots, const int *src, const int *tgt, const long long *cost,
                                                const signed char *state, const long long *pi, int count, unsigned idle_ticks, unsigned *exit_word, unsigned *dbg, int sleep_mode)
{
    __shared__ unsigned s_seq, s_quit;
    __shared__ long long wc[4]; __shared__ unsigned wp[4];
    unsigned last = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = realtime();
            unsigned seq, quit = 0;
            for (;;) {
                seq = __hip_atomic_load(&mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (seq != last) { quit = __hip_atomic_load(&mb->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                if (realtime() - t0 > idle_ticks) { quit = 2; break; }
                if (sleep_mode) __builtin_amdgcn_s_sleep(1);
            }
            s_seq = seq; s_quit = quit;
        }
        __syncthreads();
        const unsigned seq = s_seq, quit = s_quit;
        if (quit) { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(exit_word, quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        last = seq;
        const unsigned long long t_seen = realtime();
        const unsigned long long c_seen = __builtin_amdgcn_s_memtime();
        long long bc = 0; unsigned bp = 0xFFFFFFFFu;
        if (SCAN) {
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
            if ((threadIdx.x & 63) == 0) { wc[threadIdx.x >> 6] = bc; wp[threadIdx.x >> 6] = bp; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (blockIdx.x == 0) { dbg[seq & 2047] = (unsigned)(realtime() - t_seen); dbg[2048 + (seq & 2047)] = (unsigned)(__builtin_amdgcn_s_memtime() - c_seen); }
            if (SCAN) for (int w = 1; w < 4; ++w) { const bool take = wc[w] < bc || (wc[w] == bc && wp[w] < bp); bc = take ? wc[w] : bc; bp = take ? wp[w] : bp; }
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            v4u o; o.x = (unsigned)bc; o.y = (unsigned)((unsigned long long)bc >> 32); o.z = bp; o.w = seq;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(slots + blockIdx.x), "v"(o) : "memory");
        }
        __syncthreads();
    }
}

struct Pools { hsa_agent_t cpu{}, gpu{}; hsa_amd_memory_pool_t fine{}, coarse{}; bool have_cpu = false, have_gpu = false, have_fine = false; };
static hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    Pools *p = (Pools *)data;
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0; hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false; hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    printf("  gpu pool flags 0x%x alloc %d\n", flags, (int)alloc);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !p->have_fine) { p->fine = pool; p->have_fine = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t agent_cb(hsa_agent_t a, void *data)
{
    Pools *p = (Pools *)data;
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && !p->have_cpu) { p->cpu = a; p->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU && !p->have_gpu) { p->gpu = a; p->have_gpu = true; hsa_amd_agent_iterate_memory_pools(a, pool_cb, p); }
    return HSA_STATUS_SUCCESS;
}

static unsigned *g_dbg_host, *g_dbg_dev; static int g_sleep = 1;
template <bool SCAN>
static void run(const char *name, Mailbox *mb_host_view, const Mailbox *mb_dev_view, Slot *hslot, Slot *dslot, int grid,
                const int *dsrc, const int *dtgt, const long long *dcost, const signed char *dstate, const long long *dpi, int m, hipStream_t st,
                unsigned *exit_host, unsigned *exit_dev)
{
    mb_host_view->seq = 0; mb_host_view->quit = 0; *exit_host = 0;
    _mm_sfence();
    memset(hslot, 0, sizeof(Slot) * 4096);
    hipLaunchKernelGGL((resident<SCAN>), dim3(grid), dim3(256), 0, st, mb_dev_view, dslot, dsrc, dtgt, dcost, dstate, dpi, m, 100000000u /* 1 s */, exit_dev, g_dbg_dev, g_sleep);
    CK(hipGetLastError());
    const int reps = 3000;
    double rt = 0, worst = 0;
    for (int r = 1; r <= reps + 100; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        *(volatile unsigned *)&mb_host_view->seq = (unsigned)r;
        _mm_sfence();
        volatile Slot *q = hslot;
        bool ok = true;
        for (int g = 0; g < grid && ok; ++g) {
            unsigned long long spins = 0;
            while (q[g].tag != (unsigned)r) { _mm_pause(); if (++spins > 30000000ull) { printf("%s: timeout waiting for wg %d at r=%d\n", name, g, r); ok = false; break; } }
        }
        auto t1 = std::chrono::steady_clock::now();
        if (!ok) break;
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (r > 100) { rt += us; worst = us > worst ? us : worst; }
        // emulate the host's own work between requests (tree update): a short pause
        for (int k = 0; k < 200; ++k) _mm_pause();
    }
    *(volatile unsigned *)&mb_host_view->quit = 1;
    _mm_sfence();
    *(volatile unsigned *)&mb_host_view->seq = 0x7FFFFFFF;
    _mm_sfence();
    CK(hipStreamSynchronize(st));
    double dsum = 0, csum = 0; for (int k = 200; k < 2000; ++k) { dsum += g_dbg_host[k]; csum += g_dbg_host[2048 + k]; }
    printf("%-58s round trip avg %6.2f us  worst %7.2f us  (exit %u)  wg0 seen->done %.2f us, %.0f cycles (%.0f MHz)\n", name, rt / reps, worst, *exit_host, dsum / 1800 / 100.0, csum / 1800, csum / (dsum / 100.0));
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int m = 401408, n = 100001;
    std::vector<int> src(m), tgt(m); std::vector<long long> cost(m), pi(n); std::vector<signed char> state(m);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int i = 0; i < m; ++i) { src[i] = rnd() % n; tgt[i] = rnd() % n; cost[i] = (long long)(rnd() % 20001) - 10000; state[i] = (signed char)(rnd() % 3) - 1; }
    for (int i = 0; i < n; ++i) pi[i] = -(long long)(rnd() % 1000000000);
    int *dsrc, *dtgt; long long *dcost, *dpi; signed char *dstate; Slot *hslot, *dslot; unsigned *exit_host, *exit_dev;
    CK(hipMalloc(&dsrc, m * 4)); CK(hipMalloc(&dtgt, m * 4)); CK(hipMalloc(&dcost, m * 8)); CK(hipMalloc(&dpi, n * 8)); CK(hipMalloc(&dstate, m));
    CK(hipMemcpy(dsrc, src.data(), m * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtgt, tgt.data(), m * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcost, cost.data(), m * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dpi, pi.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dstate, state.data(), m, hipMemcpyHostToDevice));
    CK(hipHostMalloc(&hslot, sizeof(Slot) * 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&dslot, hslot, 0));
    CK(hipHostMalloc(&exit_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&exit_dev, exit_host, 0));

    CK(hipHostMalloc(&g_dbg_host, 4096 * 4, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&g_dbg_dev, g_dbg_host, 0));
    // mailbox A: pinned host memory
    Mailbox *mbA, *mbA_dev;
    CK(hipHostMalloc(&mbA, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&mbA_dev, mbA, 0));
    memset(mbA, 0, 4096);

    // mailbox B: fine-grained VRAM, CPU-writable through the BAR
    Mailbox *mbB = nullptr;
    HK(hsa_init());
    Pools p;
    HK(hsa_iterate_agents(agent_cb, &p));
    if (p.have_fine && p.have_cpu) {
        void *ptr = nullptr;
        hsa_status_t s = hsa_amd_memory_pool_allocate(p.fine, 4096, 0, &ptr);
        if (s == HSA_STATUS_SUCCESS) {
            hsa_agent_t both[2] = {p.cpu, p.gpu};
            s = hsa_amd_agents_allow_access(2, both, nullptr, ptr);
            if (s == HSA_STATUS_SUCCESS) { mbB = (Mailbox *)ptr; memset(mbB, 0, 4096); printf("BAR mailbox at %p\n", ptr); }
            else { const char *msg; hsa_status_string(s, &msg); printf("allow_access failed: %s\n", msg); }
        } else { const char *msg; hsa_status_string(s, &msg); printf("pool allocate failed: %s\n", msg); }
    } else printf("no fine-grained GPU pool / cpu agent found (fine %d cpu %d)\n", (int)p.have_fine, (int)p.have_cpu);

    for (int sleep : {1}) {
        g_sleep = sleep;
        printf("---- poll loop %s s_sleep\n", sleep ? "with" : "without");
        for (int grid : {1, 256}) {
            char nm[128];
            if (grid == 1) { snprintf(nm, sizeof nm, "pinned mailbox, no work, %d WG", grid); run<false>(nm, mbA, mbA_dev, hslot, dslot, grid, dsrc, dtgt, dcost, dstate, dpi, m, st, exit_host, exit_dev); }
            if (mbB) { snprintf(nm, sizeof nm, "BAR-VRAM mailbox, no work, %d WG", grid); run<false>(nm, mbB, mbB, hslot, dslot, grid, dsrc, dtgt, dcost, dstate, dpi, m, st, exit_host, exit_dev); }
        }
        for (int grid : {256, 392}) {
            char nm[128];
            if (mbB) { snprintf(nm, sizeof nm, "BAR-VRAM mailbox, 400k-arc scan, %d WG", grid); run<true>(nm, mbB, mbB, hslot, dslot, grid, dsrc, dtgt, dcost, dstate, dpi, m, st, exit_host, exit_dev); }
        }
    }
    return 0;
}
