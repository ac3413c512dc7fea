// Shared helpers of libmcf_hip.so (host side).
#pragma once

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include <sys/mman.h>

#include "../../include/mcf_hip.h"

namespace mcf {

// thread-local error text behind mcf_last_error()
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

inline double now_ns()
{
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

// SplitMix64: the generators' only source of randomness
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform integer in [lo, hi] (multiply-shift; no modulo bias worth caring about at these ranges)
    int64_t range(int64_t lo, int64_t hi)
    {
        const uint64_t span = (uint64_t)(hi - lo) + 1;
        return lo + (int64_t)(((unsigned __int128)next() * span) >> 64);
    }
};

// Host arrays of the graph's size are read at random (tree links, potentials, the candidate cache's mirrors): with 4 KB pages every access
// beyond a few MB misses the TLB as well.  Big blocks are 2 MB-aligned and marked for transparent huge pages before their first touch
// (the GPU boxes run with transparent_hugepage=madvise).
inline void *huge_block(size_t bytes)
{
    // 2 MB-aligned region marked for huge pages; the array starts a few cache lines into it, by a different amount for every block, so that
    // equal indices of different arrays do not all fall into the same cache sets; the region's address sits in front of the array
    static std::atomic<unsigned> turn{0};
    const size_t stagger = 64 + (size_t)(turn.fetch_add(1, std::memory_order_relaxed) % 31) * 4160;
    const size_t len = (bytes + stagger + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void *base = aligned_alloc((size_t)2 << 20, len);
    if (!base) return nullptr;
    static const bool off = getenv("MCF_HUGEPAGES") && getenv("MCF_HUGEPAGES")[0] == '0';      // measurement aid
    if (!off) (void)madvise(base, len, MADV_HUGEPAGE);
    void *p = (char *)base + stagger;
    ((void **)p)[-1] = base;
    return p;
}
constexpr size_t kHugeBlockMin = (size_t)2 << 20;

template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        void *p = bytes >= kHugeBlockMin ? huge_block(bytes) : malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t n) { if (n * sizeof(T) >= kHugeBlockMin) free(((void **)p)[-1]); else free(p); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using hvec = std::vector<T, HugeAlloc<T>>;

// default Block Search block size of the two reference implementations
int default_block_size(int search_arc_num, int semantics);

}  // namespace mcf
