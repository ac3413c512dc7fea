// Shared helpers of libmcf_hip.so (host side).
#pragma once

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

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

// default Block Search block size of the two reference implementations
int default_block_size(int search_arc_num, int semantics);

}  // namespace mcf
