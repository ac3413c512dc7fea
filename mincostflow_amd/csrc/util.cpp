// Error reporting and small shared helpers of libmcf_hip.so.
#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace mcf {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

// BSPO.cs:27-28 (OPTIMIZED): max((int)sqrt(m_s), MIN_BLOCK_SIZE = 10)  [NS.cs:19]
// NS.cs:1304-1336 (PLAIN) with the default OptimizationConfig (OptimizationTypes.cs:24-38: no flags,
// MinBlockSize 25, MinBlockSizeRatio 0.125), i.e. SetAutoConfiguration(false); the auto-configuration
// heuristics (Analysis/*) are host-side policy outside this path and callers pass block_size explicitly
// to reproduce them.
int default_block_size(int search_arc_num, int semantics)
{
    const int base = (int)std::sqrt((double)search_arc_num);
    if (semantics == MCF_SEM_OPTIMIZED) return base > 10 ? base : 10;
    int dyn_min = (int)(base * 0.125);
    if (dyn_min < 25) dyn_min = 25;
    return base > dyn_min ? base : dyn_min;
}

}  // namespace mcf

extern "C" {

const char *mcf_last_error(void) { return mcf::g_error; }
const char *mcf_version(void) { return "mcf_hip 0.1 (gfx950)"; }

}  // extern "C"
