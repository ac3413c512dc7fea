// Error reporting and small shared helpers of libmcf_hip.so.
#include <cmath>
#include <cstdarg>
#include <algorithm>
#include <cstdio>
#include <vector>

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

// new OptimizationConfig(): OptimizationTypes.cs:24-38
void mcf_block_config_default(mcf_block_config *c)
{
    if (!c) return;
    c->flags = MCF_OPT_NONE;
    c->min_block_size = 25;
    c->max_block_size = 100;
    c->consecutive_hits_before_adapt = 3;
    c->min_block_size_ratio = 0.125;
    c->block_size_growth_factor = 1.2;
    c->block_size_shrink_factor = 0.8;
    c->low_hit_rate_threshold = 0.05;
    c->high_hit_rate_threshold = 0.3;
}

// ProblemAnalyzer.Analyze (Lemon/ProblemAnalyzer.cs:21-106: density, node degrees) followed by OptimizationSelector.SelectConfiguration
// (Analysis/OptimizationSelector.cs:14-95), keeping the fields the plain BlockSearchPivot reads.  Same doubles in the same order.
int mcf_block_config_auto(mcf_block_config *c, int32_t n, int32_t m, const int32_t *source, const int32_t *target)
{
    if (!c || n < 0 || m < 0 || (m && (!source || !target))) return mcf::fail(MCF_ERR_INVALID, "mcf_block_config_auto: bad arguments");
    mcf_block_config_default(c);
    const int64_t max_possible = (int64_t)n * (n - 1);                                   // ProblemAnalyzer.cs:35-36
    const double density = max_possible > 0 ? (double)m / (double)max_possible : 0;
    std::vector<int32_t> degree((size_t)n, 0);                                            // :65-78 out-degree + in-degree
    for (int e = 0; e < m; ++e) {
        if ((unsigned)source[e] >= (unsigned)n || (unsigned)target[e] >= (unsigned)n) return mcf::fail(MCF_ERR_INVALID, "arc %d: end point out of range", e);
        degree[source[e]]++;
        degree[target[e]]++;
    }
    int32_t total = 0;                                                                    // `int totalDegree` (wraps like C#'s unchecked int)
    for (int v = 0; v < n; ++v) total = (int32_t)((uint32_t)total + (uint32_t)degree[v]);
    const double avg = n > 0 ? (double)total / n : 0;                                    // :80
    double variance = 0;                                                                  // :83-92
    if (n > 0) {
        for (int v = 0; v < n; ++v) { const double diff = degree[v] - avg; variance += diff * diff; }
        variance /= n;
    }
    const double degree_cv = avg > 0 ? std::sqrt(variance) / avg : 0;                     // :97
    const bool dense = density > 0.01 || m > 10000;                                       // :58-59
    const bool sparse = density < 0.005;                                                  // :60
    int flags = MCF_OPT_NONE;
    if (dense) { flags |= MCF_OPT_SMALL_BLOCKS_FOR_DENSE; c->min_block_size = 10; c->max_block_size = 50; }      // OptimizationSelector.cs:20-32
    else { c->min_block_size = 25; c->max_block_size = 100; }
    if (degree_cv > 0.5) {                                                                // :35-46
        flags |= MCF_OPT_ADAPTIVE_BLOCK_SIZE;
        c->block_size_growth_factor = 1.3;
        c->block_size_shrink_factor = 0.7;
        c->consecutive_hits_before_adapt = 2;
    } else if (degree_cv > 0.3) {
        flags |= MCF_OPT_ADAPTIVE_BLOCK_SIZE;
    }
    if (sparse && m < 50000) flags |= MCF_OPT_REDUCED_COST_CACHING;                       // :49-52
    c->low_hit_rate_threshold = m > 10000 ? 0.03 : 0.05;                                  // :76-77
    c->high_hit_rate_threshold = m > 10000 ? 0.25 : 0.3;
    c->min_block_size_ratio = m > 100000 ? 0.0625 : (m > 10000 ? 0.125 : 0.25);           // :80-91
    c->flags = flags;
    return MCF_OK;
}
const char *mcf_version(void) { return "mcf_hip 0.2 (gfx950)"; }


// BlockSearchPivot constructor, NS.cs:1304-1336
int mcf_block_initial_size(const mcf_block_config *c, int32_t m_s, int32_t graph_node_count, int32_t *block_size, int32_t *dynamic_min)
{
    if (!c || m_s < 0 || graph_node_count < 0 || !block_size || !dynamic_min) return mcf::fail(MCF_ERR_INVALID, "mcf_block_initial_size: bad arguments");
    const int base = (int)std::sqrt((double)m_s);
    const int dyn_min = std::max(c->min_block_size, (int)(base * c->min_block_size_ratio));
    int b = base;
    if (c->flags & MCF_OPT_SMALL_BLOCKS_FOR_DENSE) {
        const double density = (double)m_s / (double)graph_node_count;      // C# double division: n = 0 gives infinity (NaN for 0 / 0)
        if (density > 10) b = std::min(50, base / 4);
    }
    *block_size = std::max(b, dyn_min);
    *dynamic_min = dyn_min;
    return MCF_OK;
}

// NS.cs:1400-1438, same doubles, same truncations
int mcf_block_adapt(const mcf_block_config *c, int32_t dynamic_min, int64_t arcs_checked, int32_t *block_size, int32_t counters[2])
{
    if (!c || !block_size || !counters) return mcf::fail(MCF_ERR_INVALID, "mcf_block_adapt: bad arguments");
    if (!(c->flags & MCF_OPT_ADAPTIVE_BLOCK_SIZE)) return MCF_OK;
    const double hit_rate = arcs_checked > 0 ? 1.0 / (double)arcs_checked : 0;
    if (hit_rate < c->low_hit_rate_threshold) {
        counters[1] = 0;
        if (++counters[0] >= c->consecutive_hits_before_adapt) {
            const int smaller = (int)(*block_size * c->block_size_shrink_factor);
            *block_size = std::max(dynamic_min, smaller);
            counters[0] = 0;
        }
    } else if (hit_rate > c->high_hit_rate_threshold) {
        counters[0] = 0;
        if (++counters[1] >= c->consecutive_hits_before_adapt) {
            const int larger = (int)(*block_size * c->block_size_growth_factor);
            *block_size = std::min(c->max_block_size, larger);
            counters[1] = 0;
        }
    } else {
        counters[0] = counters[1] = 0;
    }
    return MCF_OK;
}

}  // extern "C"

// A SIGABRT (glibc's heap checks, a failed assertion inside a library underneath, std::terminate) leaves the C-level call stack in the file
// that MCF_ABORT_TRACE_FILE names before the process dies -- test runners capture stderr, where the aborting party's own message goes.
// Nothing is installed without that variable (tests/conftest.py sets it).
#include <cstring>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
namespace {
char g_trace_path[512];
struct sigaction g_abort_before;      // e.g. Python's faulthandler: it gets the signal next
void abort_trace(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const int fd = open(g_trace_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) { backtrace_symbols_fd(frames, n, fd); (void)!write(fd, "----\n", 5); close(fd); }
    backtrace_symbols_fd(frames, n, 2);
    sigaction(sig, &g_abort_before, nullptr);
    raise(sig);
}
__attribute__((constructor)) void install_abort_trace()
{
    const char *p = getenv("MCF_ABORT_TRACE_FILE");
    if (!p || !p[0] || strlen(p) >= sizeof(g_trace_path)) return;
    strcpy(g_trace_path, p);
    void *warm[4];
    (void)backtrace(warm, 4);             // loads libgcc now, not inside the handler
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = abort_trace;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, &g_abort_before);
}
}  // namespace
