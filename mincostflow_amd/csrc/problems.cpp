// problems.cpp -- build-owned instance sources: NETGEN-like and assignment generators, DIMACS min reader/writer.
//
// The reference ships NETGEN *outputs* (src/MinCostFlow.Problems/Resources/netgen/*.min) but no generator, and its
// own generators depend on .NET's System.Random, so the instances named in BASELINE.json
// are produced here from SplitMix64 and are bit-reproducible on any machine.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "common.h"

namespace {

int alloc_problem(mcf_problem *p, int n, int64_t m)
{
    memset(p, 0, sizeof(*p));
    p->node_count = n;
    p->arc_count = (int32_t)m;
    const size_t mm = (size_t)std::max<int64_t>(m, 1), nn = (size_t)std::max(n, 1);
    p->source = (int32_t *)calloc(mm, sizeof(int32_t));
    p->target = (int32_t *)calloc(mm, sizeof(int32_t));
    p->lower = (int64_t *)calloc(mm, sizeof(int64_t));
    p->upper = (int64_t *)calloc(mm, sizeof(int64_t));
    p->cost = (int64_t *)calloc(mm, sizeof(int64_t));
    p->supply = (int64_t *)calloc(nn, sizeof(int64_t));
    if (!p->source || !p->target || !p->lower || !p->upper || !p->cost || !p->supply) {
        mcf_problem_free(p);
        return mcf::fail(MCF_ERR_INVALID, "out of memory for %d nodes / %lld arcs", n, (long long)m);
    }
    return MCF_OK;
}

// `total` split into `parts` positive integers
void split_positive(mcf::SplitMix64 &rng, int64_t total, int parts, std::vector<int64_t> &out)
{
    out.assign(parts, 1);
    int64_t rest = total - parts;
    for (int i = 0; i < parts - 1 && rest > 0; ++i) {
        const int64_t avg2 = 2 * rest / (parts - i);
        const int64_t take = std::min<int64_t>(rest, rng.range(0, std::max<int64_t>(avg2, 0)));
        out[i] += take;
        rest -= take;
    }
    out[parts - 1] += rest;
}

struct ArcRec { int32_t u, v; int64_t cap, cost; };

}  // namespace

extern "C" {

void mcf_problem_free(mcf_problem *p)
{
    if (!p) return;
    free(p->source); free(p->target); free(p->lower); free(p->upper); free(p->cost); free(p->supply);
    memset(p, 0, sizeof(*p));
}

int mcf_gen_netgen_like(mcf_problem *out, uint64_t seed, int32_t n, int32_t m, int32_t n_src, int32_t n_snk,
                        int64_t min_cost, int64_t max_cost, int64_t min_cap, int64_t max_cap)
{
    if (!out) return mcf::fail(MCF_ERR_INVALID, "null argument");
    if (n_src < 1 || n_snk < 1 || n_src + n_snk > n || min_cost > max_cost || min_cap < 1 || min_cap > max_cap)
        return mcf::fail(MCF_ERR_INVALID, "mcf_gen_netgen_like: bad parameters");
    const int n_trans = n - n_src - n_snk, n_pairs = std::max(n_src, n_snk);
    const int64_t skeleton = (int64_t)n_trans + n_pairs;
    if (m < skeleton) return mcf::fail(MCF_ERR_INVALID, "need at least %lld arcs for the skeleton", (long long)skeleton);
    mcf::SplitMix64 rng(seed);
    const int64_t total = 1000 * (int64_t)n_src;
    if (total < n_pairs) return mcf::fail(MCF_ERR_INVALID, "too many sinks for the total supply");

    // sources 0..n_src-1, sinks n-n_snk..n-1, transshipment nodes in between
    std::vector<int32_t> perm_s(n_src), perm_t(n_snk);
    std::iota(perm_s.begin(), perm_s.end(), 0);
    std::iota(perm_t.begin(), perm_t.end(), n - n_snk);
    for (int i = n_src - 1; i > 0; --i) std::swap(perm_s[i], perm_s[rng.range(0, i)]);
    for (int i = n_snk - 1; i > 0; --i) std::swap(perm_t[i], perm_t[rng.range(0, i)]);
    // (source, sink) pairs: every source and every sink occurs at least once
    std::vector<std::vector<int32_t>> sinks_of(n_src);
    for (int k = 0; k < n_pairs; ++k) sinks_of[perm_s[k % n_src]].push_back(perm_t[k % n_snk]);
    // supplies: positive, sum = total, each source at least one unit per sink it feeds
    std::vector<int64_t> supply_of;
    {
        std::vector<int64_t> extra;
        split_positive(rng, total - n_pairs + n_src, n_src, extra);
        supply_of.resize(n_src);
        for (int i = 0; i < n_src; ++i) supply_of[i] = extra[i] - 1 + (int64_t)sinks_of[i].size();
    }
    // every transshipment node joins the chain of a random source
    std::vector<std::vector<int32_t>> chain(n_src);
    for (int t = 0; t < n_trans; ++t) chain[rng.range(0, n_src - 1)].push_back(n_src + t);

    std::vector<ArcRec> arcs;
    arcs.reserve(m);
    std::vector<int64_t> demand(n, 0), parts;
    for (int i = 0; i < n_src; ++i) {
        int32_t at = i;
        for (int32_t t : chain[i]) { arcs.push_back({at, t, supply_of[i], max_cost}); at = t; }   // skeleton: max cost, capacitated
        split_positive(rng, supply_of[i], (int)sinks_of[i].size(), parts);
        for (size_t k = 0; k < sinks_of[i].size(); ++k) {
            arcs.push_back({at, sinks_of[i][k], parts[k], max_cost});
            demand[sinks_of[i][k]] += parts[k];
        }
    }
    // the remaining arcs: tail is not a sink, head is not a source
    while ((int64_t)arcs.size() < m) {
        const int32_t u = (int32_t)rng.range(0, n - n_snk - 1), v = (int32_t)rng.range(n_src, n - 1);
        if (u == v) continue;
        arcs.push_back({u, v, rng.range(min_cap, max_cap), rng.range(min_cost, max_cost)});
    }
    // emit grouped by tail node (stable), like NETGEN's output files
    int rc = alloc_problem(out, n, m);
    if (rc) return rc;
    std::vector<int64_t> first(n + 1, 0);
    for (const ArcRec &a : arcs) first[a.u + 1]++;
    for (int v = 0; v < n; ++v) first[v + 1] += first[v];
    for (const ArcRec &a : arcs) {
        const int64_t e = first[a.u]++;
        out->source[e] = a.u; out->target[e] = a.v; out->lower[e] = 0; out->upper[e] = a.cap; out->cost[e] = a.cost;
    }
    for (int i = 0; i < n_src; ++i) out->supply[i] = supply_of[i];
    for (int v = n - n_snk; v < n; ++v) out->supply[v] = -demand[v];
    return MCF_OK;
}

int mcf_gen_assignment(mcf_problem *out, uint64_t seed, int32_t n, int64_t min_cost, int64_t max_cost)
{
    if (!out || n < 1 || min_cost > max_cost) return mcf::fail(MCF_ERR_INVALID, "mcf_gen_assignment: bad parameters");
    const int64_t m = (int64_t)n * n;
    if (m > INT32_MAX / 2) return mcf::fail(MCF_ERR_INVALID, "assignment too large");
    int rc = alloc_problem(out, 2 * n, m);
    if (rc) return rc;
    mcf::SplitMix64 rng(seed);
    int64_t e = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j, ++e) {
            out->source[e] = i; out->target[e] = n + j; out->lower[e] = 0; out->upper[e] = 1;
            out->cost[e] = rng.range(min_cost, max_cost);
        }
    for (int i = 0; i < n; ++i) { out->supply[i] = 1; out->supply[n + i] = -1; }
    return MCF_OK;
}

// p min N M / n id supply / a u v low cap cost, 1-based (DimacsReader.cs:36-147).  Values are taken as written,
// like the C# reader (LEMON's reader turns cap < low into "infinite", lemon/dimacs.h:178-181; callers wanting that pass
// MCF_INF_CAP themselves).
int mcf_dimacs_read(mcf_problem *out, const char *path)
{
    if (!out || !path) return mcf::fail(MCF_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "rb");
    if (!f) return mcf::fail(MCF_ERR_IO, "%s: %s", path, strerror(errno));
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)size + 1);
    if (size && fread(buf.data(), 1, (size_t)size, f) != (size_t)size) { fclose(f); return mcf::fail(MCF_ERR_IO, "%s: short read", path); }
    fclose(f);
    buf[size] = 0;
    bool have_p = false;
    int64_t arcs_seen = 0;
    char *p = buf.data(), *end = buf.data() + size;
    int line_no = 0;
    while (p < end) {
        char *eol = (char *)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        *eol = 0;
        ++line_no;
        char *q = p;
        while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
        if (*q == 'p') {
            int n = 0; long long m = 0; char kind[16] = "";
            if (sscanf(q, "p %15s %d %lld", kind, &n, &m) != 3 || strcmp(kind, "min") != 0) { mcf_problem_free(out); return mcf::fail(MCF_ERR_IO, "%s:%d: invalid problem line", path, line_no); }
            if (have_p) { mcf_problem_free(out); return mcf::fail(MCF_ERR_IO, "%s:%d: second problem line", path, line_no); }
            int rc = alloc_problem(out, n, m);
            if (rc) return rc;
            have_p = true;
        } else if (*q == 'n' || *q == 'a') {
            if (!have_p) { return mcf::fail(MCF_ERR_IO, "%s:%d: data before the problem line", path, line_no); }
            char *r = q + 1;
            long long v[5];
            const int want = *q == 'n' ? 2 : 5;
            int got = 0;
            for (; got < want; ++got) {
                char *after;
                errno = 0;
                v[got] = strtoll(r, &after, 10);
                if (after == r) break;
                r = after;
            }
            if (got != want) { mcf_problem_free(out); return mcf::fail(MCF_ERR_IO, "%s:%d: invalid %s line", path, line_no, *q == 'n' ? "node" : "arc"); }
            if (*q == 'n') {
                if (v[0] < 1 || v[0] > out->node_count) { mcf_problem_free(out); return mcf::fail(MCF_ERR_IO, "%s:%d: node id out of range", path, line_no); }
                out->supply[v[0] - 1] = v[1];
            } else {
                if (arcs_seen >= out->arc_count || v[0] < 1 || v[0] > out->node_count || v[1] < 1 || v[1] > out->node_count) {
                    mcf_problem_free(out);
                    return mcf::fail(MCF_ERR_IO, "%s:%d: arc out of range", path, line_no);
                }
                out->source[arcs_seen] = (int32_t)(v[0] - 1); out->target[arcs_seen] = (int32_t)(v[1] - 1);
                out->lower[arcs_seen] = v[2]; out->upper[arcs_seen] = v[3]; out->cost[arcs_seen] = v[4];
                ++arcs_seen;
            }
        }
        p = eol + 1;
    }
    if (!have_p) return mcf::fail(MCF_ERR_IO, "%s: no problem line", path);
    if (arcs_seen != out->arc_count) { const long long m = out->arc_count; mcf_problem_free(out); return mcf::fail(MCF_ERR_IO, "%s: header announces %lld arcs, file has %lld", path, m, (long long)arcs_seen); }
    return MCF_OK;
}

int mcf_dimacs_write(const mcf_problem *p, const char *path)
{
    if (!p || !path) return mcf::fail(MCF_ERR_INVALID, "null argument");
    FILE *f = fopen(path, "w");
    if (!f) return mcf::fail(MCF_ERR_IO, "%s: %s", path, strerror(errno));
    fprintf(f, "c written by mcf_hip\np min %d %d\n", p->node_count, p->arc_count);
    for (int v = 0; v < p->node_count; ++v)
        if (p->supply[v] != 0) fprintf(f, "n %d %lld\n", v + 1, (long long)p->supply[v]);
    for (int e = 0; e < p->arc_count; ++e)
        fprintf(f, "a %d %d %lld %lld %lld\n", p->source[e] + 1, p->target[e] + 1, (long long)p->lower[e], (long long)p->upper[e], (long long)p->cost[e]);
    if (fclose(f) != 0) return mcf::fail(MCF_ERR_IO, "%s: write failed", path);
    return MCF_OK;
}

// .sol files (Loaders/SolutionLoader.cs): "s COST", "f ARC FLOW" (0-based arc id, what SaveToFile writes, :186-210) or
// "f SRC DST FLOW" (1-based end points, what the bundled Gurobi solutions hold, :115-139), "p NODE POTENTIAL" (:141-153).
// The reference keeps end-point flows in a dictionary; here they are mapped onto arcs: the flow of a pair of end points goes to
// its parallel arcs in order of increasing cost (then arc id), lower bounds first, each up to its capacity -- for an optimal
// solution that is the only cost-minimal split up to ties.
int mcf_solution_write(const char *path, int64_t cost, int32_t arc_count, const int64_t *flow, int32_t node_count, const int64_t *pi)
{
    if (!path || arc_count < 0 || node_count < 0 || (arc_count && !flow)) return mcf::fail(MCF_ERR_INVALID, "mcf_solution_write: bad arguments");
    FILE *f = fopen(path, "w");
    if (!f) return mcf::fail(MCF_ERR_IO, "%s: %s", path, strerror(errno));
    fprintf(f, "s %lld\n", (long long)cost);
    for (int e = 0; e < arc_count; ++e)
        if (flow[e] != 0) fprintf(f, "f %d %lld\n", e, (long long)flow[e]);
    if (pi)
        for (int v = 0; v < node_count; ++v) fprintf(f, "p %d %lld\n", v, (long long)pi[v]);
    if (fclose(f) != 0) return mcf::fail(MCF_ERR_IO, "%s: write failed", path);
    return MCF_OK;
}

int mcf_solution_read(const char *path, const mcf_problem *p, int64_t *cost, int32_t *has_cost, int64_t *flow, int64_t *pi, int32_t *has_pi)
{
    if (!path || !p || !flow) return mcf::fail(MCF_ERR_INVALID, "mcf_solution_read: null argument");
    FILE *f = fopen(path, "rb");
    if (!f) return mcf::fail(MCF_ERR_IO, "%s: %s", path, strerror(errno));
    const int n = p->node_count, m = p->arc_count;
    std::fill(flow, flow + m, (int64_t)0);
    if (pi) std::fill(pi, pi + n, (int64_t)0);
    if (has_cost) *has_cost = 0;
    if (has_pi) *has_pi = 0;
    struct Pair { int32_t u, v; int64_t flow; };
    std::vector<Pair> pairs;
    char line[512];
    int line_no = 0;
    while (fgets(line, sizeof(line), f)) {
        ++line_no;
        char tag = 0;
        long long v[3] = {0, 0, 0};
        const int got = sscanf(line, " %c %lld %lld %lld", &tag, &v[0], &v[1], &v[2]);
        if (got < 1 || tag == 'c') continue;
        if (tag == 's' && got >= 2) {
            if (cost) *cost = v[0];
            if (has_cost) *has_cost = 1;
        } else if (tag == 'f' && got == 3) {
            if (v[0] < 0 || v[0] >= m) { fclose(f); return mcf::fail(MCF_ERR_IO, "%s:%d: arc id out of range", path, line_no); }
            flow[v[0]] = v[1];
        } else if (tag == 'f' && got == 4) {
            if (v[0] < 1 || v[0] > n || v[1] < 1 || v[1] > n) { fclose(f); return mcf::fail(MCF_ERR_IO, "%s:%d: node id out of range", path, line_no); }
            pairs.push_back(Pair{(int32_t)(v[0] - 1), (int32_t)(v[1] - 1), v[2]});
        } else if (tag == 'p' && got >= 3) {
            if (v[0] < 0 || v[0] >= n) { fclose(f); return mcf::fail(MCF_ERR_IO, "%s:%d: node id out of range", path, line_no); }
            if (pi) pi[v[0]] = v[1];
            if (has_pi) *has_pi = 1;
        }
    }
    fclose(f);
    if (pairs.empty()) return MCF_OK;
    // arcs ordered by (source, target, cost, id); every pair of end points finds its run by binary search
    std::vector<int32_t> order((size_t)m);
    for (int e = 0; e < m; ++e) order[e] = e;
    auto key_less = [&](int32_t a, int32_t b) {
        if (p->source[a] != p->source[b]) return p->source[a] < p->source[b];
        if (p->target[a] != p->target[b]) return p->target[a] < p->target[b];
        if (p->cost[a] != p->cost[b]) return p->cost[a] < p->cost[b];
        return a < b;
    };
    std::sort(order.begin(), order.end(), key_less);
    for (const Pair &q : pairs) {
        auto lo = std::partition_point(order.begin(), order.end(), [&](int32_t e) {
            return p->source[e] < q.u || (p->source[e] == q.u && p->target[e] < q.v);
        });
        auto hi = lo;
        while (hi != order.end() && p->source[*hi] == q.u && p->target[*hi] == q.v) ++hi;
        if (lo == hi) return mcf::fail(MCF_ERR_IO, "%s: flow on %d -> %d, which is not an arc of the problem", path, q.u + 1, q.v + 1);
        int64_t left = q.flow;
        for (auto it = lo; it != hi; ++it) { flow[*it] = p->lower[*it]; left -= p->lower[*it]; }
        for (auto it = lo; it != hi && left > 0; ++it) {
            const int64_t room = p->upper[*it] - p->lower[*it];
            const int64_t take = (it + 1 == hi) ? left : std::min(left, room);     // the last arc takes what is left, valid or not
            flow[*it] += take;
            left -= take;
        }
        if (left < 0) flow[*lo] += left;      // less than the lower bounds: visible to the validator as a bound violation
    }
    return MCF_OK;
}

}  // extern "C"
