// tools/treewalk.hip -- measurement only (not part of the library): what would the root-ward walks of a pivot cost on the device?
//
// SURVEY.md section 8f-2 asks about a device-resident cycle search.  FindJoinNode (NS.cs:925-941) and FindLeavingArc (NS.cs:943-1010) walk
// from the entering arc's end points towards the root: one dependent load chain per path (Parent, SuccNum, then Pred / PredDir / flow /
// upper of every node on the way).  This program takes a real spanning tree (a snapshot of a config-3 solve, written by
// tools/gpu_treewalk.py) and the entering arcs that followed it, performs both walks for every one of them
//   (a) on the host, exactly as mincostflow_amd/csrc/ns_host.cpp does,
//   (b) on ONE lane of one wavefront (what a single-wave "pivot kernel" would do), and
//   (c) on two lanes, one per path of FindLeavingArc,
// checks that all three find the same join node, leaving node and delta, and prints time per pivot and per hop.
// The tree is not updated between pivots (UpdateTreeStructure is the part that cannot move), so the device runs with warm caches:
// the figures are a LOWER bound for the device.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int64_t kMax = INT64_MAX, kInf = INT64_MAX / 2;

struct Tree {
    const int32_t *par, *pred, *sub, *src, *tgt;
    const int8_t *dir, *state;
    const int64_t *flow, *upper;
};
struct Result { int32_t join, u_out, side; int64_t delta; int32_t hops; };

template <typename T> __host__ __device__ inline Result one_pivot(const T &t, int in_arc)
{
    Result r{};
    int a = t.src[in_arc], b = t.tgt[in_arc], hops = 0;
    while (a != b) {                                        // NS.cs:925-941
        if (t.sub[a] < t.sub[b]) a = t.par[a]; else b = t.par[b];
        ++hops;
    }
    const int join = a;
    int first, second;
    if (t.state[in_arc] == 1) { first = t.src[in_arc]; second = t.tgt[in_arc]; } else { first = t.tgt[in_arc]; second = t.src[in_arc]; }
    int64_t delta = t.upper[in_arc];
    int u_out = -1, side = 0;
    for (int u = first; u != join; u = t.par[u]) {          // NS.cs:957-975
        const int e = t.pred[u];
        int64_t room = t.flow[e];
        if (t.dir[u] == -1) room = t.upper[e] >= kMax ? kInf : t.upper[e] - room;
        if (room < delta) { delta = room; u_out = u; side = 1; }
        ++hops;
    }
    for (int u = second; u != join; u = t.par[u]) {         // NS.cs:977-995
        const int e = t.pred[u];
        int64_t room = t.flow[e];
        if (t.dir[u] == 1) room = t.upper[e] >= kMax ? kInf : t.upper[e] - room;
        if (room <= delta) { delta = room; u_out = u; side = 2; }
        ++hops;
    }
    r.join = join; r.u_out = u_out; r.side = side; r.delta = delta; r.hops = hops;
    return r;
}

__global__ void walk_one_lane(Tree t, const int32_t *arcs, int k, Result *out, uint64_t *ticks)
{
    if (threadIdx.x != 0) return;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < k; ++i) out[i] = one_pivot(t, arcs[i]);
    *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

// lane 0 walks the first path of FindLeavingArc, lane 1 the second; the join node is found by lane 0 and broadcast
__global__ void walk_two_lanes(Tree t, const int32_t *arcs, int k, Result *out, uint64_t *ticks)
{
    const int lane = threadIdx.x;
    if (lane > 1) return;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < k; ++i) {
        const int in_arc = arcs[i];
        int a = t.src[in_arc], b = t.tgt[in_arc], hops = 0;
        while (a != b) { if (t.sub[a] < t.sub[b]) a = t.par[a]; else b = t.par[b]; ++hops; }
        const int join = a;
        const bool lower = t.state[in_arc] == 1;
        const int first = lower ? t.src[in_arc] : t.tgt[in_arc], second = lower ? t.tgt[in_arc] : t.src[in_arc];
        int64_t delta = t.upper[in_arc];
        int u_out = -1;
        // each lane its own path with '<' ; the reference's '<=' on the second path is restored when the two are combined
        const int start = lane == 0 ? first : second;
        const int8_t want = lane == 0 ? -1 : 1;
        int mine_hops = 0;
        for (int u = start; u != join; u = t.par[u]) {
            const int e = t.pred[u];
            int64_t room = t.flow[e];
            if (t.dir[u] == want) room = t.upper[e] >= kMax ? kInf : t.upper[e] - room;
            if (lane == 0 ? room < delta : room <= delta) { delta = room; u_out = u; }
            ++mine_hops;
        }
        const int64_t d1 = __shfl(delta, 1);
        const int u1 = __shfl(u_out, 1), h1 = __shfl(mine_hops, 1);
        if (lane == 0) {
            Result r;
            r.join = join; r.hops = hops + mine_hops + h1;
            // the second path wins ties (NS.cs:977-995 uses <= against the first path's minimum)
            if (u1 >= 0 && d1 <= delta) { r.delta = d1; r.u_out = u1; r.side = 2; }
            else { r.delta = delta; r.u_out = u_out; r.side = u_out >= 0 ? 1 : 0; }
            out[i] = r;
        }
    }
    if (lane == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

template <typename T> std::vector<T> rd(FILE *f, size_t n) { std::vector<T> v(n); if (fread(v.data(), sizeof(T), n, f) != n) { printf("short read\n"); exit(1); } return v; }
template <typename T> T *up(const std::vector<T> &v) { T *d; CK(hipMalloc((void **)&d, sizeof(T) * v.size())); CK(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice)); return d; }

int main(int argc, char **argv)
{
    if (argc < 2) { printf("usage: treewalk dump.bin\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
    auto hdr = rd<int32_t>(f, 3);
    const int n1 = hdr[0], A = hdr[1], K = hdr[2];
    auto par = rd<int32_t>(f, n1), pred = rd<int32_t>(f, n1), sub = rd<int32_t>(f, n1);
    auto dir = rd<int8_t>(f, n1);
    auto src = rd<int32_t>(f, A), tgt = rd<int32_t>(f, A);
    auto flow = rd<int64_t>(f, A), upper = rd<int64_t>(f, A);
    auto state = rd<int8_t>(f, A);
    auto arcs = rd<int32_t>(f, K);
    fclose(f);
    Tree h{par.data(), pred.data(), sub.data(), src.data(), tgt.data(), dir.data(), state.data(), flow.data(), upper.data()};
    std::vector<Result> hr(K);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; ++i) hr[i] = one_pivot(h, arcs[i]);
        best = std::min(best, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count());
    }
    long long hops = 0;
    for (auto &r : hr) hops += r.hops;
    printf("tree of %d nodes, %d pivots, %.1f hops per pivot (join search + both paths)\n", n1, K, (double)hops / K);
    printf("host (one core)        : %8.1f ns per pivot, %6.2f ns per hop\n", best / K, best / hops);

    Tree d{up(par), up(pred), up(sub), up(src), up(tgt), up(dir), up(state), up(flow), up(upper)};
    int32_t *d_arcs = up(arcs);
    Result *d_out; uint64_t *d_ticks;
    CK(hipMalloc((void **)&d_out, sizeof(Result) * K));
    CK(hipMalloc((void **)&d_ticks, 8));
    for (int variant = 0; variant < 2; ++variant) {
        uint64_t ticks = 0, bestt = ~0ull;
        for (int rep = 0; rep < 3; ++rep) {
            if (variant == 0) hipLaunchKernelGGL(walk_one_lane, dim3(1), dim3(64), 0, 0, d, d_arcs, K, d_out, d_ticks);
            else hipLaunchKernelGGL(walk_two_lanes, dim3(1), dim3(64), 0, 0, d, d_arcs, K, d_out, d_ticks);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
            bestt = std::min(bestt, ticks);
        }
        std::vector<Result> dr(K);
        CK(hipMemcpy(dr.data(), d_out, sizeof(Result) * K, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < K; ++i) bad += dr[i].join != hr[i].join || dr[i].u_out != hr[i].u_out || dr[i].delta != hr[i].delta || dr[i].side != hr[i].side;
        const double ns = 10.0 * (double)bestt;           // s_memrealtime ticks at 100 MHz
        printf("device, %s: %8.1f ns per pivot, %6.2f ns per hop   (%d of %d results differ from the host's)\n",
               variant == 0 ? "one lane       " : "two lanes      ", ns / K, ns / hops, bad, K);
    }
    return 0;
}
