// resident_host.hip.h -- host side of the resident grids (included by engine.hip only, inside its anonymous namespace): the mailbox in
// BAR-mapped fine-grained VRAM, launching / posting / streaming / stopping a grid, the per-device resident slots, and collecting the
// per-workgroup records of a search.  DESIGN.md sections 3.1, 3.3, 3.8.
#pragma once

// ---- fine-grained VRAM that the CPU can write through the PCIe BAR (what HIP itself uses for device-side kernel arguments)
struct HsaPick {
    int want_bdf = -1, want_domain = -1, ordinal = 0, seen = 0;
    hsa_agent_t cpu{}, gpu{};
    bool have_cpu = false, have_gpu = false, have_pool = false;
    hsa_amd_memory_pool_t pool{};
};
hsa_status_t hsa_pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    HsaPick *k = (HsaPick *)data;
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !k->have_pool) { k->pool = pool; k->have_pool = true; }
    return HSA_STATUS_SUCCESS;
}
hsa_status_t hsa_agent_cb(hsa_agent_t a, void *data)
{
    HsaPick *k = (HsaPick *)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !k->have_cpu) { k->cpu = a; k->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU) {
        uint32_t bdf = 0, domain = 0;
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        const bool by_bdf = k->want_bdf >= 0 && (int)(bdf >> 8) == k->want_bdf && (k->want_domain < 0 || (int)domain == k->want_domain);
        const bool by_ord = k->want_bdf < 0 && k->seen == k->ordinal;
        if (!k->have_gpu && (by_bdf || by_ord)) { k->gpu = a; k->have_gpu = true; hsa_amd_agent_iterate_memory_pools(a, hsa_pool_cb, k); }
        k->seen++;
    }
    return HSA_STATUS_SUCCESS;
}
// returns nullptr when the platform offers no host-writable fine-grained VRAM (resident mode is then simply not used)
uint32_t *alloc_bar_vram(int hip_device, size_t bytes)
{
    if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;
    HsaPick k;
    int bus = -1, dom = -1;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_device) == hipSuccess) k.want_bdf = bus;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_device) == hipSuccess) k.want_domain = dom;
    k.ordinal = hip_device;
    hsa_iterate_agents(hsa_agent_cb, &k);
    if (!k.have_gpu) { k = HsaPick{}; k.ordinal = hip_device; hsa_iterate_agents(hsa_agent_cb, &k); }
    if (!k.have_gpu || !k.have_cpu || !k.have_pool) return nullptr;
    void *ptr = nullptr;
    if (hsa_amd_memory_pool_allocate(k.pool, bytes, 0, &ptr) != HSA_STATUS_SUCCESS) return nullptr;
    hsa_agent_t both[2] = {k.cpu, k.gpu};
    if (hsa_amd_agents_allow_access(2, both, nullptr, ptr) != HSA_STATUS_SUCCESS) { hsa_amd_memory_pool_free(ptr); return nullptr; }
    return (uint32_t *)ptr;
}

constexpr int kBucketNodes = 131072;        // 1 MB of int64 potentials per range: measured best on config 5 (profiles/r01_bucketed_layout_feasibility.txt)
constexpr int kBucketMinArcs = 2 << 20;
constexpr int kResidentMaxGrid = 256;       // one workgroup (64..1024 threads) per CU: always co-resident, every CU gathers
constexpr uint32_t kResidentIdleTicks = 25000000u;   // 0.25 s of s_memrealtime
uint32_t resident_idle_ticks()                       // MCF_HIP_IDLE_MS: tests shorten it so that grids leave between two searches
{
    uint32_t x = kResidentIdleTicks;
    if (const char *u = getenv("MCF_HIP_IDLE_MS")) { const long long ms = atoll(u); if (ms >= 1 && ms <= 10000) x = (uint32_t)(ms * 100000); }
    return x;
}

template <typename T, int RULE, bool OPT>
void launch_resident_r(mcf_engine *e, const ResidentParams<T> &p)
{
    const dim3 grid(e->res_grid), block(e->res_threads);
    const bool lpi = e->lds_pi;
    if (e->cand_on) {     // candidates: Best Eligible, register-resident tiles only
        if (e->shift_grid) {
            if constexpr (sizeof(T) == 8) {
                if (e->cand_tiles == 2) hipExtLaunchKernelGGL((resident_cand_kernel<T, 2>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p, e->shift_base, e->max_shift_lines);
                else hipExtLaunchKernelGGL((resident_cand_kernel<T, 4>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p, e->shift_base, e->max_shift_lines);
            }
        }
        else if (lpi) hipExtLaunchKernelGGL((resident_kernel<T, MCF_RULE_BEST_ELIGIBLE, false, true, true, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
        else if (e->res_threads <= kPiRegThreads && !e->no_pireg)       // the end points' potentials stay in registers between the requests
            hipExtLaunchKernelGGL((resident_kernel<T, MCF_RULE_BEST_ELIGIBLE, false, true, false, true, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
        else hipExtLaunchKernelGGL((resident_kernel<T, MCF_RULE_BEST_ELIGIBLE, false, true, false, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    }
    else if (e->resident_reg && lpi) hipExtLaunchKernelGGL((resident_kernel<T, RULE, OPT, true, true, false>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    else if (e->resident_reg && e->res_threads <= kPiRegThreads && !e->no_pireg)
        hipExtLaunchKernelGGL((resident_kernel<T, RULE, OPT, true, false, false, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    else if (e->resident_reg) hipExtLaunchKernelGGL((resident_kernel<T, RULE, OPT, true, false, false>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    else if (lpi) hipExtLaunchKernelGGL((resident_kernel<T, RULE, OPT, false, true, false>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    else {
        if constexpr (RULE == MCF_RULE_BEST_ELIGIBLE) {
            if (e->bucket_nodes > 0) {
                hipExtLaunchKernelGGL((resident_kernel<T, RULE, OPT, false, false, false, false, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
                return;
            }
        }
        hipExtLaunchKernelGGL((resident_kernel<T, RULE, OPT, false, false, false>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    }
}

template <int RULE, bool OPT>
void launch_resident_rc_r(mcf_engine *e, const ResidentRcParams &p)
{
    const dim3 grid(e->res_grid), block(e->res_threads);
    if constexpr (RULE == MCF_RULE_BEST_ELIGIBLE) {
        if (e->cand_on) {          // the candidate cache's grid: every search publishes each group's smallest few, not only the best
            if (e->rc_lds) hipExtLaunchKernelGGL((resident_rc_kernel<RULE, OPT, true, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
            else hipExtLaunchKernelGGL((resident_rc_kernel<RULE, OPT, false, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
            return;
        }
    }
    if (e->rc_lds) hipExtLaunchKernelGGL((resident_rc_kernel<RULE, OPT, true>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
    else hipExtLaunchKernelGGL((resident_rc_kernel<RULE, OPT, false>), grid, block, 0, e->res_stream, e->res_start, e->res_stop, 0, p);
}

int launch_resident_rc(mcf_engine *e, uint32_t start_seq)
{
    ResidentRcParams p;
    p.state = e->d_state; p.rc = e->d_rc; p.pi = e->d_pi; p.adj_start = e->d_adj_start; p.adj = e->d_adj; p.slots = e->d_slots;
    p.mailbox = e->mailbox; p.exit_word = e->d_exit;
    p.base = e->begin; p.count_padded = e->count_padded; p.m_s = e->d.search_arc_num; p.window = e->rc_window;
    p.start_seq = start_seq; p.idle_ticks = resident_idle_ticks(); p.narrow = e->d.int_width == 32 ? 1 : 0;
    p.max_pi = e->patch_capacity; p.max_st = e->mailbox_max_st; p.poll_replicas = e->poll_replicas; p.poll_sleep = e->poll_sleep;
    p.src = e->d_src; p.tgt = e->d_tgt; p.cost = e->d_cost; p.n_nodes = e->d.node_count;
    p.host_pi = e->d_ext_pi; p.barrier = e->d_barrier;
    if (e->d_barrier) HIP_TRY(hipMemsetAsync(e->d_barrier, 0, 64, e->res_stream));
    switch (e->d.rule) {
    case MCF_RULE_BEST_ELIGIBLE: launch_resident_rc_r<MCF_RULE_BEST_ELIGIBLE, false>(e, p); break;
    case MCF_RULE_FIRST_ELIGIBLE: launch_resident_rc_r<MCF_RULE_FIRST_ELIGIBLE, false>(e, p); break;
    default:
        if (e->d.semantics == MCF_SEM_OPTIMIZED) launch_resident_rc_r<MCF_RULE_BLOCK_SEARCH, true>(e, p);
        else launch_resident_rc_r<MCF_RULE_BLOCK_SEARCH, false>(e, p);
    }
    HIP_TRY(hipGetLastError());
    return MCF_OK;
}

template <typename T>
int launch_resident(mcf_engine *e, uint32_t start_seq)
{
    ResidentParams<T> p;
    p.src = e->d_src; p.tgt = e->d_tgt; p.cost = (const T *)e->d_cost; p.state = e->d_state; p.pi = (T *)e->d_pi;
    p.slots = e->d_slots; p.orig = e->bucket_nodes > 0 ? e->d_orig : nullptr; p.mailbox = e->mailbox; p.exit_word = e->d_exit;
    p.base = e->begin; p.count_padded = e->count_padded; p.m_s = e->d.search_arc_num;
    p.start_seq = start_seq; p.idle_ticks = resident_idle_ticks(); p.n_nodes = e->d.node_count; p.max_pi = e->patch_capacity; p.max_st = e->mailbox_max_st; p.poll_replicas = e->poll_replicas; p.poll_sleep = e->poll_sleep;
    p.host_pi = e->shift_grid && e->d_barrier ? e->d_ext_pi : nullptr;
    p.barrier = e->d_barrier;
    if (p.host_pi) HIP_TRY(hipMemsetAsync(e->d_barrier, 0, 64, e->res_stream));
    const bool opt = e->d.semantics == MCF_SEM_OPTIMIZED;
    switch (e->d.rule) {
    case MCF_RULE_BEST_ELIGIBLE: launch_resident_r<T, MCF_RULE_BEST_ELIGIBLE, false>(e, p); break;
    case MCF_RULE_FIRST_ELIGIBLE: launch_resident_r<T, MCF_RULE_FIRST_ELIGIBLE, false>(e, p); break;
    default:
        if (opt) launch_resident_r<T, MCF_RULE_BLOCK_SEARCH, true>(e, p);
        else launch_resident_r<T, MCF_RULE_BLOCK_SEARCH, false>(e, p);
    }
    HIP_TRY(hipGetLastError());
    return MCF_OK;
}

// HIP multiplexes a process's streams onto a few hardware queues per device (GPU_MAX_HW_QUEUES, 4 by default) and a resident grid never
// leaves its queue: a fifth grid on the same device could be queued behind one that only ends when its solve does.  So at most
// kResidentSlots grids of one process run on a device at a time; an engine that finds no slot serves that search with one dispatch.
constexpr int kMaxDevices = 64;
std::atomic<int> g_resident_running[kMaxDevices];
int resident_slot_limit()
{
    static const int limit = [] {
        int v = 4;
        if (const char *q = getenv("GPU_MAX_HW_QUEUES")) { const int x = atoi(q); if (x >= 1 && x <= 64) v = x; }
        return v;
    }();
    return limit;
}
bool resident_slot_acquire(mcf_engine *e)
{
    if (e->has_slot) return true;
    std::atomic<int> &c = g_resident_running[e->d.device % kMaxDevices];
    if (c.fetch_add(1, std::memory_order_acq_rel) >= resident_slot_limit()) { c.fetch_sub(1, std::memory_order_acq_rel); return false; }
    e->has_slot = true;
    return true;
}
void resident_slot_release(mcf_engine *e)
{
    if (!e->has_slot) return;
    g_resident_running[e->d.device % kMaxDevices].fetch_sub(1, std::memory_order_acq_rel);
    e->has_slot = false;
}

int resident_start(mcf_engine *e, uint32_t start_seq)
{
    if (e->resident_running) return MCF_OK;
    if (!e->has_slot) {       // given back by a stop in between (a list too long for the mailbox): the other grids leave when their solves park them
        const double t0 = mcf::now_ns();
        while (!resident_slot_acquire(e)) {
            _mm_pause();
            if (mcf::now_ns() - t0 > 20e9) return mcf::fail(MCF_ERR_TIMEOUT, "no resident slot on device %d became free within 20 s", e->d.device);
        }
    }
    for (int i = 0; i < 16; ++i) ((volatile uint32_t *)e->h_exit)[i] = 0;
    // whatever the engine's own stream still carries (uploads, a flush) is what the grid reads when it starts
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!e->res_stream) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&e->res_stream, hipStreamNonBlocking, greatest));
    }
    int rc = e->rc_mode ? launch_resident_rc(e, start_seq) : (e->d.int_width == 32 ? launch_resident<int32_t>(e, start_seq) : launch_resident<int64_t>(e, start_seq));
    if (rc) return rc;
    e->resident_running = true;
    e->st.resident_launches += 1;
    return MCF_OK;
}

// one 64-byte line into the write-combining BAR mapping: four 16-byte stores, the tag goes out with the last one
inline void mailbox_write_line(uint32_t *dst, const uint32_t *line16)
{
    const __m128i *src = (const __m128i *)line16;
    __m128i *d = (__m128i *)dst;
    _mm_store_si128(d + 0, _mm_loadu_si128(src + 0));
    _mm_store_si128(d + 1, _mm_loadu_si128(src + 1));
    _mm_store_si128(d + 2, _mm_loadu_si128(src + 2));
    _mm_store_si128(d + 3, _mm_loadu_si128(src + 3));
}

void resident_post(mcf_engine *e, uint32_t seq, uint32_t cmd, bool with_patches)
{
    alignas(16) uint32_t line[16];
    const int n_pi = with_patches ? (int)e->pend_node.size() : 0, n_st = with_patches ? (int)e->pend_arc.size() : 0;
    // entries beyond the header: potential patches 1.., then state patches 2..
    const int extra_pi = n_pi > 1 ? n_pi - 1 : 0, extra_st = n_st > 2 ? n_st - 2 : 0, entries = extra_pi + extra_st;
    alignas(16) uint32_t line1[16];
    memset(line1, 0, sizeof(line1));
    for (int l = 0, i = 0; i < entries; ++l) {
        if (l > 0 && l + 1 <= e->stream_lines) { i += kMailboxPatchesPerLine; continue; }   // already in place (resident_stream); line 1 always goes out again
        memset(line, 0, sizeof(line));
        for (int k = 0; k < kMailboxPatchesPerLine && i < entries; ++k, ++i) {
            if (i < extra_pi) {
                const uint64_t v = (uint64_t)e->pend_val[i + 1];
                line[3 * k] = (uint32_t)e->pend_node[i + 1];
                line[3 * k + 1] = (uint32_t)v;
                line[3 * k + 2] = (uint32_t)(v >> 32);
            } else {
                const int j = i - extra_pi + 2;
                line[3 * k] = (uint32_t)e->pend_arc[j];
                line[3 * k + 1] = (uint32_t)e->pend_state[j];
            }
        }
        line[15] = seq;
        if (l == 0) memcpy(line1, line, sizeof(line));                                  // line 1 goes out with every copy of the poll unit
        else mailbox_write_line(e->mailbox + kMailboxTail + 16 * (size_t)(l - 1), line);  // lines 2.. : the tail
    }
    if (with_patches) e->stream_lines = 0;
    memset(line, 0, sizeof(line));
    line[0] = seq;
    line[1] = cmd;
    const int na = e->next_arc >= e->d.search_arc_num ? 0 : e->next_arc;
    line[2] = (uint32_t)na;
    int rstar = -1;
    if (e->d.rule == MCF_RULE_BLOCK_SEARCH && e->d.semantics == MCF_SEM_OPTIMIZED && e->next_arc < e->d.search_arc_num) {
        const int len1 = e->d.search_arc_num - e->next_arc;
        if (len1 % e->block_size != 0) rstar = len1 / e->block_size;
    }
    line[3] = (uint32_t)rstar;
    line[13] = (uint32_t)e->block_size;      // per request: the adaptive rule of the plain Block Search changes it between searches
    line[4] = (uint32_t)n_pi;
    line[5] = (uint32_t)n_st;
    for (int k = 0; k < n_st && k < 2; ++k) { line[6 + 2 * k] = (uint32_t)e->pend_arc[k]; line[7 + 2 * k] = (uint32_t)e->pend_state[k]; }
    if (n_pi > 0) {
        const uint64_t v = (uint64_t)e->pend_val[0];
        line[10] = (uint32_t)e->pend_node[0];
        line[11] = (uint32_t)v;
        line[12] = (uint32_t)(v >> 32);
    }
    line[15] = seq;
    if (entries > kMailboxPatchesPerLine) _mm_sfence();   // tail lines leave the write-combining buffers before any header does
    for (int r = 0; r < e->poll_replicas; ++r) {
        uint32_t *unit = e->mailbox + (size_t)r * kReplicaStride;
        if (entries > 0) mailbox_write_line(unit + 16, line1);
        mailbox_write_line(unit, line);
    }
    _mm_sfence();
}

// Long potential lists start travelling while the host is still producing them (mcf_engine_append_potential): the complete entry lines
// gathered so far go into the mailbox and an "apply" post (cmd 2) tells the grid how far the list of the COMING scan request reaches.
// No answer is expected; the posts are cumulative and the scan request finishes the list (kernels.hip.h, mailbox layout).
int stream_min_lines()                         // 1920 entries per post at least by default: one piece of the host driver's walk (2048 nodes)
{
    static const int v = [] { int x = 384; if (const char *u = getenv("MCF_HIP_STREAM_LINES")) { const int y = atoi(u); if (y >= 16 && y <= 65536) x = y; } return x; }();
    return v;
}

void resident_stream(mcf_engine *e)
{
    // candidate mode: only one pivot's own big list travels ahead (its entries are final and repeat no node), and never beside a list refresh
    // that is still on its way (the mailbox holds one request)
    if (e->cand_on ? (e->async_posted || e->blind_epoch != e->cand_now || e->blind_sets > 1) : e->pend_arc.size() > 2) return;
    if (!e->resident_running || e->rc_mode || e->in_flight != mcf_engine::kNoSearch) return;
    const int n_pi = (int)e->pend_node.size();
    const int complete = (n_pi > 1 ? n_pi - 1 : 0) / kMailboxPatchesPerLine;
    if (complete - e->stream_lines < stream_min_lines()) return;
    uint32_t next_seq = e->seq + 1;
    if (next_seq == 0) next_seq = 1;
    alignas(16) uint32_t line[16], line1[16];
    memset(line1, 0, sizeof(line1));
    for (int l = e->stream_lines == 0 ? 0 : e->stream_lines; l < complete; ++l) {          // l = entry line l + 1
        memset(line, 0, sizeof(line));
        for (int k = 0; k < kMailboxPatchesPerLine; ++k) {
            const int i = l * kMailboxPatchesPerLine + k;
            const uint64_t v = (uint64_t)e->pend_val[i + 1];
            line[3 * k] = (uint32_t)e->pend_node[i + 1];
            line[3 * k + 1] = (uint32_t)v;
            line[3 * k + 2] = (uint32_t)(v >> 32);
        }
        line[15] = next_seq;
        if (l == 0) memcpy(line1, line, sizeof(line));
        else mailbox_write_line(e->mailbox + kMailboxTail + 16 * (size_t)(l - 1), line);
    }
    const bool first_post = e->stream_lines == 0;
    e->stream_sub += 1;
    if (e->stream_sub == 0) e->stream_sub = 1;
    memset(line, 0, sizeof(line));
    line[0] = next_seq;
    line[1] = 2u;
    line[13] = (uint32_t)complete;
    line[14] = e->stream_sub;
    line[15] = next_seq;
    _mm_sfence();                                  // the entry lines leave the write-combining buffers before any header does
    for (int r = 0; r < e->poll_replicas; ++r) {
        uint32_t *unit = e->mailbox + (size_t)r * kReplicaStride;
        if (first_post) mailbox_write_line(unit + 16, line1);
        mailbox_write_line(unit, line);
    }
    _mm_sfence();
    e->stream_lines = complete;
}

int search_end(mcf_engine *e, Key *k);
int device_sync_from_mirrors(mcf_engine *e);
void shift_post(mcf_engine *e, uint32_t seq, uint32_t cmd, bool with_patches);
void shift_stream(mcf_engine *e);
int cand_collect(mcf_engine *e, uint32_t at);
bool cand_records_ready(const mcf_engine *e, int g);
int resident_stop(mcf_engine *e);
void resident_stream(mcf_engine *e);

// the grid has been told to leave (or left by itself): wait for it, give its stream (and with it the hardware queue) back
int resident_join(mcf_engine *e, bool harvest = true);
void resident_harvest(mcf_engine *e);
int resident_join(mcf_engine *e, bool harvest)
{
    if (!e->res_stream) return MCF_OK;
    const hipError_t r = hipStreamSynchronize(e->res_stream);
    if (r == hipSuccess && harvest) resident_harvest(e);          // (the launch's events, before their stream goes)
    (void)hipStreamDestroy(e->res_stream);
    e->res_stream = nullptr;
    if (r != hipSuccess) return mcf::fail(MCF_ERR_HIP, "the resident grid ended with: %s", hipGetErrorString(r));
    return MCF_OK;
}

// statistics of a resident launch that has ended (the grid wrote them into the exit record before it left)
void resident_harvest(mcf_engine *e)
{
    const volatile uint32_t *x = e->h_exit;
    e->st.resident_requests += x[1];
    e->st.resident_scan_ns += 10.0 * (double)(((uint64_t)x[3] << 32) | x[2]);
    if (e->shift_grid) {
        auto u64 = [&](int i) { return (double)(((uint64_t)x[i + 1] << 32) | x[i]); };
        e->st.phase_shift_ns += 10.0 * u64(4); e->st.phase_values_ns += 10.0 * u64(6); e->st.phase_scan_ns += 10.0 * u64(8);
        if (getenv("MCF_HIP_CAND_DEBUG")) fprintf(stderr, "[grid] shader clock over the launch: %.0f MHz\n", u64(10));
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->res_start, e->res_stop) == hipSuccess) e->st.resident_kernel_ns += (double)ms * 1e6;
}

int resident_stop(mcf_engine *e)
{
    // (a slot taken by a search that was then answered on the host, without a grid ever starting, goes back too: a parked or destroyed
    // engine must not keep one of the device's few slots)
    if (!e->resident_running) { resident_slot_release(e); return MCF_OK; }
    if (e->async_posted) {       // a list refresh is on its way: take it in before the grid is told to leave
        const int rc = cand_collect(e, e->async_at);
        if (rc) return rc;
    }
    if (e->in_flight == mcf_engine::kResidentSearch || e->in_flight == mcf_engine::kCandSearch) {
        // a posted search is answered before the grid is told to leave; mcf_engine_search_end then finds the answer waiting
        Key k;
        const int rc = search_end(e, &k);
        if (rc) return rc;
        e->answered = k;
        e->in_flight = mcf_engine::kAnswered;
    }
    e->prev_seq = e->seq;
    e->seq += 1;
    if (e->seq == 0) e->seq = 1;
    if (e->shift_grid) shift_post(e, e->seq, 1u, false);
    else resident_post(e, e->seq, 1u, false);
    { const int rcj = resident_join(e); if (rcj) return rcj; }       // bounded: the grid leaves on quit, or by itself after kResidentIdleTicks; counts what the launch served
    e->resident_running = false;
    e->stream_lines = 0;
    resident_slot_release(e);
    if (e->shift_grid) return device_sync_from_mirrors(e);
    return MCF_OK;
}

// the resident grid left on its idle timeout while a request was on its way: count what that launch served, start the grid again; it
// finds the request in the mailbox (start_seq = the previous request)
int resident_restart(mcf_engine *e)
{
    { const int rcj = resident_join(e); if (rcj) return rcj; }
    e->resident_running = false;
    if (e->shift_grid) {
        // that grid's registers are gone and the arrays in memory are not what it had: write them again from the host's mirrors (they are
        // at least as new as the request in flight, which is all a candidate list needs -- see cand_decide), and put the request there
        // again WITHOUT patches, or the new grid would apply a shift the values already contain
        e->stream_lines = 0;
        const int rc = device_sync_from_mirrors(e);
        if (rc) return rc;
        shift_post(e, e->seq, 0u, false);
    }
    return resident_start(e, e->prev_seq);
}

// wait for the `grid` records of dispatch `seq`, merge them with the rule's ordering
int collect(mcf_engine *e, int grid, Key *out)
{
    constexpr int stride = kSlotStride;
    const double t0 = (double)__rdtsc();
    double t0_wall = 0;
    const bool block_rule = e->d.rule == MCF_RULE_BLOCK_SEARCH, best_rule = e->d.rule == MCF_RULE_BEST_ELIGIBLE;
    Key best{0, kNone, kNone};
    const volatile Slot *slots = e->h_slots;
    const uint32_t seq = e->seq;
    const int rstar = [&] {
        if (!(block_rule && e->d.semantics == MCF_SEM_OPTIMIZED) || e->next_arc >= e->d.search_arc_num) return -1;
        const int len1 = e->d.search_arc_num - e->next_arc;
        return len1 % e->block_size ? len1 / e->block_size : -1;
    }();
    const int na = e->next_arc >= e->d.search_arc_num ? 0 : e->next_arc;
    // OPTIMIZED Block Search: records 0, 1 of a line = the block key, records 2, 3 = the range key (kernels.hip.h: kDual)
    const bool dual = block_rule && e->d.semantics == MCF_SEM_OPTIMIZED;
    Key range{0, kNone, kNone};
    for (int g = 0; g < grid; ++g) {
        uint64_t spins = 0;
        auto ready = [&](const volatile Slot &r) { const int64_t c = r.c; const uint32_t q = r.p; return r.tag == record_tag(seq, c, q); };
        auto pair_ready = [&](const volatile Slot *r) { return ready(r[0]) && ready(r[1]) && r[0].c == r[1].c && r[0].p == r[1].p; };
        while (!(pair_ready(slots + (size_t)g * stride) && (!dual || pair_ready(slots + (size_t)g * stride + 2)))) {
            _mm_pause();
            if (e->resident_running && (spins & 0xFFF) == 0xFFF && ((const volatile uint32_t *)e->h_exit)[0] != 0) {
                if (((const volatile uint32_t *)e->h_exit)[0] == 4u) {
                    (void)resident_join(e, false);
                    e->resident_running = false;
                    resident_slot_release(e);
                    return mcf::fail(MCF_ERR_TIMEOUT, "the resident grid could not meet at its grid-wide barrier while a list was being applied (are its workgroups all resident?): the device arrays are undefined");
                }
                int rc = resident_restart(e);
                if (rc) return rc;
            }
            if ((++spins & 0xFFFFF) == 0) {
                const hipError_t q = hipStreamQuery(e->resident_running && e->res_stream ? e->res_stream : e->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return mcf::fail(MCF_ERR_HIP, "scan dispatch failed: %s", hipGetErrorString(q));
                if (t0_wall == 0) t0_wall = mcf::now_ns();
                else if (mcf::now_ns() - t0_wall > 20e9) return mcf::fail(MCF_ERR_TIMEOUT, "no answer from the device after 20 s (workgroup %d of %d)", g, grid);
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        Key k;
        k.c = slots[(size_t)g * stride].c;
        k.p = slots[(size_t)g * stride].p;
        k.r = 0;
        if (k.p == kNone) continue;
        bool take;
        if (best_rule) take = best.p == kNone || k.c < best.c || (k.c == best.c && k.p < best.p);
        else if (!block_rule) take = k.p < best.p;
        else {
            uint32_t r = k.p / (uint32_t)e->block_size;
            const int arc = (int)((k.p + (uint32_t)na) % (uint32_t)e->d.search_arc_num);
            k.r = 2 * r + ((rstar >= 0 && (int)r == rstar && arc < e->next_arc) ? 1u : 0u);
            take = best.p == kNone || k.r < best.r || (k.r == best.r && (k.c < best.c || (k.c == best.c && k.p < best.p)));
        }
        if (take) best = k;
        if (dual) {
            Key q;
            q.c = slots[(size_t)g * stride + 2].c;
            q.p = slots[(size_t)g * stride + 2].p;
            if (q.p != kNone) {
                q.r = (int)((q.p + (uint32_t)na) % (uint32_t)e->d.search_arc_num) < na ? 1u : 0u;
                if (range.p == kNone || q.r < range.r || (q.r == range.r && (q.c < range.c || (q.c == range.c && q.p < range.p)))) range = q;
            }
        }
    }
    e->range_key = range;
    e->wait_ticks += (double)__rdtsc() - t0;
    *out = best;
    return MCF_OK;
}

