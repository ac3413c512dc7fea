// exchange.cpp -- the host-side MINLOC exchange of sharded engines ("each GPU writing its pair to ... host memory and the
// host reducing 8 pairs"): one process per GPU, every rank publishes its 32-byte candidate in POSIX shared memory and reads the others'.
// No collective library, no device involvement: the records are already on the host when a search ends.
//
// Life cycle of a segment: RANK 0 OWNS IT.  Rank 0 unlinks whatever an earlier (possibly crashed) run left under the name, creates the
// segment exclusively, sizes it, zeroes it and publishes a fresh GENERATION nonce in its header; the other ranks open without O_CREAT,
// retrying until the segment exists and carries a nonce that differs from the one they used last under this name in this process (so a
// re-open cannot latch onto the segment of the previous solve while rank 0 is still replacing it).  mcf_exchange_open then is a barrier:
// every rank raises its presence flag in the segment and waits until all `world` flags are up -- nobody exchanges before everybody is on
// the same segment.  A flag that is already up means somebody has been this rank on this segment before (a crashed run's leftover): the
// segment is dropped and the name opened again, and a rank waiting on a leftover notices when rank 0 replaces it (the name then leads to
// another inode).  Only rank 0 unlinks, when it closes; the others' mappings stay valid until they unmap.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include <immintrin.h>

#include "common.h"

namespace {

struct alignas(128) XHeader {
    std::atomic<uint64_t> generation;   // 0 while rank 0 is still setting the segment up
    uint32_t world;
};

// one cache line pair per rank and parity: the record and the sequence number that publishes it
struct alignas(128) XSlot {
    mcf_candidate rec;
    std::atomic<uint64_t> seq;
    std::atomic<uint32_t> present;      // parity-0 row only: the rank has mapped THIS generation (a second arrival means the segment is a leftover)
};

constexpr mcf_candidate kNoCandidate{0, 0xFFFFFFFFu, -1, 0, 0xFFFFFFFFu, -1};
constexpr double kOpenTimeoutNs = 60e9;

std::mutex g_seen_mutex;
std::map<std::string, uint64_t> g_seen;     // name -> the generation this process used last (ranks other than 0)

uint64_t fresh_generation()
{
    static std::atomic<uint64_t> counter{0};
    uint64_t g = (uint64_t)mcf::now_ns() ^ ((uint64_t)getpid() << 40) ^ (counter.fetch_add(1) << 56);
    return g ? g : 1;
}

// does `name` still lead to the segment with this inode?  (rank 0 replaces a leftover by unlink + create)
bool still_linked(const char *name, ino_t ino)
{
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return false;
    struct stat sb;
    const bool same = fstat(fd, &sb) == 0 && sb.st_ino == ino;
    close(fd);
    return same;
}

}  // namespace

struct mcf_exchange {
    std::string name;
    int rank = 0, world = 1, fd = -1;
    size_t bytes = 0;
    XHeader *header = nullptr;
    XSlot *slots = nullptr;      // [2][world], behind the header
    uint64_t seq = 0;
};

extern "C" {

int mcf_exchange_open(mcf_exchange **out, const char *name, int32_t rank, int32_t world)
{
    if (!out || !name || name[0] != '/' || world < 1 || rank < 0 || rank >= world) return mcf::fail(MCF_ERR_INVALID, "mcf_exchange_open: bad arguments (name must start with '/')");
    *out = nullptr;
    mcf_exchange *x = new mcf_exchange();
    x->name = name; x->rank = rank; x->world = world;
    x->bytes = sizeof(XHeader) + sizeof(XSlot) * 2 * (size_t)world;
    const double t0 = mcf::now_ns();
    auto drop = [&] {
        if (x->header) munmap(x->header, x->bytes);
        if (x->fd >= 0) close(x->fd);
        x->header = nullptr; x->slots = nullptr; x->fd = -1;
    };
    auto give_up = [&](int code, const char *what) {
        drop();
        delete x;
        return mcf::fail(code, "mcf_exchange_open(%s, rank %d of %d): %s", name, rank, world, what);
    };
    auto all_present = [&] {
        for (int r = 0; r < world; ++r) if (x->slots[r].present.load(std::memory_order_acquire) == 0) return false;
        return true;
    };
    if (rank == 0) {
        shm_unlink(name);                                     // a leftover of an earlier run (whoever still maps it keeps its own copy)
        x->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (x->fd < 0) return give_up(MCF_ERR_IO, "shm_open(O_CREAT | O_EXCL) failed: is another rank 0 using this name?");
        if (ftruncate(x->fd, (off_t)x->bytes) != 0) return give_up(MCF_ERR_IO, "ftruncate failed");
        void *m = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
        if (m == MAP_FAILED) return give_up(MCF_ERR_IO, "mmap failed");
        x->header = (XHeader *)m;
        x->slots = (XSlot *)((char *)m + sizeof(XHeader));
        for (size_t i = 0; i < 2 * (size_t)world; ++i) {
            x->slots[i].rec = kNoCandidate;
            x->slots[i].seq.store(0, std::memory_order_relaxed);
            x->slots[i].present.store(0, std::memory_order_relaxed);
        }
        x->header->world = (uint32_t)world;
        x->slots[0].present.store(1, std::memory_order_relaxed);
        x->header->generation.store(fresh_generation(), std::memory_order_release);        // the segment is open for the others
        // the barrier the header of this file promises: nobody leaves before all ranks have mapped this generation
        while (!all_present()) {
            _mm_pause();
            if (mcf::now_ns() - t0 > kOpenTimeoutNs) return give_up(MCF_ERR_TIMEOUT, "not all ranks arrived within 60 s");
        }
        *out = x;
        return MCF_OK;
    }
    uint64_t last = 0;
    { std::lock_guard<std::mutex> lock(g_seen_mutex); auto it = g_seen.find(name); if (it != g_seen.end()) last = it->second; }
    for (;;) {
        drop();
        if (mcf::now_ns() - t0 > kOpenTimeoutNs) return give_up(MCF_ERR_TIMEOUT, "no fresh segment with all ranks on it within 60 s (is rank 0 running?)");
        x->fd = shm_open(name, O_RDWR, 0600);
        if (x->fd < 0) { usleep(200); continue; }
        struct stat sb;
        if (fstat(x->fd, &sb) != 0 || (size_t)sb.st_size < x->bytes) { usleep(200); continue; }      // not sized yet (or another world's)
        void *m = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
        if (m == MAP_FAILED) { x->header = nullptr; return give_up(MCF_ERR_IO, "mmap failed"); }
        x->header = (XHeader *)m;
        x->slots = (XSlot *)((char *)m + sizeof(XHeader));
        // this segment's nonce; the previous solve's segment (same process, rank 0 has not replaced it yet) is recognised by its nonce,
        // a crashed run's by the presence flag below, and either is dropped and the name opened again
        uint64_t gen = 0;
        const double t1 = mcf::now_ns();
        while ((gen = x->header->generation.load(std::memory_order_acquire)) == 0 && mcf::now_ns() - t1 < 50e6) _mm_pause();
        if (gen == 0 || gen == last || x->header->world != (uint32_t)world) { usleep(200); continue; }
        if (x->slots[rank].present.exchange(1, std::memory_order_acq_rel) != 0) { usleep(200); continue; }      // somebody has been this rank here before
        bool replaced = false;
        uint64_t spins = 0;
        while (!all_present()) {
            _mm_pause();
            if ((++spins & 0x3FFFF) == 0) {
                if (!still_linked(name, sb.st_ino)) { replaced = true; break; }      // rank 0 made a new one: this was a leftover
                if (mcf::now_ns() - t0 > kOpenTimeoutNs) return give_up(MCF_ERR_TIMEOUT, "not all ranks arrived within 60 s");
            }
        }
        if (replaced) continue;
        { std::lock_guard<std::mutex> lock(g_seen_mutex); g_seen[name] = gen; }
        *out = x;
        return MCF_OK;
    }
}

void mcf_exchange_close(mcf_exchange *x)
{
    if (!x) return;
    if (x->header) munmap(x->header, x->bytes);
    if (x->fd >= 0) close(x->fd);
    if (x->rank == 0) shm_unlink(x->name.c_str());     // the owner alone; a rank that is still exchanging keeps its mapping
    delete x;
}

int mcf_exchange_all_gather(mcf_exchange *x, const mcf_candidate *mine, mcf_candidate *all)
{
    if (!x || !mine || !all) return mcf::fail(MCF_ERR_INVALID, "mcf_exchange_all_gather: null argument");
    // Two slots per rank, used alternately: a rank can only be one exchange ahead of the slowest one (it needs everybody's record of
    // exchange k before it publishes k + 1), so the record of exchange k is never overwritten before everybody has read it.
    const uint64_t seq = ++x->seq;
    XSlot *row = x->slots + (size_t)(seq & 1) * x->world;
    row[x->rank].rec = *mine;
    row[x->rank].seq.store(seq, std::memory_order_release);
    double t0 = 0;
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) { all[r] = *mine; continue; }
        uint64_t spins = 0;
        while (row[r].seq.load(std::memory_order_acquire) != seq) {
            _mm_pause();
            if ((++spins & 0xFFFFF) == 0) {
                if (t0 == 0) t0 = mcf::now_ns();
                else if (mcf::now_ns() - t0 > 60e9) return mcf::fail(MCF_ERR_TIMEOUT, "rank %d did not publish exchange %llu within 60 s", r, (unsigned long long)seq);
            }
        }
        all[r] = row[r].rec;
    }
    return MCF_OK;
}

}  // extern "C"
