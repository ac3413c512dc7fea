// exchange.cpp -- the host-side MINLOC exchange of sharded engines ("each GPU writing its pair to ... host memory and the
// host reducing 8 pairs"): one process per GPU, every rank publishes its 16-byte candidate in POSIX shared memory and reads the others'.
// No collective library, no device involvement: the records are already on the host when a search ends.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstring>
#include <string>

#include <immintrin.h>

#include "common.h"

namespace {

// one cache line pair per rank and parity: the record and the sequence number that publishes it
struct alignas(128) XSlot {
    mcf_candidate rec;
    std::atomic<uint64_t> seq;
};

}  // namespace

struct mcf_exchange {
    std::string name;
    int rank = 0, world = 1, fd = -1;
    size_t bytes = 0;
    XSlot *slots = nullptr;      // [2][world]
    uint64_t seq = 0;
};

extern "C" {

int mcf_exchange_open(mcf_exchange **out, const char *name, int32_t rank, int32_t world)
{
    if (!out || !name || name[0] != '/' || world < 1 || rank < 0 || rank >= world) return mcf::fail(MCF_ERR_INVALID, "mcf_exchange_open: bad arguments (name must start with '/')");
    *out = nullptr;
    mcf_exchange *x = new mcf_exchange();
    x->name = name; x->rank = rank; x->world = world;
    x->bytes = sizeof(XSlot) * 2 * (size_t)world;
    x->fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (x->fd < 0) { delete x; return mcf::fail(MCF_ERR_IO, "shm_open(%s) failed", name); }
    if (ftruncate(x->fd, (off_t)x->bytes) != 0) { close(x->fd); delete x; return mcf::fail(MCF_ERR_IO, "ftruncate(%s) failed", name); }
    void *m = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
    if (m == MAP_FAILED) { close(x->fd); delete x; return mcf::fail(MCF_ERR_IO, "mmap(%s) failed", name); }
    x->slots = (XSlot *)m;
    // a segment left over by an earlier run under the same name: this rank's slots start from zero (the caller's barrier follows)
    for (int par = 0; par < 2; ++par) {
        XSlot &sl = x->slots[(size_t)par * world + rank];
        sl.rec = mcf_candidate{0, 0xFFFFFFFFu, -1};
        sl.seq.store(0, std::memory_order_release);
    }
    *out = x;
    return MCF_OK;
}

void mcf_exchange_close(mcf_exchange *x)
{
    if (!x) return;
    if (x->slots) munmap(x->slots, x->bytes);
    if (x->fd >= 0) close(x->fd);
    shm_unlink(x->name.c_str());          // the mapping of a rank that is still exchanging stays valid; the last close frees the memory
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
