"""N > 1 path on CPU: two `gloo` ranks each own an arc shard, exchange their 32-byte candidates with an all-gather,
resolve the global entering arc with the product's MINLOC (mcf_resolve_candidates) and drive replicated copies of the
product's sequential host driver.  The only thing that is not product code is the per-shard reduced-cost scan, done here
with numpy / the oracle because there is no GPU (on the GPU box that scan is mcf_engine_find_entering_local and the
exchange is one ncclAllGather: mcf_engine_find_entering_sharded)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NONE = 0xFFFFFFFF


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_candidate(M, L, a, m_s, shard, rule, optimized, block, next_arc):
    """Best candidate of arcs [b, e) under the rule's ordering: numpy restatement of the kernel's per-shard result."""
    b, e = shard
    cand = L.Candidate(0, NONE, -1)
    if e <= b:
        return cand
    idx = np.arange(b, e)
    rc = a["state"][b:e].astype(np.int64) * (a["cost"][b:e] + a["pi"][a["source"][b:e]] - a["pi"][a["target"][b:e]])
    elig = rc < 0
    if not elig.any():
        return cand
    idx, rc = idx[elig], rc[elig]
    if rule == M.PivotRule.BestEligible:
        k = np.lexsort((idx, rc))[0]
        return L.Candidate(int(rc[k]), int(idx[k]), int(idx[k]))
    na = 0 if next_arc >= m_s else next_arc
    pos = (idx - na) % m_s
    if rule == M.PivotRule.FirstEligible:
        k = int(np.argmin(pos))
        return L.Candidate(int(rc[k]), int(pos[k]), int(idx[k]))
    r = pos // block
    rank = 2 * r
    if optimized and next_arc < m_s and (m_s - next_arc) % block:
        rstar = (m_s - next_arc) // block
        rank = rank + ((r == rstar) & (idx < next_arc))
    k = np.lexsort((pos, rc, rank))[0]
    if not optimized:
        return L.Candidate(int(rc[k]), int(pos[k]), int(idx[k]))
    # the range key of the optimized flavour: best arc of the first of the two ranges ([next_arc, m_s), then [0, next_arc)) with an eligible arc
    q = np.lexsort((pos, rc, (idx < na).astype(np.int64)))[0]
    return L.Candidate(int(rc[k]), int(pos[k]), int(idx[k]), int(rc[q]), int(pos[q]), int(idx[q]))


def _rank_main(rank, world, port, fixture, rule, optimized, out_dir, exchange="gloo", vw=4):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import mincostflow_amd as M
    from helpers import load
    from mincostflow_amd import _lib as L
    from oracle import ns_oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = load(fixture)
    ns = M.NetworkSimplex(p.n, p.src, p.tgt).set_problem(p.lower, p.upper, p.cost, p.supply)
    assert ns.begin() == 0
    m_s = ns.internal()["search_arc_num"]
    shard = M.shard_range(m_s, rank, world)
    block = max(int(np.sqrt(m_s)), 10)
    next_arc, trace = 0, []
    xchg = None
    if exchange == "shm":           # the product's host exchange (mcf_exchange_*): POSIX shared memory instead of a collective
        xchg = M.HostExchange(f"/mcf_test_{port}", rank, world)
        dist.barrier()              # everybody has opened (and zeroed its slots) before the first exchange
    while True:
        a = ns.internal()
        mine = _local_candidate(M, L, a, m_s, shard, rule, optimized, block, next_arc)
        if xchg is not None:
            cands = xchg.all_gather(mine)
        else:
            send = torch.frombuffer(bytearray(bytes(mine)), dtype=torch.uint8)          # the 32-byte mcf_candidate record
            got = [torch.zeros(32, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(got, send)
            cands = [L.Candidate.from_buffer_copy(bytes(t.numpy().tobytes())) for t in got]
        found, arc, rc, next_arc = M.resolve_candidates(rule, optimized, m_s, block, next_arc, cands, vector_width=vw)
        if not found:
            break
        trace.append(arc)
        assert not ns.apply_pivot(arc)
    status = ns.finish()
    # the reference answer: one un-sharded oracle solve with the same rule
    o = O.Oracle(p, O.SEM_CSHARP_OPT if optimized else O.SEM_CSHARP, {0: O.RULE_FIRST, 1: O.RULE_BEST, 2: O.RULE_BLOCK}[rule], block_size=block, vector_width=vw)
    st_o, tr_o = o.solve(trace_cap=1 << 22)
    assert status == st_o == 1
    assert np.array_equal(np.array(trace, np.int32), tr_o), "sharded pivot sequence differs from the single-rank one"
    assert ns.get_total_cost() == o.total_cost and np.array_equal(ns.flows(), o.flow())
    # replicated host state must be identical on all ranks
    digest = torch.tensor([ns.get_total_cost(), int(ns.potentials().sum()), len(trace)], dtype=torch.int64)
    all_d = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_d, digest)
    assert all(torch.equal(all_d[0], d) for d in all_d)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(len(trace)))
    dist.barrier()
    if xchg is not None:
        xchg.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("fixture,rule,optimized,vw", [("netgen_8_08a", 1, True, 4), ("netgen_8_08a", 2, True, 4), ("netgen_8_08a", 2, True, 0), ("transport_40x30", 2, True, 2),
                                                       ("transport_40x30", 2, False, 4), ("circulation_100_0_10", 0, True, 4)])
def test_two_rank_sharded_solve_over_gloo(tmp_path, fixture, rule, optimized, vw):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, fixture, rule, optimized, str(tmp_path), "gloo", vw), nprocs=world, join=True)
    counts = [int(open(tmp_path / f"ok{r}").read()) for r in range(world)]
    assert counts[0] == counts[1] > 0


@pytest.mark.parametrize("vw", [4, 0])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_over_the_shared_memory_exchange(tmp_path, world, vw):
    """The same replicated pivot loop with the candidates exchanged by mcf_exchange_all_gather (what mcf_ns_set_sharding_host uses on the
    GPU box) instead of a gloo collective: thousands of lock-step exchanges between processes, identical pivots on every rank."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, "netgen_8_08a", 2, True, str(tmp_path), "shm", vw), nprocs=world, join=True)
    counts = [int(open(tmp_path / f"ok{r}").read()) for r in range(world)]
    assert len(set(counts)) == 1 and counts[0] > 0


def test_exchange_rejects_bad_arguments_and_survives_a_stale_segment():
    import mincostflow_amd as M
    from mincostflow_amd import _lib as L
    with pytest.raises(M.McfError):
        M.HostExchange("no-leading-slash", 0, 1)
    with pytest.raises(M.McfError):
        M.HostExchange("/mcf_test_bad", 2, 2)
    name = f"/mcf_test_stale_{os.getpid()}"
    a = M.HostExchange(name, 0, 1)
    for k in range(5):
        got = a.all_gather(L.Candidate(-k, k, k))
        assert (got[0].reduced_cost, got[0].pos, got[0].arc) == (-k, k, k)
    # a second user of the same name (the first never closed: a crashed run) starts from a clean segment of its own
    b = M.HostExchange(name, 0, 1)
    got = b.all_gather(L.Candidate(-7, 7, 7))
    assert got[0].arc == 7
    a.close(); b.close()


def _reopen_main(rank, world, name, out_dir):
    """Several solves' worth of open / exchange / close under ONE name, the ranks arriving at each open and leaving each close at different
    times (what mcf_ns_prepare + mcf_ns_solve do, twice in a row, without any barrier of the caller's)."""
    sys.path.insert(0, ROOT)
    import random
    import time

    import mincostflow_amd as M
    from mincostflow_amd import _lib as L
    rnd = random.Random(1000 + rank)
    for round_ in range(6):
        time.sleep(rnd.random() * 0.03)
        x = M.HostExchange(name, rank, world)         # the barrier sits inside the library (mcf_exchange_open)
        for k in range(300):
            got = x.all_gather(L.Candidate(-(1000 * round_ + k) - rank, 7 * k + rank, rank, -k, k, round_))
            for r in range(world):
                assert (got[r].reduced_cost, got[r].pos, got[r].arc, got[r].range_cost, got[r].range_pos, got[r].range_arc) == \
                       (-(1000 * round_ + k) - r, 7 * k + r, r, -k, k, round_), (round_, k, r)
        time.sleep(rnd.random() * 0.03)
        x.close()
    open(os.path.join(out_dir, f"reopen{rank}"), "w").write("ok")


@pytest.mark.parametrize("leftover", ["of-another-world-size", "of-this-world-size-with-every-rank-present"])
@pytest.mark.parametrize("world", [2, 3])
def test_exchange_reopened_under_one_name_by_ranks_that_are_out_of_step(tmp_path, world, leftover):
    """Round 2's advisor finding: every rank unlinked the segment on close and re-created it on open, so a late closer could unlink what an
    early opener had just made.  Now rank 0 owns create / unlink, the segment carries a generation nonce and presence flags, and
    mcf_exchange_open is itself the barrier.  A leftover of a crashed run sits under the name when the ranks start."""
    import struct

    import torch.multiprocessing as mp
    name = f"/mcf_test_reopen_{os.getpid()}_{world}_{len(leftover)}"
    import mincostflow_amd as M
    if leftover == "of-another-world-size":
        stale = M.HostExchange(name, 0, 1)
        stale._h = None     # never closed: the segment stays linked
    else:
        # exchange.cpp's layout: a 128-byte header {generation u64, world u32}, then 2 * world slots of 128 bytes {record 32 B, seq u64, present u32}
        blob = bytearray(128 + 128 * 2 * world)
        struct.pack_into("<QI", blob, 0, 0x1234567, world)
        for i in range(2 * world):
            struct.pack_into("<QI", blob, 128 + 128 * i + 32, 99, 1)
        with open("/dev/shm" + name, "wb") as f:
            f.write(blob)
    mp.spawn(_reopen_main, args=(world, name, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"reopen{r}").exists() for r in range(world))
    assert not os.path.exists("/dev/shm" + name)          # rank 0 unlinked its last segment
