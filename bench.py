#!/usr/bin/env python3
"""bench.py -- pivots/s and solve ms of the MI355X network-simplex pivot engine on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[2] = NETGEN-like 100 000 nodes / 300 000 arcs (seed 13502460, 316 sources,
316 sinks), Best-Eligible full-arc scan, int64, behind EnableOptimizedPivot(true).  One "step" = one full
Solve() pivot loop of that instance with the SoA arc arrays and potentials ALREADY resident in HBM (mcf_ns_prepare runs
before the timed region); value = pivots performed by all ranks / max-over-ranks wall time of the K steps.

N > 1 (`"scaling": "weak"`): `value` keeps the N = 1 metric and workload -- one config-3 solve per GPU (seed + rank), no data-path
collective -- so the driver's per-N values are comparable.  The line ALSO carries, by default, the path BASELINE.json configs[4]
names: ONE 1M-node / 8M-arc instance with its arcs sharded over the ranks and a MINLOC exchange per pivot (block "sharded":
ranks as seen by the communicator, us per pivot for the RCCL all-gather exchange and for the shared-memory host exchange, and the
single-GPU figure for the same pivots).  Without a launcher `--gpus N` starts the N ranks itself (before anything touches a GPU);
a WORLD_SIZE that differs from --gpus is refused, and `n_gpus` is always the number of ranks that ran.

The JSON line also carries
  roofline      dominant kernel = the resident entering-arc scan grid (one dispatch serves every pivot of a solve through a mailbox).
                achieved = algorithmic bytes per launch (requests served x (17*m_s + 8*(n+1))) / the launch's duration,
                measured in this run with HIP events attached to the dispatch on the engine's stream (hipExtLaunchKernelGGL start/stop =
                dispatch begin..end, what rocprofv3 --kernel-trace reports).  The grid waits for the host between requests, so this is
                the END-TO-END figure; "in_kernel" (device clock, request seen -> record published) and "scan_dispatch" (the same scan as
                one dispatch per search, timed alone) isolate the scan itself.
  cpu_baseline  the CPU oracle (C restatement of the reference, C# semantics) timed on this host, one core, on a bounded
                sample of the same workload; kind "port" (the C# reference has no toolchain here, LEMON is unbuildable: DESIGN.md)
  scan_microbench  scan-only kernel durations at larger sizes, where the HBM roofline fraction is adjudicated
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
SEED = 13502460
TRAFFIC_FILE = "traffic_r03.json"   # PMC FETCH_SIZE of a separate rocprofv3 --pmc pass (gpurun refuses --pmc beside tracing)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config3", choices=["config2", "config3", "config4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-validator", action="store_true", help="skip the SolutionValidator measurement")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact config-2 / config-4 blocks")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget of the cpu_baseline sample")
    ap.add_argument("--sharded-pivots", type=int, default=-1, help="N>1: pivots of config 5 timed with the arcs sharded over the ranks (-1 = 5000 when N > 1, 2000 with one rank; 0 = skip)")
    ap.add_argument("--dispatch", action="store_true", help="one scan dispatch per search instead of the resident grid")
    ap.add_argument("--concurrent", type=int, default=4, help="extra measurement: this many independent solves at once on the GPU (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (gloo: rehearsals on one GPU)")
    return ap.parse_args()


def workload(name, seed):
    import mincostflow_amd as M
    if name == "config2":
        return (M.netgen_like(seed, 10_000, 30_000, 100, 100), M.PivotRule.BlockSearch, 32,
                "NETGEN-like 10k nodes / 30k arcs, Block Search, int32")
    if name == "config4":
        return (M.assignment(42 + seed - SEED, 1000, 1, 100), M.PivotRule.BestEligible, 64,
                "assignment 1000x1000 (1M arcs), Best Eligible, int64")
    return (M.netgen_like(seed, 100_000, 300_000, 316, 316), M.PivotRule.BestEligible, 64,
            "NETGEN-like 100k nodes / 300k arcs, Best-Eligible full-arc scan, int64")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


VECTOR_WIDTH = 4               # Vector<long>.Count of the reference's host that every EnableOptimizedPivot(true) leg reproduces (x64; BSPO.cs:74-99)


def cpu_baseline(g, rule, budget_s, reference_default=False, vector_width=VECTOR_WIDTH):
    """Oracle (C port of the reference, EnableOptimizedPivot semantics), same instance and rule, one core; with the reference's three
    phase buckets (SolverMetrics: pivot search / tree update / potential update).
    reference_default: the plain rule with the reference's auto-configuration instead -- what `new NetworkSimplex(g).Solve()` runs.
    vector_width: Vector<long>.Count of the machine whose optimized Block Search is reproduced (4 = x64: a boundary hit in the "SIMD" part
    scans on to the end of the range; 0 = not hardware accelerated: stop at the block boundary).  Only that rule reads it."""
    from oracle import ns_oracle as O
    core, affinity_before = pin_to_one_core()
    p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
    if reference_default:
        o = O.Oracle(p, O.SEM_CSHARP, {0: O.RULE_FIRST, 1: O.RULE_BEST, 2: O.RULE_BLOCK}[rule], auto_config=True)
    else:
        o = O.Oracle(p, O.SEM_CSHARP_OPT, {0: O.RULE_FIRST, 1: O.RULE_BEST, 2: O.RULE_BLOCK}[rule], vector_width=vector_width)
    o.enable_timing()
    o.init()
    done, ended, t0 = 0, False, time.perf_counter()
    chunk = 256
    while not ended and time.perf_counter() - t0 < budget_s:
        ended, k = o.run_pivots(chunk)
        done += k
        chunk = min(chunk * 2, 1 << 16)
    dt = time.perf_counter() - t0
    restore_affinity(affinity_before)
    ph = o.phase_us()
    sample = (f"whole solve ({done} pivots)" if ended else f"first {done} pivots of the same solve") + f", {dt:.1f} s of CPU work"
    return {"value": done / dt, "unit": "pivots/s", "cores": 1, "kind": "port", "sample": sample,
            "semantics": "plain rule, auto-configured" if reference_default else f"EnableOptimizedPivot(true), Vector<long>.Count = {vector_width}" + (" (not hardware accelerated)" if vector_width == 0 else ""),
            "pinned_to_cpu": core,
            "host_cores_available": os.cpu_count(), "host_cpu": cpu_model(), "us_per_pivot": dt / max(done, 1) * 1e6,
            "phase_us_per_pivot": {"pivot_search": ph[0] / max(done, 1), "tree_update": ph[1] / max(done, 1), "potential_update": ph[2] / max(done, 1)},
            "solve_ms_if_whole": dt * 1e3 if ended else None,
            "not_timed": "LEMON (needs its CMake-generated config.h: unbuildable here) and the C# reference (no .NET toolchain in the image)"}


def pin_to_one_core():
    """The CPU legs run on ONE core (the reference is single-threaded): the core this thread is on when the leg starts.
    Returns (core, the affinity to give back to restore_affinity when the leg is over)."""
    before = os.sched_getaffinity(0)
    try:
        import ctypes
        here = ctypes.CDLL(None).sched_getcpu()
        if here < 0 or here not in before:
            here = min(before)
        os.sched_setaffinity(0, {here})
        return here, before
    except (AttributeError, OSError):
        return -1, before


def restore_affinity(before):
    try:
        os.sched_setaffinity(0, before)
    except OSError:
        pass


def other_config(M, name, local_rank, cpu_seconds):
    """BASELINE.json configs[1] / configs[3] in one compact block: one warm solve on the GPU, the same-rule CPU port beside it.
    Block Search (config 2) runs in the three forms the reference has: EnableOptimizedPivot(true) as an x64 host executes it
    (Vector<long>.Count = 4: nearly every search scans on to the end of the range, BSPO.cs:84-99), the same class without hardware
    vectors (stops at the block boundary), and the plain BlockSearchPivot with the reference's auto-configuration (its default)."""
    g, rule, width, desc = workload(name, SEED)

    def leg(optimized, vw, cpu_s):
        def solve():
            ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(optimized).set_vector_width(vw).set_device(local_rank, width, 0, 0).prepare()
            assert ns.solve() == M.SolverStatus.Optimal
            return ns
        solve()
        ns = solve()
        m = ns.get_metrics(); it = max(m["iterations"], 1)
        out = {"pivots": m["iterations"], "solve_ms": m["loop_us"] / 1e3, "pivots_per_s": it / (m["loop_us"] / 1e6),
               "us_per_pivot": m["loop_us"] / it, "pivot_search_us": m["pivot_search_us"] / it, "total_cost": ns.get_total_cost(), "int_width": m["int_width"]}
        del ns
        if cpu_s > 0:
            b = cpu_baseline(g, rule, cpu_s, reference_default=not optimized, vector_width=vw)
            out["cpu_same_rule"] = {"pivots_per_s": b["value"], "us_per_pivot": b["us_per_pivot"], "sample": b["sample"], "cores": 1, "kind": "port",
                                    "solve_ms_if_whole": b["solve_ms_if_whole"]}
            out["gpu_over_cpu_per_pivot"] = b["us_per_pivot"] / out["us_per_pivot"]
        return out

    out = {"workload": desc, "semantics": f"EnableOptimizedPivot(true), Vector<long>.Count = {VECTOR_WIDTH}"}
    out.update(leg(True, VECTOR_WIDTH, cpu_seconds))
    if rule == M.PivotRule.BlockSearch:
        out["same_class_without_hardware_vectors"] = dict(leg(True, 0, cpu_seconds), semantics="EnableOptimizedPivot(true), Vector.IsHardwareAccelerated == false")
        out["plain_rule_auto_configured"] = dict(leg(False, VECTOR_WIDTH, cpu_seconds), semantics="new NetworkSimplex(g).Solve(): plain BlockSearchPivot, auto-configuration on")
    return out


def sharded_leg(M, torch, dist, args, rank, world, dev, barrier, red_dev):
    """BASELINE.json configs[4]: ONE NETGEN-like 1M-node / 8M-arc instance, arcs sharded over the ranks, one MINLOC exchange per pivot.
    Every rank runs the same host loop on its own shard; timed: the first --sharded-pivots pivots, max over ranks."""
    import numpy as np
    P = args.sharded_pivots
    g5 = M.netgen_like(SEED, 1_000_000, 8_000_000, 1000, 1000)

    # ranks that share a GPU (gloo rehearsal) cannot all keep a full resident grid on it: one dispatch per search there
    shard_flags = M.ENGINE_DISPATCH if world > M.device_count() else 0

    def run(configure):
        ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(dev, 64, 0, shard_flags)
        configure(ns5)
        ns5.set_pivot_limit(P).record_trace(P).prepare()
        barrier()                                  # also: every rank has opened the exchange before the first pivot
        ts = time.perf_counter()
        ns5.solve()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        m5 = ns5.get_metrics()
        return float(t.item()), m5, ns5.trace()

    out = {"workload": "NETGEN-like 1M nodes / 8M arcs, Best Eligible, int64, arcs sharded over the ranks (contiguous ranges, potentials replicated)",
           "ranks": world, "pivots_timed": P, "search_arcs": 9_000_000, "variants": {}}
    if world == 1:
        out["note"] = "a world of ONE rank: the sharded path's code (shard engine, exchange, MINLOC, RCCL all-gather) runs on this box, there is nothing to scale"
    port = os.environ.get("MASTER_PORT", "0")
    sec, m5, tr_host = run(lambda ns: ns.set_sharding_host(f"/mcf_bench_{port}_{os.getppid()}", rank, world))
    out["variants"]["host_exchange"] = {"exchange": "16-byte records through POSIX shared memory (mcf_exchange_all_gather), resident or dispatch scan per shard",
                                        "ranks": world, "pivots": m5["iterations"], "seconds": sec, "us_per_pivot": sec / max(m5["iterations"], 1) * 1e6,
                                        "pivots_per_s": m5["iterations"] / sec, "shard_engine_resident": bool(m5["engine"]["resident"]),
                                        "shard_candidate_cache": bool(m5["engine"]["candidates"]), "rank0_searches_answered_on_the_host": m5["engine"]["host_decided"],
                                        "rank0_device_searches": m5["engine"]["resident_requests"] if m5["engine"]["resident"] else m5["engine"]["scan_launches"]}
    same = True
    # the same pivots on ONE GPU (rank 0 alone, un-sharded), the figure the sharded ones have to beat
    if rank == 0:
        ns1 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(dev, 64, 0, 0)
        ns1.set_pivot_limit(P).record_trace(P).prepare()
        ts = time.perf_counter()
        ns1.solve()
        sec1 = time.perf_counter() - ts
        m1 = ns1.get_metrics()
        out["single_gpu_same_pivots"] = {"pivots": m1["iterations"], "seconds": sec1, "us_per_pivot": sec1 / max(m1["iterations"], 1) * 1e6,
                                         "candidate_cache": bool(m1["engine"]["candidates"]), "searches_answered_on_the_host": m1["engine"]["host_decided"]}
        same = same and bool(np.array_equal(ns1.trace(), tr_host))
        del ns1
    barrier()
    if args.backend == "nccl":
        # LAST, and under a watchdog: this is the one leg that no box of the builder's could run with more than one rank (one-GPU boxes), and a
        # collective that never completes must not cost the whole line.  After a time-out the process group is left alone (no further collective).
        import threading
        box = {}

        def rccl_leg():
            ident = torch.from_numpy(M.comm_unique_id() if rank == 0 else np.zeros(128, "uint8")).cuda()
            if dist is not None:
                dist.broadcast(ident, src=0)
            box["res"] = run(lambda ns: ns.set_sharding(ident.cpu().numpy(), rank, world))

        th = threading.Thread(target=rccl_leg, daemon=True)
        th.start()
        th.join(timeout=180.0)
        if "res" in box:
            sec, m5, tr_rccl = box["res"]
            out["variants"]["rccl_all_gather"] = {"exchange": "scan records folded on the device -> ncclAllGather of 16 B per rank on the engine's stream -> MINLOC on the host, every pivot",
                                                  "ranks": int(m5["engine"]["comm_ranks"]), "pivots": m5["iterations"], "seconds": sec,
                                                  "us_per_pivot": sec / max(m5["iterations"], 1) * 1e6, "pivots_per_s": m5["iterations"] / sec}
            same = same and bool(np.array_equal(tr_rccl, tr_host))
        else:
            out["variants"]["rccl_all_gather"] = {"error": "did not finish within 180 s on this rank; the line was printed without it"}
            out["abandon_process_group"] = True
    out["identical_pivot_sequence"] = same
    return out


def microbench():
    """Scan-only kernel (one dispatch, HIP events) at sizes where bandwidth, not latency, decides.
    cold = after reading 512 MB of other data (evicts L2 and the 256 MB Infinity Cache)."""
    import numpy as np

    import mincostflow_amd as M
    rng = np.random.default_rng(7)
    out = []

    def run(label, n_nodes, m_s, src, tgt, cost, state, pi, endpoints, env=None, width=64):
        for k, v in (env or {}).items():
            os.environ[k] = str(v)
        eng = M.PivotEngine(n_nodes, m_s, m_s, rule=M.PivotRule.BestEligible, int_width=width, flags=M.ENGINE_DISPATCH)
        for k in (env or {}):
            os.environ.pop(k, None)
        eng.upload(src, tgt, cost, state, pi)
        st = eng.stats()
        survey = st["bytes_per_scan"]              # 17 B per arc + the potentials once
        nbytes = st["scan_bytes_read"]             # what this layout's scan has to read (9 B per arc in the RC layout: no potential gathers)
        warm = eng.bench_scan(reps=20)
        cold = eng.bench_scan(reps=8, cold=True, flush_bytes=512 << 20)
        out.append({"case": label, "arcs": m_s, "nodes": n_nodes, "dtype": f"i{width}", "endpoints": endpoints,
                    "layout": "reduced costs kept per arc (RC): state + reduced cost streamed, nothing gathered" if st["rc_layout"] else
                              ("potentials in LDS" if n_nodes <= 16384 else "SoA arcs + two potential gathers per arc"),
                    "bytes": nbytes, "survey_bytes": survey,
                    "grid": f"{st['scan_workgroups']}x{st['scan_threads']}", "potentials_in_lds": n_nodes <= 16384,
                    "warm_us": warm[0] / 1e3, "cold_us": cold[0] / 1e3, "warm_GBs": nbytes / warm[0], "cold_GBs": nbytes / cold[0],
                    "warm_frac_of_hbm_peak": nbytes / warm[0] / HBM_PEAK_GBS, "cold_frac_of_hbm_peak": nbytes / cold[0] / HBM_PEAK_GBS,
                    "cold_GBs_in_survey_bytes": survey / cold[0]})

    # SURVEY.md 8d sizes: m_s in {4e5, 1e6, 8e6, 6.4e7}, uniform random arrays
    for label, m_s, n in (("config-3 size, uniform random end points", 400_000, 100_001),
                          ("1M arcs, uniform random end points over 125k nodes", 1_000_000, 125_001),
                          ("8M arcs, uniform random end points over 1M nodes", 8_000_000, 1_000_001),
                          ("dense (assignment-like), potentials fit LDS", 64_000_000, 2_001),
                          ("uniform random end points over 1M nodes (gather worst case)", 64_000_000, 1_000_001)):
        arrs = (rng.integers(0, n, m_s, dtype=np.int32), rng.integers(0, n, m_s, dtype=np.int32),
                rng.integers(-10 ** 4, 10 ** 4, m_s, dtype=np.int64), rng.integers(-1, 2, m_s, dtype=np.int8),
                rng.integers(-10 ** 9, 1, n, dtype=np.int64))
        run(label, n, m_s, *arrs, "uniform random")
        # the same arrays with 32-bit costs and potentials on the device (13 B per arc + 4 B per node; SURVEY.md 8d asks for both widths)
        run(label, n, m_s, *arrs, "uniform random", width=32)
        del arrs
    # BASELINE.json configs[4] as generated: arcs grouped by tail node, random heads
    g5 = M.netgen_like(SEED, 1_000_000, 8_000_000, 1000, 1000)
    ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
    assert ns5.begin() == 0
    it = ns5.internal()
    ms = it["search_arc_num"]
    run("NETGEN-like 1M nodes / 8M arcs start basis (config 5 arrays)", g5.node_count + 1, ms, it["source"][:ms], it["target"][:ms],
        it["cost"][:ms], it["state"][:ms], it["pi"], "generator order: grouped by tail, random heads")
    run("the same arrays with the gathering scan (bucketed layout, round 1)", g5.node_count + 1, ms, it["source"][:ms], it["target"][:ms],
        it["cost"][:ms], it["state"][:ms], it["pi"], "generator order: grouped by tail, random heads", env={"MCF_HIP_RC": 0})
    return out


def large_instance_sample(M, local_rank, with_cpu, gpu_pivots=3000, cpu_seconds=4.0):
    """BASELINE.json configs[4] on ONE GPU: the first pivots of NETGEN-like 1M nodes / 8M arcs, Best Eligible (9M search arcs per scan,
    RC layout, one dispatch per search).  Same-rule CPU port on the same first pivots beside it."""
    g5 = M.netgen_like(SEED, 1_000_000, 8_000_000, 1000, 1000)
    ns = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True)
    ns.set_device(local_rank, 64, 0, 0).set_pivot_limit(gpu_pivots).record_trace(gpu_pivots).prepare()
    ns.solve()
    gpu_trace = ns.trace()
    m = ns.get_metrics(); it = max(m["iterations"], 1)
    out = {"workload": "NETGEN-like 1M nodes / 8M arcs (config 5 on one GPU), Best Eligible, int64, first pivots only", "pivots": m["iterations"],
           "us_per_pivot": m["loop_us"] / it, "pivot_search_us": m["pivot_search_us"] / it, "pivots_per_s": it / (m["loop_us"] / 1e6),
           "engine_mode": "resident grid" if m["engine"]["resident"] else "one dispatch per search",
           "layout": "reduced costs kept per arc (RC)" if m["engine"]["rc_layout"] else "SoA arcs + potential gathers",
           "candidate_cache": bool(m["engine"]["candidates"]), "searches_answered_on_the_host": m["engine"]["host_decided"],
           "device_searches": m["engine"]["resident_requests"] if m["engine"]["resident"] else m["engine"]["scan_launches"],
           "search_arcs": m["search_arc_num"], "bytes_read_per_scan": m["engine"]["scan_bytes_read"]}
    del ns
    # the same pivots with every search on the device (what the whole solve's late phase does too: big subtrees leave the cache nothing to decide)
    os.environ["MCF_HIP_CANDIDATES"] = "0"
    try:
        nsd = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True)
        nsd.set_device(local_rank, 64, 0, 0).set_pivot_limit(gpu_pivots).record_trace(gpu_pivots).prepare()
        nsd.solve()
    finally:
        os.environ.pop("MCF_HIP_CANDIDATES", None)
    md = nsd.get_metrics(); itd = max(md["iterations"], 1)
    out["every_search_on_the_device"] = {"pivots": md["iterations"], "us_per_pivot": md["loop_us"] / itd, "pivot_search_us": md["pivot_search_us"] / itd,
                                         "scan_GBps_incl_round_trip": md["engine"]["scan_bytes_read"] / (md["pivot_search_us"] / itd) / 1e3,
                                         "identical_pivot_sequence": bool((nsd.trace() == gpu_trace).all())}
    del nsd
    # the same pivots with the arcs as 8 shards driven by ONE host thread, all eight engines on this one GPU (a rehearsal of mcf_ns_set_shard_group:
    # eight dispatches share the device, so this is the host-side cost of the sharded path plus 8 x a ninth of the scan, not a scaling figure)
    try:
        nsg = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible).enable_optimized_pivot(True).set_device(local_rank, 64, 0, 0)
        nsg.set_shard_group([local_rank] * 8).set_pivot_limit(gpu_pivots).prepare()
        nsg.solve()
        mg = nsg.get_metrics(); itg = max(mg["iterations"], 1)
        out["as_8_shards_on_this_one_gpu"] = {"pivots": mg["iterations"], "us_per_pivot": mg["loop_us"] / itg, "pivot_search_us": mg["pivot_search_us"] / itg,
                                              "what": "mcf_ns_set_shard_group([gpu] * 8): one host thread posts each search to 8 engines (1.125 M arcs each, RC layout, one dispatch "
                                                      "each) and reduces their 8 answers; on 8 GPUs the 8 dispatches run side by side"}
        del nsg
    except M.McfError as err:
        out["as_8_shards_on_this_one_gpu"] = {"error": str(err)}
    # what ONE of eight GPUs would do per pivot: an engine that holds the first eighth of the search arcs (1.125 M arcs: its windows of reduced
    # costs live in LDS), one search after a one-node shift and two state writes, 2000 times -- the sharded pivot's device leg without the exchange
    try:
        import numpy as np
        ns0 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
        assert ns0.begin() == 0
        it0 = ns0.internal(); ms = it0["search_arc_num"]
        eng = M.PivotEngine(g5.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, device=local_rank, shard=M.shard_range(ms, 0, 8))
        eng.upload(it0["source"][:ms], it0["target"][:ms], it0["cost"][:ms], it0["state"][:ms], it0["pi"])
        avg, mn = eng.bench_search(2000)
        st = eng.stats()
        out["one_eighth_shard_on_its_own_gpu"] = {"arcs": int(M.shard_range(ms, 0, 8)[1]), "searches": 2000, "us_per_search": avg / 1e3, "min_us": mn / 1e3,
                                                  "resident": bool(st["resident"]), "rc_layout": bool(st["rc_layout"]), "workgroups": st["scan_workgroups"],
                                                  "in_kernel_us_per_request": st["resident_scan_ns"] / max(st["resident_requests"], 1) / 1e3,
                                                  "what": "device leg of a sharded pivot on an 8-GPU node (each GPU holds such a shard: windows of reduced costs in LDS): "
                                                          "mailbox post -> scan -> 256 records merged on the host, no patches, no exchange (mcf_engine_bench_search)"}
        del eng, ns0
    except M.McfError as err:
        out["one_eighth_shard_on_its_own_gpu"] = {"error": str(err)}
    if with_cpu:
        b = cpu_baseline(g5, M.PivotRule.BestEligible, cpu_seconds)
        out["cpu_port_same_rule"] = {"us_per_pivot": b["us_per_pivot"], "sample": b["sample"], "cores": 1}
        out["gpu_over_cpu_per_pivot"] = b["us_per_pivot"] / out["us_per_pivot"]
        # the sampled pivots are the CPU port's pivots: its first entering arcs against the GPU run's trace
        from oracle import ns_oracle as O
        p5 = O.Problem(g5.node_count, g5.arc_count, g5.source, g5.target, g5.lower, g5.upper, g5.cost, g5.supply)
        o = O.Oracle(p5, O.SEM_CSHARP_OPT, O.RULE_BEST)
        o.init()
        same = 0
        for k in range(min(150, len(gpu_trace))):
            f, e = o.find_entering()
            if not f or e != int(gpu_trace[k]):
                break
            o.apply_pivot(e)
            same += 1
        out["pivots_identical_to_cpu_port"] = {"checked": min(150, len(gpu_trace)), "identical": same}
        assert same == min(150, len(gpu_trace)), "the GPU run's pivots differ from the CPU port's"
    return out


def potential_update_microbench(M, g3, local_rank):
    """the potential update against its bytes -- 20 per node of a list (12 with 32-bit potentials) for update_kernel; where
    reduced costs are kept per arc, update_rc_kernel also walks the nodes' arc lists (8 per node + 20 per arc-list entry).  Lists of distinct nodes
    spread over the whole table; HIP events around every launch (mcf_engine_bench_update)."""
    import numpy as np
    out = []

    def one(label, kernel, node_count, m_s, arrs, width, counts, env=None):
        for k, v in (env or {}).items():
            os.environ[k] = str(v)
        eng = M.PivotEngine(node_count, m_s, m_s, rule=M.PivotRule.BestEligible, int_width=width, device=local_rank, flags=M.ENGINE_DISPATCH)
        for k in (env or {}):
            os.environ.pop(k, None)
        eng.upload(*arrs)
        for cnt in counts:
            avg, mn, nb = eng.bench_update(min(cnt, node_count), reps=20)
            out.append({"case": label, "kernel": kernel, "dtype": f"i{width}", "nodes_in_list": min(cnt, node_count), "bytes": nb, "avg_us": avg / 1e3, "min_us": mn / 1e3,
                        "achieved_GBps": nb / avg, "frac_of_hbm_peak": nb / avg / HBM_PEAK_GBS})
        del eng

    ns3 = M.NetworkSimplex.from_problem(g3).set_pivot_rule(M.PivotRule.BestEligible)
    assert ns3.begin() == 0
    it = ns3.internal(); ms = it["search_arc_num"]
    arrs3 = (it["source"][:ms], it["target"][:ms], it["cost"][:ms], it["state"][:ms], it["pi"])
    one("config-3 arrays", "update_kernel<int64>", g3.node_count + 1, ms, arrs3, 64, (1_000, 16_000, 100_000))
    g2 = M.netgen_like(SEED, 10_000, 30_000, 100, 100)
    ns2 = M.NetworkSimplex.from_problem(g2).set_pivot_rule(M.PivotRule.BlockSearch)
    assert ns2.begin() == 0
    it2 = ns2.internal(); ms2 = it2["search_arc_num"]
    one("config-2 arrays", "update_kernel<int32>", g2.node_count + 1, ms2, (it2["source"][:ms2], it2["target"][:ms2], it2["cost"][:ms2], it2["state"][:ms2], it2["pi"]), 32, (1_000, 10_000))
    g5 = M.netgen_like(SEED, 1_000_000, 8_000_000, 1000, 1000)
    ns5 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
    assert ns5.begin() == 0
    it5 = ns5.internal(); ms5 = it5["search_arc_num"]
    arrs5 = (it5["source"][:ms5], it5["target"][:ms5], it5["cost"][:ms5], it5["state"][:ms5], it5["pi"])
    one("config-5 arrays, gathering layout", "update_kernel<int64>", g5.node_count + 1, ms5, arrs5, 64, (64_000, 1_000_000), env={"MCF_HIP_RC": 0})
    one("config-5 arrays, reduced costs kept per arc", "update_rc_kernel<int64>", g5.node_count + 1, ms5, arrs5, 64, (1_000, 64_000, 500_000))
    return {"rows": out,
            "note": "a list of k nodes is k scattered 8-byte updates into a table of up to 8 MB: at the sizes a pivot produces these are launch-latency-bound kernels "
                    "(an empty dispatch measures ~4 us by this method); the resident grids apply their lists inside the grid instead (in_kernel_phases_us)"}


def hbm_probe():
    """What this box's HBM delivers to a plain device-to-device copy (read + write), beside the nominal 8 TB/s used as `peak`."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1)
    for _ in range(3):
        b.copy_(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    del a, b
    return {"what": "torch device-to-device copy of 1 GiB (bytes read + bytes written) / time", "GBps": 2 * n / ms / 1e6, "ms": ms}


def validator_bench(M, g, ns, local_rank, with_cpu):
    """SolutionValidator on the solved instance and on config 5's shape: inputs resident, HIP events around the run."""
    import numpy as np
    out = []

    def one(label, n, m, src, tgt, lower, upper, cost, supply, flow, pi, reported, cpu):
        v = M.SolutionValidator(n, m, local_rank).upload_network(src, tgt, lower, upper, cost, supply).upload_solution(flow, pi)
        runs = [v.run(M.SupplyType.Geq, reported) for _ in range(12)][2:]
        us = sum(r["kernel_us"] for r in runs) / len(runs)
        r = runs[-1]
        entry = {"case": label, "arcs": m, "nodes": n, "valid": r["valid"], "objective": r["objective"], "dual_cost": r["dual_cost"],
                 "kernels": "memset + validate_arcs + validate_nodes + validate_fold", "avg_us": us, "min_us": min(x["kernel_us"] for x in runs),
                 "algorithmic_bytes": r["algorithmic_bytes"], "achieved_GBps": r["algorithmic_bytes"] / us / 1e3,
                 "frac_of_hbm_peak": r["algorithmic_bytes"] / us / 1e3 / HBM_PEAK_GBS}
        if cpu:
            from oracle import validator as V
            t0 = time.perf_counter()
            ref = V.validate(n, src, tgt, lower, upper, cost, supply, V.GEQ, flow, pi, reported)
            entry["cpu_restatement_ms"] = (time.perf_counter() - t0) * 1e3
            entry["cpu_kind"] = "port (numpy, 1 thread)"
            entry["matches_cpu"] = all(ref[k] == r[k] for k in ("valid", "objective", "dual_cost", "errors", "first"))
        out.append(entry)
        del v

    one("config-3 optimum (the timed solve's flows and potentials)", g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply,
        ns.flows(), ns.potentials(), ns.get_total_cost(), with_cpu)
    rng = np.random.default_rng(8)
    n, m = 1_000_000, 8_000_000
    src = rng.integers(0, n, m, dtype=np.int32); tgt = rng.integers(0, n, m, dtype=np.int32)
    upper = rng.integers(1, 1000, m).astype(np.int64); cost = rng.integers(1, 10001, m).astype(np.int64)
    flow = np.where(rng.random(m) < 0.12, rng.integers(0, 1000, m), 0).astype(np.int64)          # ~n arcs carry flow, like a basis
    pi = -rng.integers(0, 10 ** 7, n).astype(np.int64)
    supply = np.zeros(n, np.int64); np.add.at(supply, src, flow); np.subtract.at(supply, tgt, flow)
    one("config-5 shape, random solution-like vectors", n, m, src, tgt, np.zeros(m, np.int64), upper, cost, supply, flow, pi, 0, with_cpu)
    return out


def pin_near_gpu(dev):
    """Best effort: run this rank's host thread on the cores of the GPU's NUMA node (the pivot loop is a host <-> device latency chain)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            # MCF_BENCH_PIN=core: one core of that node (the one this thread is on, if it belongs to it) instead of the whole node
            if os.environ.get("MCF_BENCH_PIN") == "core":
                here = os.sched_getcpu()
                cpus = {here} if here in cpus else {min(cpus)}
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the N ranks ourselves.  Nothing has touched the GPU yet (no torch.cuda / HIP call, no exec of an initialised process).
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    # stdout carries the ONE JSON line and nothing else: whatever a library prints there (RCCL's version banner on some boxes) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to label a {world}-rank run as {args.gpus} GPUs")
    if args.sharded_pivots < 0:
        args.sharded_pivots = 5000 if world > 1 else 2000      # one rank: the same legs with a world of 1 (the code paths on this box, not a scaling figure)

    # Extra leg, BEFORE this process touches the GPU: K independent solves with the CUs partitioned per solve (mcf_ns_set_device_share: every
    # solver's grid gets 256 / K workgroups, i.e. CUs of its own).  In a process of its own: K resident grids need K hardware queues and
    # GPU_MAX_HW_QUEUES is read when HIP starts (this process keeps HIP's default); and first, because every hardware queue this process
    # opens later -- idle or not -- is one more for the GPU's scheduler to rotate the child's resident grids against (measured: 550 - 590 k
    # pivots/s with eight solves when run after the other legs, 770 - 930 k alone on the device).
    partitioned = None
    if args.concurrent > 1 and args.gpus == 1 and args.workload == "config3":
        try:
            import subprocess
            pr = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "gpu_concurrent.py"), "split", "4", "8", "--json"],
                                capture_output=True, text=True, timeout=300, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
            partitioned = json.loads(pr.stdout.strip().splitlines()[-1]) if pr.returncode == 0 and pr.stdout.strip() else {"error": (pr.stderr or "no output")[-300:]}
        except Exception as ex:          # the extra leg must not cost the line
            partitioned = {"error": repr(ex)[:300]}

    import torch
    import mincostflow_amd as M
    if M.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libmcf_hip.so has no CPU path")
    if world > M.device_count() and args.backend == "nccl":
        raise SystemExit(f"{world} ranks but {M.device_count()} GPU(s): one rank per GPU (rehearse more ranks on one GPU with --backend gloo)")
    dev = local_rank % M.device_count()          # one rank per GPU; ranks only share a GPU in gloo rehearsals
    dist = None
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))     # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    local_rank = dev
    numa = pin_near_gpu(dev)

    g, rule, width, desc = workload(args.workload, SEED + rank)

    def new_solver():
        ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True)
        ns.set_device(local_rank, width, 0, (M.ENGINE_DISPATCH | M.ENGINE_SAMPLE_KERNEL_TIME) if args.dispatch else 0)
        return ns.prepare()            # standard form, start basis, engine, upload: arrays resident before the clock starts

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ns = new_solver()
        assert ns.solve() == M.SolverStatus.Optimal
        del ns
    solvers = [new_solver() for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for ns in solvers:
        st = ns.solve()
        assert st == M.SolverStatus.Optimal, st
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_max = float(t.item())
    else:
        elapsed_max = elapsed
    barrier()

    mets = [ns.get_metrics() for ns in solvers]
    cost = solvers[0].get_total_cost()
    pivots = sum(m["iterations"] for m in mets)
    if dist is not None:
        t = torch.tensor([pivots], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        pivots_all = int(t.item())
    else:
        pivots_all = pivots

    sharded = None
    if args.sharded_pivots > 0:            # by default only for N > 1; `--sharded-pivots P` with one rank runs the same legs with a world of 1
        sharded = sharded_leg(M, torch, dist, args, rank, world, dev, barrier, red_dev)

    abandon = bool(sharded and sharded.pop("abandon_process_group", False))
    if rank != 0:
        if abandon:
            os._exit(0)          # a collective of the RCCL leg never completed: nothing further can be synchronised
        if dist is not None:
            dist.destroy_process_group()
        return

    e = [m["engine"] for m in mets]
    bytes_per_scan = e[0]["bytes_per_scan"]
    resident = bool(e[0]["resident"])
    if resident:
        launches = sum(x["resident_launches"] for x in e)
        requests = sum(x["resident_requests"] for x in e)
        kernel_ns = sum(x["resident_kernel_ns"] for x in e)
        bytes_per_launch = bytes_per_scan * requests / max(launches, 1)
        avg_launch_ns = kernel_ns / max(launches, 1)
        achieved = bytes_per_launch / avg_launch_ns if avg_launch_ns > 0 else 0.0       # bytes/ns == GB/s
        in_kernel_ns = sum(x["resident_scan_ns"] for x in e) / max(requests, 1)
        rule_name = {0: "FirstEligible", 1: "BestEligible", 2: "BlockSearch"}[rule]
        lds = g.node_count + 1 <= 16384
        variant = "REG, LPI (potentials in LDS)" if lds else ("REG, PIREG (potentials in registers)" if e[0]["scan_threads"] <= 512 else "REG")
        if e[0]["candidates"]:
            variant = ("REG, LPI, CAND" if lds else "REG, CAND") + " (candidate list per search)"
        kernel_name = f"resident_kernel<int{width}, {rule_name}, {variant}>"
        if e[0].get("shift_grid"):
            kernel_name = f"resident_cand_kernel<int{width}, 2 tiles> (Best Eligible + candidate list; arcs and end-point potentials in registers, patched straight from the request)"
        extra = {"launches": launches, "requests_per_launch": requests / max(launches, 1), "avg_launch_ms": avg_launch_ns / 1e6,
                 "in_kernel_phases_us": {"fetch_lines_and_set_bits": sum(x["phase_shift_ns"] for x in e) / max(requests, 1) / 1e3,
                                         "apply_values_and_states": sum(x["phase_values_ns"] for x in e) / max(requests, 1) / 1e3,
                                         "evaluate_reduce_publish": sum(x["phase_scan_ns"] for x in e) / max(requests, 1) / 1e3,
                                         "shift_lists": sum(x["shift_lists"] for x in e)} if e[0].get("shift_grid") else None,
                 "node_relabellings_per_solve": sum(x["renumberings"] for x in e) / len(e),
                 "in_kernel": {"avg_request_us": in_kernel_ns / 1e3, "achieved": bytes_per_scan / in_kernel_ns if in_kernel_ns > 0 else 0.0,
                               "frac": bytes_per_scan / in_kernel_ns / HBM_PEAK_GBS if in_kernel_ns > 0 else 0.0,
                               "what": "device clock (s_memrealtime), workgroup 0: request seen -> 16-byte record published; includes fetching and applying the patch list"}}
    else:
        timed = sum(x["timed_scans"] for x in e)
        scan_ns = sum(x["timed_scan_ns"] for x in e) / max(timed, 1)
        bytes_per_launch = bytes_per_scan
        achieved = bytes_per_scan / scan_ns if scan_ns > 0 else 0.0
        kernel_name = f"scan_kernel<int{width}, " + {0: "FirstEligible", 1: "BestEligible", 2: "BlockSearch"}[rule] + ">"
        extra = {"avg_kernel_us": scan_ns / 1e3, "timed_launches": timed}
    # the same scan as a stand-alone dispatch on the same arrays (HIP events, 200 repetitions, warm)
    it0 = solvers[0].internal()
    eng = M.PivotEngine(g.node_count + 1, it0["search_arc_num"], it0["search_arc_num"], rule=rule, optimized=True, int_width=width,
                        device=local_rank, flags=M.ENGINE_DISPATCH)
    ms = it0["search_arc_num"]
    eng.upload(it0["source"][:ms], it0["target"][:ms], it0["cost"][:ms], it0["state"][:ms], it0["pi"])
    d_avg, d_min = eng.bench_scan(reps=200)
    del eng
    extra["scan_dispatch"] = {"avg_kernel_us": d_avg / 1e3, "min_kernel_us": d_min / 1e3, "achieved": bytes_per_scan / d_avg,
                              "frac": bytes_per_scan / d_avg / HBM_PEAK_GBS,
                              "what": "scan_kernel as one dispatch per search, timed alone with HIP events; an EMPTY dispatch measures ~4 us by this method"}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)                  # separate rocprofv3 --pmc passes, see profiles/README.md
    if os.path.exists(tpath) and args.workload == "config3":
        tj = json.load(open(tpath))
        if resident:
            traffic = tj["resident_config3"]["hbm_bytes_per_request"] * requests / max(launches, 1)    # per launch, like achieved
        else:
            traffic = tj["scan_dispatch_config3"]["hbm_bytes_per_scan"]
    per = lambda k: sum(m[k] for m in mets) / max(pivots, 1)
    line = {
        "metric": "pivots/sec + solve ms, NETGEN 100k-node/300k-arc; arc-scan GB/s vs HBM peak",
        "value": pivots_all / elapsed_max,
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64" if width == 64 else "int32",
        "data": "synthetic",
        "config": {"workload": desc, "instance": f"netgen_like(seed={SEED}+rank)" if args.workload != "config4" else "assignment(seed 42)",
                   "pivot_rule": {0: "FirstEligible", 1: "BestEligible", 2: "BlockSearch"}[rule],
                   "semantics": f"EnableOptimizedPivot(true), Vector<long>.Count = {VECTOR_WIDTH} (x64; only the optimized Block Search depends on it)",
                   "search_arcs": mets[0]["search_arc_num"], "parallelism": f"{world} ranks, 1 solve per GPU" if world > 1 else "1 GPU", "host_thread_numa_node": numa},
        "solve_ms": sum(m["loop_us"] for m in mets) / len(mets) / 1e3,
        "solve_ms_incl_setup_and_upload": sum(m["total_solve_us"] for m in mets) / len(mets) / 1e3,
        "pivots_per_solve": pivots / len(mets),
        "total_cost": cost,
        "us_per_pivot": {"total": per("loop_us"), "pivot_search": per("pivot_search_us"), "tree_update": per("tree_update_us"),
                         "potential_update": per("potential_update_us")},
        "engine": {"mode": "resident grid + BAR mailbox" if resident else "one dispatch per search", "scan_workgroups": e[0]["scan_workgroups"], "inline_update_share": sum(x["inline_updates"] for x in e) / max(pivots, 1),
                   "separate_update_launches": sum(x["update_launches"] for x in e), "avg_subtree_nodes": per("potential_nodes"),
                   "candidate_cache": bool(e[0]["candidates"]), "searches": sum(x["searches"] for x in e), "device_searches": sum(x["resident_requests"] for x in e),
                   "answered_on_the_host": sum(x["host_decided"] for x in e), "lists_requested_ahead": sum(x["async_refreshes"] for x in e)},
        "roofline": dict({"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                          "traffic_source": f"profiles/{TRAFFIC_FILE} (separate rocprofv3 --pmc FETCH_SIZE pass of the same workload, not measured in this run)" if traffic is not None else None,
                          "bytes_per_launch": bytes_per_launch,
                          "note": "7.6 MB per scan lives in L2 / Infinity Cache and the per-pivot cost is host <-> device latency, not bandwidth; the grid waits "
                                  "for the host between requests, and with the candidate cache only about one search in seven is a request at all, so this "
                                  "end-to-end figure falls as the solve gets faster; scan_microbench holds the bandwidth-bound sizes"}, **extra),
    }
    if sharded is not None:
        line["sharded"] = sharded
    if args.gpus == 1 and rule == M.PivotRule.BestEligible and e[0]["candidates"]:
        # the same solve with every search sent to the device (MCF_ENGINE_NO_CANDIDATES): what the candidate cache of the headline run buys
        nc = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True).set_device(local_rank, width, 0, M.ENGINE_NO_CANDIDATES).prepare()
        assert nc.solve() == M.SolverStatus.Optimal and nc.get_total_cost() == cost
        mc = nc.get_metrics()
        line["every_search_on_the_device"] = {"flag": "MCF_ENGINE_NO_CANDIDATES", "pivots_per_s": mc["iterations"] / (mc["loop_us"] / 1e6), "solve_ms": mc["loop_us"] / 1e3,
                                              "pivots": mc["iterations"], "device_searches": mc["engine"]["resident_requests"],
                                              "us_per_pivot": {"total": mc["loop_us"] / mc["iterations"], "pivot_search": mc["pivot_search_us"] / mc["iterations"]},
                                              "in_kernel_avg_request_us": mc["engine"]["resident_scan_ns"] / max(mc["engine"]["resident_requests"], 1) / 1e3,
                                              "identical_pivot_sequence": mc["iterations"] == mets[0]["iterations"]}
        del nc
    if args.concurrent > 1 and args.gpus == 1:
        # throughput mode: several independent instances in flight on ONE GPU (one host thread, stream and resident grid each);
        # the per-pivot host <-> device latency of one solve is hidden behind the others.  Not the headline: solve latency is unchanged.
        import threading
        def shared_solver():
            ns = M.NetworkSimplex.from_problem(g).set_pivot_rule(rule).enable_optimized_pivot(True)
            return ns.set_device(local_rank, width, 0, M.ENGINE_SHARE_DEVICE).prepare()
        conc = {}
        for k_in_flight in sorted({2, 3, args.concurrent}):
            cs = [shared_solver() for _ in range(k_in_flight)]
            torch.cuda.synchronize()
            tc = time.perf_counter()
            th = [threading.Thread(target=x.solve) for x in cs]
            [t_.start() for t_ in th]
            [t_.join() for t_ in th]
            torch.cuda.synchronize()
            dt = time.perf_counter() - tc
            pv = sum(x.get_metrics()["iterations"] for x in cs)
            assert all(x.status == M.SolverStatus.Optimal and x.get_total_cost() == cost for x in cs)
            conc[str(k_in_flight)] = {"solves_in_flight": k_in_flight, "pivots_per_s": pv / dt, "seconds": dt,
                                      "solve_ms_each": sum(x.get_metrics()["loop_us"] for x in cs) / len(cs) / 1e3}
            del cs
        part = partitioned
        if isinstance(part, dict):
            for v in part.values():
                if isinstance(v, dict) and "pivots_of_each_solve" in v:
                    v["first_solve_has_the_timed_solves_pivot_count"] = v["pivots_of_each_solve"][0] == mets[0]["iterations"]      # (solver 0 has the headline instance, the others other seeds)
        line["concurrent_solves_one_gpu"] = dict(conc[str(args.concurrent)], by_solves_in_flight=conc, cus_partitioned_per_solve=part,
                                                 note="one host thread and one resident grid per solve (MCF_ENGINE_SHARE_DEVICE: the small-footprint grid); two such grids fit a CU (seven waves of 97 VGPRs each: "
                                                      "four wave slots per SIMD), the workgroups of a third wait for CUs until another solve ends -- so three or four solves in flight run two at a time; "
                                                      "a single solve is a host <-> device latency chain, so this is what the idle device buys")
    if not args.no_cpu_baseline and args.gpus == 1:
        line["cpu_baseline"] = cpu_baseline(g, rule, args.cpu_seconds)
        if rule != M.PivotRule.BlockSearch:
            blk = cpu_baseline(g, M.PivotRule.BlockSearch, args.cpu_seconds)
            line["cpu_baseline_block_search"] = blk       # EnableOptimizedPivot(true) + Block Search as an x64 host runs it: nearly every search scans every arc
            blk0 = cpu_baseline(g, M.PivotRule.BlockSearch, args.cpu_seconds, vector_width=0)
            line["cpu_baseline_block_search_without_hardware_vectors"] = blk0      # the same class where Vector.IsHardwareAccelerated is false: stops at block boundaries
            dflt = cpu_baseline(g, M.PivotRule.BlockSearch, max(args.cpu_seconds, 45.0), reference_default=True)      # long enough to finish: 1.1 M pivots
            dflt["what"] = "new NetworkSimplex(g).Solve(): plain Block Search with the reference's auto-configuration (SmallBlocksForDense / adaptive block size as its analyser picks them)"
            line["cpu_baseline_reference_default"] = dflt
            line["solve_time_vs_cpu"] = {"gpu_best_eligible_ms": line["solve_ms"],
                                         "cpu_port_best_eligible_ms_extrapolated": line["pivots_per_solve"] / line["cpu_baseline"]["value"] * 1e3,
                                         "cpu_port_block_search_optimized_x64_ms": blk["solve_ms_if_whole"] if blk["solve_ms_if_whole"] else
                                             f"not finished in the sample: {blk['us_per_pivot']:.0f} us per pivot",
                                         "cpu_port_block_search_optimized_without_hardware_vectors_ms": blk0["solve_ms_if_whole"],
                                         "cpu_port_reference_default_ms": dflt["solve_ms_if_whole"],
                                         "note": "like for like (Best Eligible) the GPU path is tens of times faster.  EnableOptimizedPivot(true) + Block Search on an x64 host IS "
                                                 "(nearly) Best Eligible: a block-boundary hit in the 'SIMD' part of the range falls through with cnt == 0 and the scan runs to the end "
                                                 "of the range (BSPO.cs:84-99), so the CPU port of that pays a full scan per pivot as well.  The same class without hardware vectors "
                                                 "(fixed block of sqrt(m) arcs, stop at the boundary) is the fast CPU rule: the GPU solve is about twice as fast as that, not ten times; "
                                                 "against what `new NetworkSimplex(g).Solve()` actually runs -- the plain Block Search whose adaptive rule shrinks the block to its "
                                                 "minimum on this instance and needs 1.1 M pivots -- it is more than ten times faster"}
    if not args.no_other_configs and args.gpus == 1 and args.workload == "config3":
        line["other_configs"] = {c: other_config(M, c, local_rank, 0 if args.no_cpu_baseline else min(args.cpu_seconds, 8.0)) for c in ("config2", "config4")}
    if not args.no_microbench and args.gpus == 1:
        line["scan_microbench"] = microbench()
        line["hbm_measured"] = hbm_probe()
        c5 = [x for x in line["scan_microbench"] if x["case"].startswith("NETGEN-like 1M nodes / 8M arcs start basis")]
        if c5:
            # the kernel of this build that IS bandwidth-bound: the RC layout's scan over config 5's arrays (the roofline above is a latency figure)
            x = c5[0]
            tr = None
            if os.path.exists(tpath):
                tr = json.load(open(tpath)).get("scan_rc_config5", {}).get("hbm_bytes_per_scan")
            line["roofline"]["bandwidth_bound_kernel"] = {
                "kernel": "scan_rc_kernel<BestEligible> over NETGEN-like 1M nodes / 8M arcs (BASELINE config 5's arrays)", "bound": "hbm",
                "bytes_per_launch": x["bytes"], "what_the_bytes_are": "9 B per arc: state + the arc's reduced cost (this layout's scan reads nothing else; SURVEY.md 8d's 17 B per arc + potentials would be " + str(x["survey_bytes"]) + ")",
                "avg_launch_us_cold": x["cold_us"], "avg_launch_us_warm": x["warm_us"], "achieved": x["cold_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": x["cold_frac_of_hbm_peak"], "frac_warm": x["warm_frac_of_hbm_peak"], "traffic": tr,
                "traffic_source": f"profiles/{TRAFFIC_FILE} (separate rocprofv3 --pmc FETCH_SIZE pass)" if tr else None}
    if not args.no_microbench and args.gpus == 1:
        line["potential_update_microbench"] = potential_update_microbench(M, g, local_rank)
        line["large_instance_sample"] = large_instance_sample(M, local_rank, not args.no_cpu_baseline)
    if not args.no_validator and args.gpus == 1:
        line["solution_validator"] = validator_bench(M, g, solvers[0], local_rank, not args.no_cpu_baseline)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())
    if abandon:
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
