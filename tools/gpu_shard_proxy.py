"""What ONE of eight GPUs does per pivot of config 5: an engine holding an eighth of the search arcs (windows of reduced costs in LDS), whole searches
back to back (mcf_engine_bench_search); the same for the whole instance on one GPU and for the gathering layouts, for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
ns0 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
assert ns0.begin() == 0
it0 = ns0.internal(); ms = it0["search_arc_num"]
def one(label, shard, flags=0, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    eng = M.PivotEngine(g5.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, shard=shard, flags=flags)
    for k in (env or {}):
        os.environ.pop(k, None)
    eng.upload(it0["source"][:ms], it0["target"][:ms], it0["cost"][:ms], it0["state"][:ms], it0["pi"])
    avg, mn = eng.bench_search(3000)
    st = eng.stats()
    print(f"{label:62s} arcs {shard[1]-shard[0] if shard[1] else ms:8d} resident {st['resident']} rc {st['rc_layout']} grid {st['scan_workgroups']:4d}x{st['scan_threads']:4d} | "
          f"{avg/1e3:6.2f} us per search (min {mn/1e3:5.2f}) | in-kernel {st['resident_scan_ns']/max(st['resident_requests'],1)/1e3:5.2f} us", flush=True)
    del eng
for world in (8, 4, 2):
    one(f"shard 0 of {world}, resident RC grid", M.shard_range(ms, 0, world))
one("whole instance, resident RC grid (streamed)", (0, 0))
one("whole instance, RC layout, one dispatch per search", (0, 0), M.ENGINE_DISPATCH)
one("shard 0 of 8, RC layout, one dispatch per search", M.shard_range(ms, 0, 8), M.ENGINE_DISPATCH)
one("shard 0 of 8, gathering layout, one dispatch per search (round 1)", M.shard_range(ms, 0, 8), M.ENGINE_DISPATCH, {"MCF_HIP_RC": 0})
