import os, sys
sys.path.insert(0, "/root/repo")
import mincostflow_amd as M
g5 = M.netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
ns0 = M.NetworkSimplex.from_problem(g5).set_pivot_rule(M.PivotRule.BestEligible)
assert ns0.begin() == 0
it0 = ns0.internal(); ms = it0["search_arc_num"]
def one(label, shard, flags=0):
    eng = M.PivotEngine(g5.node_count + 1, ms, ms, rule=M.PivotRule.BestEligible, int_width=64, shard=shard, flags=flags)
    eng.upload(it0["source"][:ms], it0["target"][:ms], it0["cost"][:ms], it0["state"][:ms], it0["pi"])
    avg, mn = eng.bench_search(3000)
    st = eng.stats()
    print(f"{label:50s} cand {st['candidates']} resident {st['resident']} grid {st['scan_workgroups']} | {avg/1e3:6.2f} us per search (min {mn/1e3:5.2f}) | in-kernel {st['resident_scan_ns']/max(st['resident_requests'],1)/1e3:5.2f} us | requests {st['resident_requests']}", flush=True)
for world in (8, 2):
    one(f"shard 0 of {world}, cache", M.shard_range(ms, 0, world))
    one(f"shard 0 of {world}, no cache", M.shard_range(ms, 0, world), M.ENGINE_NO_CANDIDATES)
one("whole, cache", (0, 0))
one("whole, no cache", (0, 0), M.ENGINE_NO_CANDIDATES)
