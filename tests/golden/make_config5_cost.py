"""Generates tests/golden/config5_cost.json: the optimal cost of BASELINE.json configs[4] -- netgen_like(13502460, 1_000_000, 8_000_000, 1000, 1000)
-- as solved by the CPU oracle (oracle/ns_oracle.c, C#-optimized semantics, Block Search).  The oracle needs 15-20 minutes of one core for
this instance, too long for the GPU test run, so its answer is kept as a golden value; the GPU test compares the HIP path's Best-Eligible
optimum with it and certifies that optimum independently with the device validator (primal = dual, complementary slackness).

    python tests/golden/make_config5_cost.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import mincostflow_amd as M          # the generator lives in the product library (the reference ships none: SURVEY.md F6)
from oracle import ns_oracle as O

ARGS = (13502460, 1_000_000, 8_000_000, 1000, 1000)
g = M.netgen_like(*ARGS)
p = O.Problem(g.node_count, g.arc_count, g.source, g.target, g.lower, g.upper, g.cost, g.supply)
o = O.Oracle(p, O.SEM_CSHARP_OPT, O.RULE_BLOCK)
t0 = time.time()
status, _ = o.solve()
out = {"instance": "netgen_like" + repr(ARGS), "oracle": "SEM_CSHARP_OPT, Block Search", "status": status, "total_cost": int(o.total_cost),
       "pivots": int(o.pivots), "search_arc_num": int(o.search_arc_num), "oracle_seconds": round(time.time() - t0, 1),
       "checksum_source": int(g.source.astype("int64").sum()), "checksum_cost": int(g.cost.sum()), "checksum_supply_abs": int(abs(g.supply).sum())}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config5_cost.json"), "w"), indent=1)
print(out)
