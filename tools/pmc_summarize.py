"""Summarises a rocprofv3 --pmc counter csv: per kernel (short name) the number of dispatches and the average / min / max counter value."""
import csv, re, sys, collections
rows = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        rows[(r["Counter_Name"], name)].append(float(r["Counter_Value"]))
for (ctr, name), v in sorted(rows.items()):
    print(f"{ctr:12s} {name:70s} n={len(v):6d} avg={sum(v)/len(v):14.2f} min={min(v):14.2f} max={max(v):14.2f}")
