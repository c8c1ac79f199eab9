#!/usr/bin/env python
"""Sum the counters of a `rocprofv3 --pmc ... --output-format csv` run per kernel (template arguments kept, parameter list cut).
    python tools/pmc_sum.py <dir with *counter_collection.csv> [substring of the kernel name]"""
import csv, glob, os, sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("stito::", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
names = sorted({c for v in rows.values() for c in v})
print(f"{'kernel':44s} calls " + " ".join(f"{n:>24s}" for n in names))
for k, v in sorted(rows.items()):
    print(f"{k[:44]:44s} {len(calls[k]):5d} " + " ".join(f"{v.get(n, 0.0):24.4g}" for n in names))
