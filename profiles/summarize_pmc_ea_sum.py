#!/usr/bin/env python
"""EA-side read counters of one `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum --kernel-trace`
pass, SUMMED per kernel name over the dispatches behind the warm-up (a chunked trunk schedule launches a layer once per chunk):
launches, ms, EA read GB, average EA read latency in L2 clocks (LEVEL / RDREQ: ~ 640 - 700 = an Infinity-Cache-resident stream,
1 230 - 1 650 = an HBM one at these rates, profiles/round5_mall_curve.txt).

    python profiles/summarize_pmc_ea_sum.py a.db [skip_dispatches_per_kernel_fraction]"""
import sqlite3
import sys
from collections import OrderedDict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events order by dispatch_id").fetchall()
d = OrderedDict()
for did, name, cn, cv, dur in rows:
    e = d.setdefault(did, {"name": name, "dur": dur, "c": {}})
    e["c"][cn] = e["c"].get(cn, 0.0) + cv
disp = [e for e in d.values() if "k_conv" in e["name"]]
heads = [i for i, e in enumerate(d.values()) if "k_head" in e["name"]]
# the last trunk pass only: dispatches between the last two k_head launches
allv = list(d.values())
lo = heads[-2] + 1 if len(heads) > 1 else 0
last = [e for e in allv[lo:heads[-1]] if "k_conv" in e["name"]]
agg = OrderedDict()
for e in last:
    n = e["name"].replace("void ", "").replace("stito::", "").split("(")[0][:60]
    a = agg.setdefault(n, {"n": 0, "ms": 0.0, "rd": 0.0, "lvl": 0.0, "r32": 0.0})
    a["n"] += 1; a["ms"] += e["dur"] / 1e6
    a["rd"] += e["c"].get("TCC_EA0_RDREQ_sum", 0.0); a["lvl"] += e["c"].get("TCC_EA0_RDREQ_LEVEL_sum", 0.0); a["r32"] += e["c"].get("TCC_EA0_RDREQ_32B_sum", 0.0)
print(f"{'kernel':60s} {'launches':>8s} {'ms':>8s} {'EA rd GB':>9s} {'TB/s':>6s} {'latency clk':>11s}")
tot = {"ms": 0.0, "gb": 0.0}
for n, a in agg.items():
    gb = (32 * a["r32"] + 64 * (a["rd"] - a["r32"])) / 1e9
    tot["ms"] += a["ms"]; tot["gb"] += gb
    print(f"{n:60s} {a['n']:8d} {a['ms']:8.3f} {gb:9.3f} {gb / max(a['ms'], 1e-9):6.2f} {a['lvl'] / max(a['rd'], 1.0):11.0f}")
print(f"{'all conv launches of the pass':60s} {'':8s} {tot['ms']:8.3f} {tot['gb']:9.3f}")
