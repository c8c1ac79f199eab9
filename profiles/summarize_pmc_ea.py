#!/usr/bin/env python
"""Per-launch table of arbitrary PMC counters for the conv launches of the bench process's LAST trunk pass (or, with
`--all <substring>`, of every dispatch whose kernel name contains the substring), from one or more `rocprofv3 --pmc ...
--kernel-trace` runs (rocpd databases; one pass per database, dispatches aligned by order).

Used for the DRAM-vs-fabric attribution of the streaming convolutions: the TCC's memory-side counters
TCC_EA0_RDREQ (requests, 32 / 64 / 128 B), TCC_EA0_RDREQ_32B, TCC_EA0_RDREQ_DRAM (requests routed to the memory controller
rather than GMI / IO), TCC_EA0_RDREQ_LEVEL (requests in flight summed per cycle: LEVEL / RDREQ = average latency of an EA
read in L2 clocks) and the L2's own TCC_HIT / TCC_MISS / TCC_REQ.  The Infinity Cache sits behind the EA interface: its
hits can only show as a LOWER average EA latency (calibrated by tools/ubench/mall_probe.hip under the same counters).

    python profiles/summarize_pmc_ea.py a.db [b.db ...]            # conv launches of the last trunk pass
    python profiles/summarize_pmc_ea.py --all k_probe a.db [b.db]   # every dispatch of a kernel"""
import sqlite3
import sys
from collections import OrderedDict

args = sys.argv[1:]
sub = None
if args and args[0] == "--all":
    sub, args = args[1], args[2:]


def load(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events order by dispatch_id").fetchall()
    d = OrderedDict()
    for did, name, cn, cv, dur in rows:
        e = d.setdefault(did, {"name": name, "dur": dur, "c": {}})
        e["c"][cn] = e["c"].get(cn, 0.0) + cv
    return list(d.values())


def short(name):
    n = name.replace("void ", "").replace("stito::", "").split("(")[0]
    return n[:46]


def pick(rows):
    if sub is not None:
        return [r for r in rows if sub in r["name"]]
    heads = [i for i, r in enumerate(rows) if "k_head" in r["name"]]
    assert heads, "no trunk pass in the trace"
    lo = heads[-2] + 1 if len(heads) > 1 else 0
    return [r for r in rows[lo:heads[-1]] if "k_conv" in r["name"]]


passes = [pick(load(p)) for p in args]
n = len(passes[0])
assert all(len(p) == n and [r["name"] for r in p] == [r["name"] for r in passes[0]] for p in passes), [len(p) for p in passes]
names = []
for p in passes:
    for cn in p[0]["c"]:
        if cn not in names:
            names.append(cn)
hdr = f"{'kernel':46s} {'ms':>7s} " + " ".join(f"{c.replace('TCC_', '').replace('_sum', ''):>18s}" for c in names)
derived = []
if "TCC_EA0_RDREQ_LEVEL_sum" in names and "TCC_EA0_RDREQ_sum" in names:
    derived.append("ea_rd_latency_clk")
if "TCC_EA0_RDREQ_32B_sum" in names and "TCC_EA0_RDREQ_sum" in names:
    derived.append("ea_rd_GB")
if "TCC_EA0_RDREQ_DRAM_sum" in names and "TCC_EA0_RDREQ_sum" in names:
    derived.append("dram_routed_frac")
if "TCC_HIT_sum" in names and "TCC_MISS_sum" in names:
    derived.append("l2_hit_rate")
print(hdr + " " + " ".join(f"{d:>18s}" for d in derived))
for i in range(n):
    c = {}
    for p in passes:
        c.update(p[i]["c"])
    ms = sum(p[i]["dur"] for p in passes) / len(passes) / 1e6
    line = f"{short(passes[0][i]['name']):46s} {ms:7.3f} " + " ".join(f"{c.get(k, 0.0):18.5g}" for k in names)
    for d in derived:
        if d == "ea_rd_latency_clk":
            v = c["TCC_EA0_RDREQ_LEVEL_sum"] / max(c["TCC_EA0_RDREQ_sum"], 1.0)
        elif d == "ea_rd_GB":
            # RDREQ counts 32-B, 64-B and 128-B requests alike; _32B counts the 32-B ones.  Bounds: all others 64 B / all others 128 B
            r, r32 = c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_32B_sum"]
            v = (32 * r32 + 64 * (r - r32)) / 1e9
        elif d == "dram_routed_frac":
            v = c["TCC_EA0_RDREQ_DRAM_sum"] / max(c["TCC_EA0_RDREQ_sum"], 1.0)
        else:
            v = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
        line += f" {v:18.4f}"
    print(line)
