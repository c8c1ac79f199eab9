#!/usr/bin/env python
"""HBM-side traffic of the NON-trunk kernels of an evaluate step (effect-chain render + log-mel) from two rocprofv3 --pmc
passes over the bench process (FETCH_SIZE and WRITE_SIZE separately, --kernel-trace only, STITO_GRAPH=0 so that every
kernel is a host-side dispatch the counters are attributed to):

    STITO_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline
    STITO_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d w -- python bench.py ... (same)
    python profiles/summarize_pmc_dsp.py f/*/*_results.db w/*/*_results.db <pop> <n_samples> [out.json]

Per kernel: launches per step, device time per step, FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md section HBM:
128-byte requests are tallied at 64 B; Infinity-Cache hits are inside: L2-miss traffic) and WRITE_SIZE per step.  A step = the
dispatches between two consecutive k_head launches (one trunk pass per step at pop 256); the LAST step of the process is used.
Algorithmic bytes, per candidate: shared input read, rendered audio written, log-mel written: P (4 C L + 4 C L + 4 C T M)."""
import json
import os
import sqlite3
import sys
from collections import OrderedDict

fetch_db, write_db, P, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
DSP = ("k_prepare", "k_eq", "k_comp_", "k_reverb", "k_pointwise", "k_delay", "k_peak", "k_normalize", "k_logmel", "k_chorus", "k_cr_", "k_zero_words")


def rows(path, counter):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select name, counter_value, duration from pmc_events where counter_name=? order by dispatch_id", (counter,)).fetchall()


def last_step(r):
    heads = [i for i, x in enumerate(r) if "k_head" in x[0]]
    assert len(heads) >= 2, "need at least two trunk passes in the trace"
    return r[heads[-2] + 1:heads[-1]]


def short(name):
    return name.replace("void ", "").replace("stito::", "").split("(")[0]


f, w = last_step(rows(fetch_db, "FETCH_SIZE")), last_step(rows(write_db, "WRITE_SIZE"))
assert [short(a[0]) for a in f] == [short(b[0]) for b in w], "the two passes dispatched different kernels"
tab = OrderedDict()
for (name, fv, fd), (_, wv, wd) in zip(f, w):
    k = short(name)
    if not any(k.startswith(d) for d in DSP):
        continue
    e = tab.setdefault(k, dict(n=0, ms=0.0, fetch=0.0, write=0.0))
    e["n"] += 1
    e["ms"] += 0.5 * (fd + wd) / 1e6
    e["fetch"] += fv * 1024 * 2 / 1e9
    e["write"] += wv * 1024 / 1e9
C = 2
T = n // 1024 + 1
# SURVEY 8(d), per candidate: the shared input read (4 C L), the rendered audio written (4 C L), the log-mel written (4 C T M)
alg = P * (4.0 * C * n + 4.0 * C * n + 4.0 * C * T * 128) / 1e9
print(f"{'kernel':40s} {'launches':>8s} {'ms/step':>9s} {'fetch_GB(x2)':>13s} {'write_GB':>9s}")
tf = tw = tm = 0.0
for k, e in tab.items():
    print(f"{k[:40]:40s} {e['n']:8d} {e['ms']:9.3f} {e['fetch']:13.3f} {e['write']:9.3f}")
    tf += e["fetch"]; tw += e["write"]; tm += e["ms"]
print(f"{'total (render + log-mel), one step':40s} {'':8s} {tm:9.3f} {tf:13.3f} {tw:9.3f}   traffic {tf + tw:.3f} GB vs algorithmic {alg:.3f} GB = {(tf + tw) / alg:.2f} x")
if len(sys.argv) > 5:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import dsp_source_hash
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, STITO_GRAPH=0) on the bench process: "
                           "the effect-chain and log-mel kernels of its last step.  FETCH_SIZE doubled (gfx950), Infinity-Cache hits included.",
               "dsp_source_hash": dsp_source_hash(), "pop": P, "n_samples": n, "fetch_GB_x2": tf, "write_GB": tw, "traffic_bytes_per_step": (tf + tw) * 1e9,
               "algorithmic_bytes_per_step": alg * 1e9, "kernel_ms_sum": tm,
               "kernels": {k: {"launches": e["n"], "ms": round(e["ms"], 4), "fetch_GB_x2": round(e["fetch"], 4), "write_GB": round(e["write"], 4)} for k, e in tab.items()}},
              open(sys.argv[5], "w"), indent=1)
