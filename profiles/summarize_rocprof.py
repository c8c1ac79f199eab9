#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) per kernel name.
Usage: python profiles/summarize_rocprof.py gpurun_out/prof_x/*/*_results.db > profiles/xxx.txt"""
import sqlite3
import sys

for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {path}\n# total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:10.3f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {100 * r[2] / tot:6.2f}")
    # The roofline kernel family as bench.py counts it: the 11 MFMA 3x3-conv layers of every full-size trunk pass.  A layer is
    # one k_conv_wino* launch (> 1 ms at full size), plus -- where the F(4x4,3x3) input transform is hoisted -- its transform
    # pass (template MODE 2, > 0.08 ms at full size; the 2-stream target-embedding pass is far below both thresholds).
    n, total = cur.execute("select count(*), sum(end-start)/1e6 from kernels where name like '%k_conv_wino%' and (end-start) > 1e6").fetchone()
    n2, total2 = cur.execute("select count(*), sum(end-start)/1e6 from kernels where name like '%k_conv_wino43%false, 2>%' "
                             "and (end-start) > 8e4 and (end-start) <= 1e6").fetchone()
    if n:
        total += total2 or 0.0
        print(f"# MFMA conv family (k_conv_wino*): {n} conv layers ({n2} of them with a separate transform pass), avg {total / n:.4f} ms per layer, "
              f"total {total:.2f} ms   <- compare with roofline.avg_launch_ms of the bench line")
