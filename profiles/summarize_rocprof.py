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
    # The conv launches as bench.py counts them: the MFMA 3x3-conv launches of every full-size trunk pass (10 with conv_block1 in one launch).  A layer is one
    # convolution kernel (> 1 ms at full size: k_conv_wino43 MODE 0 / 1, k_conv_wino43s / s2 / h, k_conv_wino23r, k_conv_wino8, k_conv3x3) plus
    # -- where the F(4x4,3x3) input transform is hoisted -- its transform pass (template MODE 2 / 3 / 4 of k_conv_wino43,
    # > 0.08 ms at full size; the 2-stream target-embedding pass is far below both thresholds).
    import re
    fam = {"f32": [0, 0.0], "f16-stream": [0, 0.0], "f16-reg": [0, 0.0]}   # bench.py's roofline families
    n_tr, t_tr = 0, 0.0
    # (round 6: a k_conv_wino43s layer runs as several grids on several queues -- dispatches of the same instantiation that OVERLAP in time
    # are one layer, from the first one's start to the last one's end)
    merged = []
    for name, t0, t1 in cur.execute("select name, start, end from kernels where name like '%k_conv_wino%' or name like '%k_conv3x3%' order by start"):
        if merged and merged[-1][0] == name and t0 < merged[-1][2] and "k_conv_wino43s<" in name:
            merged[-1][2] = max(merged[-1][2], t1)
        else:
            merged.append([name, t0, t1])
    for name, t0, t1 in merged:
        dur = t1 - t0
        m = re.search(r"k_conv_wino43<([^>]*)>", name)
        mode = int(m.group(1).split(",")[-1]) if m and len(m.group(1).split(",")) >= 4 else 0
        if mode in (2, 3, 4, 5):
            if 8e4 < dur:
                n_tr += 1; t_tr += dur / 1e6
                fam["f16-stream" if mode in (3, 4, 5) else "f32"][1] += dur / 1e6
        elif dur > 1e6:
            pipe = "f16-stream" if re.search(r"k_conv_wino43(s3|s2|s|h)<", name) else ("f16-reg" if "k_conv_wino23r" in name else "f32")
            fam[pipe][0] += 1; fam[pipe][1] += dur / 1e6
    n = sum(v[0] for v in fam.values())
    if n:
        total = sum(v[1] for v in fam.values())
        print(f"# conv launches: {n} conv layers ({n_tr} of them with a separate transform pass, {t_tr:.2f} ms in those passes), avg {total / n:.4f} ms per layer, "
              f"total {total:.2f} ms")
        for pipe in fam:
            if fam[pipe][0]:
                print(f"#   {pipe} matrix pipe: {fam[pipe][0]} layers, avg {fam[pipe][1] / fam[pipe][0]:.4f} ms per layer   <- compare with roofline*.avg_launch_ms of the bench line")
