#!/usr/bin/env python
"""Per-dispatch timeline of the LAST bench step in a rocprofv3 --kernel-trace run (rocpd .db):
queue, start offset, duration and kernel name -- shows whether the render and embed streams overlap.
    python tools/step_timeline.py gpurun_out/prof_x/*/*_results.db [n_last_dispatches]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = cur.execute("select queue_id, start, end, name, grid_x, workgroup_x from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
busy_end = t0
idle = 0.0
for q, s, e, name, gx, wx in rows:
    short = name.split("(")[0].replace("void ", "").replace("stito::", "")
    gap = max(0, s - busy_end)   # nothing was running on any queue for this long before the dispatch started
    idle += gap
    busy_end = max(busy_end, e)
    print(f"q{q:<3d} +{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:10.1f} us  idle before {gap / 1e3:7.1f} us  blocks {gx // max(wx, 1):7d}  {short[:60]}")
print(f"span {(max(r[2] for r in rows) - t0) / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms")
