#!/usr/bin/env python
"""Per-dispatch list of a rocprofv3 --kernel-trace run (rocpd sqlite .db): name, duration, grid -- in launch order.
Usage: python tools/kernel_trace_dump.py <results.db> [substring]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for name, start, end in cur.execute("select name, start, end from kernels order by start"):
    if sub in name:
        print(f"{(end - start) / 1e3:10.1f} us  {name[:150]}")
