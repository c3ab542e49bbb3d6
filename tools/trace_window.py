#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel trace: kernel, duration, gap since the previous kernel ended.
  python3 tools/trace_window.py <rocprof output dir> [first_from_end=60] [count=30]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
a = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sel = rows[len(rows) - a: len(rows) - a + n]
prev = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-58s %8.1f us   gap %6.1f us" % (r["Kernel_Name"].replace("void ", "")[:58], (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0))
    prev = e
