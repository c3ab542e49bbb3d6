#!/usr/bin/env python3
"""Print the tail of a rocprofv3 kernel trace as a timeline with the queue of every kernel (several streams):
  python3 tools/trace_timeline.py <rocprof output dir> [last_n=90]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 90
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f -> %9.1f us  q%-3s %-44s %8.1f us" % ((s - t0) / 1e3, (e - t0) / 1e3, r.get("Queue_Id", "?"),
                                                      r["Kernel_Name"].replace("void ", "").replace("s2d::", "")[:44], (e - s) / 1e3))
