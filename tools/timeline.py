#!/usr/bin/env python3
"""Print a merged device timeline (kernels + memory copies) of the LAST `window_ms` milliseconds of a rocprofv3 run made with
`--kernel-trace --memory-copy-trace --output-format csv`:  python3 tools/timeline.py <dir> [window_ms]"""
import csv
import glob
import sys

d, win = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s" % r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r.get("Direction", "")[:30]))
ev.sort()
t_end = max(e[1] for e in ev)
t0 = t_end - int(win * 1e6)
prev = None
for s, e, kind, name in ev:
    if s < t0:
        continue
    gap = "" if prev is None else ("  (+%.0f us idle)" % ((s - prev) / 1e3) if s - prev > 20000 else "")
    print("%9.1f us  %7.1f us  %-8s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, kind, name, gap))
    prev = max(prev or 0, e)
