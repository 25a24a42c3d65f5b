#!/usr/bin/env python3
"""Kernel sequence (start, duration, gap to the previous kernel) of one headline step from a rocprofv3 kernel trace of
bench.py:   python tools/step_trace.py <kernel_trace.csv> [nth project_fwd from the end, default 8]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "project_fwd_kernel" in r["Kernel_Name"]]
want = int(sys.argv[2]) if len(sys.argv) > 2 else None
cands = range(len(idx) - 1) if want is None else [len(idx) - 1 - want]
for n in cands:
    a, b = idx[n], idx[n + 1]
    names = [r["Kernel_Name"] for r in rows[a:b]]
    if want is not None or (n > 10 and any("l1_fwd" in x for x in names) and any("adam1" in x for x in names)):
        break
t0, prev = int(rows[a]["Start_Timestamp"]), None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r["Kernel_Name"][:100]))
    prev = e
print("step span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
