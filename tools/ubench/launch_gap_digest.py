#!/usr/bin/env python3
import csv, re, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
gaps, durs = {}, {}
for a, b in zip(rows, rows[1:]):
    if "producer" in a["Kernel_Name"] and "consumer" in b["Kernel_Name"]:
        tag = re.search(r"consumer<(\d+)>", b["Kernel_Name"]).group(1)
        gaps.setdefault(tag, []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
        durs.setdefault(tag, []).append((int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3)
for tag in gaps:
    print("consumer grid tag %6s: gap after the producer %6.2f us (median), consumer runs %7.2f us" % (tag, statistics.median(gaps[tag]), statistics.median(durs[tag])))
