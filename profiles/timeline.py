#!/usr/bin/env python3
"""Prints a window of a rocprofv3 --kernel-trace CSV as a timeline: start (us from the window's first kernel), duration, queue,
grid, kernel.   python profiles/timeline.py <kernel_trace.csv> <name filter regex> <first match to show> <rows>"""
import csv, re, sys
path, pat, first, rows = sys.argv[1], re.compile(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
recs = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(recs) if pat.search(r["Kernel_Name"])]
lo = idx[first] if len(idx) > first else 0
t0 = int(recs[lo]["Start_Timestamp"])
for r in recs[lo:lo + rows]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    print("%9.1f us  %7.1f us  q%s  %6d wg  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], wg, name))
