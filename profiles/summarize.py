#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, mean/min/max duration.
    python profiles/summarize.py gpurun_out/prof/.../*_kernel_trace.csv > profiles/rNN_name.md"""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
for path in sys.argv[1:]:
    with open(path) as fh:
        for rec in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", rec["Kernel_Name"]).replace("void ", "")
            # workgroups = product over the three dimensions of grid size (work-items) / workgroup size
            wgs = 1
            for d in "XYZ":
                wgs *= -(-int(rec.get("Grid_Size_" + d) or 1) // max(1, int(rec.get("Workgroup_Size_" + d) or 1)))
            key = (name, wgs, int(rec["Workgroup_Size_X"]) * int(rec.get("Workgroup_Size_Y") or 1) * int(rec.get("Workgroup_Size_Z") or 1),
                   int(rec["VGPR_Count"]), int(rec["SGPR_Count"]), int(rec["LDS_Block_Size"]))
            rows[key].append(int(rec["End_Timestamp"]) - int(rec["Start_Timestamp"]))
print("| kernel | workgroups | wg size | VGPR | SGPR | LDS B | calls | mean us | min us | max us | total ms |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("| %s | %d | %d | %d | %d | %d | %d | %.1f | %.1f | %.1f | %.2f |" % (
        key + (len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(d) / 1e6)))
