#!/usr/bin/env python3
"""Mean PMC counter value per (kernel, grid) from rocprofv3 --pmc counter_collection.csv files.
    python profiles/pmc_summary.py gpurun_out/pmc*/**/*_counter_collection.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    for rec in csv.DictReader(open(path)):
        name = rec["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(rec["Counter_Name"], name, int(rec["Grid_Size"]))].append(float(rec["Counter_Value"]))
print("| counter | kernel | grid (work-items) | dispatches | mean value |")
print("|---|---|---|---|---|")
for (counter, name, grid), vals in sorted(agg.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
    if sum(vals) / len(vals) >= 1000:
        print("| %s | %s | %d | %d | %.1f |" % (counter, name, grid, len(vals), sum(vals) / len(vals)))
