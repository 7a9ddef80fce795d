#!/usr/bin/env python3
"""Timeline of the last gf2_rref_batch_dev call in a rocprofv3 kernel trace: per kernel its queue, start and duration (us)
relative to the call's first kernel.   python profiles/r05_timeline.py <kernel_trace.csv> [max rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last state-init kernel
starts = [i for i, r in enumerate(rows) if "state_init" in r["Kernel_Name"]]
first = starts[-1] - 1 if starts else 0
t0 = int(rows[first]["Start_Timestamp"])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80
busy = []
for r in rows[first:first + limit]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("rref_sweep_", "")[:32]
    print("q%-2s %-32s start %8.1f  dur %7.1f  end %8.1f  grid %s" % (r["Queue_Id"], name, s, e - s, e, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
end = max(int(r["End_Timestamp"]) for r in rows[first:])
print("call: %.1f us from first start to last end" % ((end - t0) / 1e3))
