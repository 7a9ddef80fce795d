#!/usr/bin/env python3
"""Copies the summaries `bash profiles/r03_evidence.sh` left under gpurun_out/r03/ into profiles/ (each with the commit they were
taken at in its first line) and rewrites the round-3 figures of profiles/traffic.json from the PMC tables.
    python profiles/r03_collect.py <commit> [gpurun_out/r03]"""
import json
import os
import re
import shutil
import sys

commit = sys.argv[1][:12]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r03"
here = os.path.dirname(os.path.abspath(__file__))
header = "<!-- rocprofv3, profiles/r03_evidence.sh at commit %s (round 3) -->\n" % commit


def table(path):
    rows = {}
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) == 5 and cells[0] in ("FETCH_SIZE", "WRITE_SIZE"):
            rows.setdefault(cells[0], []).append((cells[1], int(cells[2]), int(cells[3]), float(cells[4])))
    return rows


for name in sorted(os.listdir(src)):
    m = re.match(r"ev_(.*)\.md$", name)
    if m:
        with open(os.path.join(here, "r03_" + m.group(1) + ".md"), "w") as out:
            out.write(header)
            out.write(open(os.path.join(src, name)).read())
if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(here, "r03_bench_default.json"))

tj_path = os.path.join(here, "traffic.json")
tj = json.load(open(tj_path))
KIB = 1024


def mean_of(rows, counter, kernel):
    hit = [r for r in rows.get(counter, []) if r[0].startswith(kernel)]
    return max(hit, key=lambda r: r[1] * r[2])[3] if hit else 0.0        # the benchmark's launches, not the oracle check's small one


slab = table(os.path.join(src, "ev_slab_pipeline_pmc.md"))
per = 4.0                                                                # one pass = 2^22 samples, figures per 2^20
redo = [r for r in slab.get("FETCH_SIZE", []) if r[0].startswith("slab_redo_kernel")]
passes = [r for r in slab.get("FETCH_SIZE", []) if r[0].startswith("slab_compact_kernel")]
redo_share = (redo[0][2] / passes[0][2]) if redo and passes else 0.0      # redo runs for one of the two components
raw = {
    "compact_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_compact_kernel") * KIB * 2 / per),
    "compact_write": round(mean_of(slab, "WRITE_SIZE", "slab_compact_kernel") * KIB / per),
    "gather_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_gather_fast_kernel") * KIB * 2 / per),
    "gather_write": round(mean_of(slab, "WRITE_SIZE", "slab_gather_fast_kernel") * KIB / per),
    "combine_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_combine_kernel") * KIB * 2 / per),
    "redo_fetch_avg_of_two_components": round(mean_of(slab, "FETCH_SIZE", "slab_redo_kernel") * KIB * 2 / per * redo_share),
}
tj["slab_pipeline_raw_r03_per_2^20"] = raw
tj["slab_pipeline_bytes_per_launch"] = sum(raw.values())
dense = table(os.path.join(src, "ev_dense_pmc.md"))
fetch, write = mean_of(dense, "FETCH_SIZE", "syndrome_tiled_kernel"), mean_of(dense, "WRITE_SIZE", "syndrome_tiled_kernel")
tj["syndrome_tiled_kernel_raw_r03"]["FETCH_SIZE_KiB"] = fetch
tj["syndrome_tiled_kernel_raw_r03"]["WRITE_SIZE_KiB"] = write
tj["syndrome_tiled_kernel_bytes_per_launch"] = round(fetch * KIB * 2 + write * KIB)
for shape, calls in (("2048x4096x256", 4), ("32768x65536x1", 2)):
    rows = table(os.path.join(src, "ev_rref_%s_pmc.md" % shape))
    total = sum(v * n * KIB * 2 for _, _, n, v in rows.get("FETCH_SIZE", [])) + sum(v * n * KIB for _, _, n, v in rows.get("WRITE_SIZE", []))
    key = shape.rsplit("x", 1)[0] + "_x" + shape.rsplit("x", 1)[1]
    tj["rref_bytes_per_call"][key] = round(total / calls)
tj["captured_at_commit"] = commit
tj["_how_r03"] = re.sub(r"at commit [0-9a-f]+", "at commit " + commit, tj["_how_r03"])
json.dump(tj, open(tj_path, "w"), indent=1)
print("slab pipeline %d bytes per 2^20 samples, dense %d, rref %s" % (tj["slab_pipeline_bytes_per_launch"],
      tj["syndrome_tiled_kernel_bytes_per_launch"], {k: v for k, v in tj["rref_bytes_per_call"].items() if not k.startswith("_")}))
