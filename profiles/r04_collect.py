#!/usr/bin/env python3
"""Copies the summaries `bash profiles/r04_evidence.sh` left under gpurun_out/r04/ into profiles/ (each with the commit they were
taken at in its first line) and rewrites the round-3 figures of profiles/traffic.json from the PMC tables.
    python profiles/r04_collect.py <commit> [gpurun_out/r04]"""
import json
import os
import re
import shutil
import sys

commit = sys.argv[1][:12]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r04"
here = os.path.dirname(os.path.abspath(__file__))
header = "<!-- rocprofv3, profiles/r04_evidence.sh at commit %s (round 4) -->\n" % commit


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
        with open(os.path.join(here, "r04_" + m.group(1) + ".md"), "w") as out:
            out.write(header)
            out.write(open(os.path.join(src, name)).read())
if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(here, "r04_bench_default.json"))

tj_path = os.path.join(here, "traffic.json")
tj = json.load(open(tj_path))
KIB = 1024


def mean_of(rows, counter, kernel):
    hit = [r for r in rows.get(counter, []) if r[0].startswith(kernel)]
    return max(hit, key=lambda r: r[1] * r[2])[3] if hit else 0.0        # the benchmark's launches, not the oracle check's small one


slab = table(os.path.join(src, "ev_slab_pipeline_pmc.md"))
per = 4.0                                                                # one pass = 2^22 samples, figures per 2^20
# Round 4: the combine step of a pass rides in the next pass' gather launch (its 8 B of partial weights per sample are in the gather
# kernel's fetch); only a call's last pass has a combine launch: one per 32 passes in the default run of 2^27 samples per call.  The
# redo pass is gone for the benchmark's H2 (its left-out column is zero).
raw = {
    "compact_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_compact_kernel") * KIB * 2 / per),
    "compact_write": round(mean_of(slab, "WRITE_SIZE", "slab_compact_kernel") * KIB / per),
    "gather_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_gather_fast_kernel") * KIB * 2 / per),
    "gather_write": round(mean_of(slab, "WRITE_SIZE", "slab_gather_fast_kernel") * KIB / per),
    "combine_fetch_one_launch_per_32_passes": round(mean_of(slab, "FETCH_SIZE", "slab_combine_kernel") * KIB * 2 / per / 32),
    "redo_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_redo_kernel") * KIB * 2 / per),
}
tj["slab_pipeline_raw_r04_per_2^20"] = raw
tj["slab_pipeline_bytes_per_launch"] = sum(raw.values())
stored = table(os.path.join(src, "ev_stored_pmc.md"))
tj["slab_pipeline_stored_syndromes_raw_r04_per_2^20"] = {
    "compact_fetch": round(mean_of(stored, "FETCH_SIZE", "slab_compact_kernel") * KIB * 2 / per),
    "compact_write": round(mean_of(stored, "WRITE_SIZE", "slab_compact_kernel") * KIB / per),
    "gather_fetch": round(mean_of(stored, "FETCH_SIZE", "slab_gather_fast_kernel") * KIB * 2 / per),
    "gather_write": round(mean_of(stored, "WRITE_SIZE", "slab_gather_fast_kernel") * KIB / per),
    "algorithmic": 2 ** 20 * 768,
}
small = table(os.path.join(src, "ev_rref_small_pmc.md"))
tj["rref_small_raw_r04"] = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in small.get(c, []):
        if r[0].startswith("rref_small_kernel"):
            tj["rref_small_raw_r04"].setdefault(r[0][:40], {})[c + "_bytes"] = round(r[3] * (2 if c == "FETCH_SIZE" else 1) * KIB)
dense = table(os.path.join(src, "ev_dense_pmc.md"))
fetch, write = mean_of(dense, "FETCH_SIZE", "syndrome_tiled_kernel"), mean_of(dense, "WRITE_SIZE", "syndrome_tiled_kernel")
tj["syndrome_tiled_kernel_raw_r04"] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write}
tj["syndrome_tiled_kernel_bytes_per_launch"] = round(fetch * KIB * 2 + write * KIB)
for shape, calls in (("2048x4096x256", 4), ("32768x65536x1", 2)):
    rows = table(os.path.join(src, "ev_rref_%s_pmc.md" % shape))
    total = sum(v * n * KIB * 2 for _, _, n, v in rows.get("FETCH_SIZE", [])) + sum(v * n * KIB for _, _, n, v in rows.get("WRITE_SIZE", []))
    key = shape.rsplit("x", 1)[0] + "_x" + shape.rsplit("x", 1)[1]
    tj["rref_bytes_per_call"][key] = round(total / calls)
tj["captured_at_commit"] = commit
tj["_how_r04"] = ("round 4: profiles/r04_evidence.sh at commit %s -- rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (program "
                  "after `--`); KiB counters, FETCH_SIZE doubled for gfx950; slab pipeline: `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                  "--no-secondary --no-settle --batch-log2 24 --one-stream` (four passes of 2^22 samples per call; values per 2^20 samples of one "
                  "Pauli component; the combine step rides in the gather launches, its own launch is counted once per 32 passes as in the default "
                  "run); stored syndromes: profiles/time_slabs_stored.py; dense kernel: --algo dense --batch-log2 20; RREF: profiles/time_rref.py "
                  "per shape and profiles/time_rref_small.py. Sources: profiles/r04_slab_pipeline_pmc.md, r04_stored_pmc.md, r04_dense_pmc.md, "
                  "r04_rref_*_pmc.md" % commit)
json.dump(tj, open(tj_path, "w"), indent=1)
print("slab pipeline %d bytes per 2^20 samples, dense %d, rref %s" % (tj["slab_pipeline_bytes_per_launch"],
      tj["syndrome_tiled_kernel_bytes_per_launch"], {k: v for k, v in tj["rref_bytes_per_call"].items() if not k.startswith("_")}))
