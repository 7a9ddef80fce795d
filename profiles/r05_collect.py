#!/usr/bin/env python3
"""Copies the summaries `bash profiles/r05_evidence.sh` left under gpurun_out/r05/ into profiles/ (each with the source digest and
the commit it was taken at in its first line) and rewrites the slab-pipeline and RREF figures of profiles/traffic.json from the
PMC tables.  Refuses when the kernel sources of the working tree are not the ones that were measured (the digest of
profiles/csrc_digest.py, which is also what bench.py checks at run time).
    python profiles/r05_collect.py [gpurun_out/r05]"""
import json
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import csrc_digest  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05"
measured = open(os.path.join(src, "ev_csrc_digest.txt")).read().strip()
if measured != csrc_digest.digest():
    sys.exit("r05_collect: the kernel sources have changed since the evidence run (digest %s.. measured, %s.. now): run "
             "profiles/r05_evidence.sh again" % (measured[:12], csrc_digest.digest()[:12]))
try:
    commit = subprocess.run(["git", "-C", os.path.dirname(here), "rev-parse", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip()[:12]
    dirty = subprocess.run(["git", "-C", os.path.dirname(here), "diff", "--quiet", "HEAD", "--", "quantum_css_codes_amd/csrc", "include"]).returncode != 0
except OSError:
    commit, dirty = "?", False
if dirty:
    sys.exit("r05_collect: quantum_css_codes_amd/csrc or include/ differ from HEAD: commit first, so that the commit named in "
             "traffic.json holds the measured sources")
header = "<!-- rocprofv3, profiles/r05_evidence.sh, kernel sources %s.. at commit %s (round 5) -->\n" % (measured[:12], commit)


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
        with open(os.path.join(here, "r05_" + m.group(1) + ".md"), "w") as out:
            out.write(header)
            out.write(open(os.path.join(src, name)).read())

tj_path = os.path.join(here, "traffic.json")
tj = json.load(open(tj_path))
KIB = 1024


def mean_of(rows, counter, kernel):
    hit = [r for r in rows.get(counter, []) if r[0].startswith(kernel)]
    return max(hit, key=lambda r: r[1] * r[2])[3] if hit else 0.0        # the benchmark's launches, not the oracle check's small one


slab = table(os.path.join(src, "ev_slab_pipeline_pmc.md"))
per = 4.0                                                                # one pass = 2^22 samples, figures per 2^20
raw = {
    "compact_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_compact_kernel") * KIB * 2 / per),
    "compact_write": round(mean_of(slab, "WRITE_SIZE", "slab_compact_kernel") * KIB / per),
    "gather_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_gather_fast_kernel") * KIB * 2 / per),
    "gather_write": round(mean_of(slab, "WRITE_SIZE", "slab_gather_fast_kernel") * KIB / per),
    "combine_fetch_one_launch_per_32_passes": round(mean_of(slab, "FETCH_SIZE", "slab_combine_kernel") * KIB * 2 / per / 32),
    "redo_fetch": round(mean_of(slab, "FETCH_SIZE", "slab_redo_kernel") * KIB * 2 / per),
}
tj["slab_pipeline_raw_r05_per_2^20"] = raw
tj["slab_pipeline_bytes_per_launch"] = sum(raw.values())
calls = {"2048x4096x1": 4, "2048x4096x256": 4, "32768x65536x1": 2}      # timed calls of profiles/time_rref.py per shape
for shape, n_calls in calls.items():
    rows = table(os.path.join(src, "ev_rref_%s_pmc.md" % shape))
    total = sum(v * n * KIB * 2 for _, _, n, v in rows.get("FETCH_SIZE", [])) + sum(v * n * KIB for _, _, n, v in rows.get("WRITE_SIZE", []))
    key = shape.rsplit("x", 1)[0] + "_x" + shape.rsplit("x", 1)[1]
    tj["rref_bytes_per_call"][key] = round(total / n_calls)
tj["captured_at_commit"] = commit
tj["captured_at_csrc_sha256"] = measured
tj["_how_r05"] = ("round 5: profiles/r05_evidence.sh, kernel sources %s (profiles/csrc_digest.py) at commit %s -- rocprofv3 --pmc FETCH_SIZE / "
                  "--pmc WRITE_SIZE in separate passes (program after `--`); KiB counters, FETCH_SIZE doubled for gfx950; slab pipeline: "
                  "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream`, values "
                  "per 2^20 samples of one Pauli component; RREF: profiles/time_rref.py per shape, bytes of the whole call (every kernel of "
                  "it).  The other entries (dense kernel, stored syndromes, small matrices) are round 4's: their kernels have not changed.  "
                  "Sources: profiles/r05_slab_pipeline_pmc.md, r05_rref_*_pmc.md" % (measured[:12], commit))
json.dump(tj, open(tj_path, "w"), indent=1)
print("slab pipeline %d bytes per 2^20 samples; rref %s" % (tj["slab_pipeline_bytes_per_launch"],
      {k: v for k, v in tj["rref_bytes_per_call"].items() if not k.startswith("_")}))
