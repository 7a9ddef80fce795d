#!/usr/bin/env python3
"""End-to-end gf2_mc_run (sampler included) on the n = 4096 workload of bench.py."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native  # noqa: E402

ctx = _native.default_context()
rng = np.random.default_rng(1)
hm1 = rng.integers(0, 2, (2048, 4096), dtype=np.uint8)
hm1[:, :2048] = np.eye(2048, dtype=np.uint8)
hm2 = rng.integers(0, 2, (2047, 4096), dtype=np.uint8)
hm2[:, 2048:4095] = np.eye(2047, dtype=np.uint8)
c1 = ctx.check_create(_native.pack_rows(hm1), 2048, 4096)
c2 = ctx.check_create(_native.pack_rows(hm2), 2047, 4096)
p = 0.01 / 3
count = 1 << 24
ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)          # (workspaces are sized by the first call)
for waves in [int(w) for w in os.environ.get("GF2_SAMPLER_WAVES", "").split()]:      # record-sampler wavefronts per CU, in turn
    ctx.set_option(_native.OPT_MC_SAMPLER_WAVES, waves)
    t = []
    for rep in range(4):
        t0 = time.perf_counter()
        ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
        t.append(time.perf_counter() - t0)
    print("  %2d sampler wavefronts per CU: %.3e samples/s" % (waves, count / min(t)))
ctx.set_option(_native.OPT_MC_SAMPLER_WAVES, None)
for cap in [int(w) for w in os.environ.get("GF2_TAIL_CAP", "").split()]:      # qubits of a segment the sampler's lanes take in step
    ctx.set_option(_native.OPT_MC_TAIL_CAP, cap)
    t = []
    for rep in range(4):
        t0 = time.perf_counter()
        ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
        t.append(time.perf_counter() - t0)
    print("  in step up to %d qubits per segment: %.3e samples/s" % (cap, count / min(t)))
ctx.set_option(_native.OPT_MC_TAIL_CAP, None)
times = []
for rep in range(6):
    t0 = time.perf_counter()
    hz, hx = ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
    times.append(time.perf_counter() - t0)
assert int(hz.sum()) == count
print("gf2_mc_run n=4096 end to end (sampler + syndromes + histograms), %d samples, ms per call: %s" % (count, " ".join("%.2f" % (t * 1e3) for t in times)))
print("gf2_mc_run n=4096 end to end (sampler + syndromes + histograms): %.3e samples/s" % (count / min(times)))
