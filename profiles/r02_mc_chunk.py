#!/usr/bin/env python3
"""gf2_mc_run at n = 4096 over the chunk size of its three-stream pipeline (GF2_OPT_MC_CHUNK_LOG2): do chunks that fit the
256 MiB Infinity Cache spare the rows their trip through HBM?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native  # noqa: E402

rng = np.random.default_rng(1)
hm1 = rng.integers(0, 2, (2048, 4096), dtype=np.uint8)
hm1[:, :2048] = np.eye(2048, dtype=np.uint8)
hm2 = rng.integers(0, 2, (2047, 4096), dtype=np.uint8)
hm2[:, 2048:4095] = np.eye(2047, dtype=np.uint8)
p = 0.01 / 3
count = 1 << 24
ctx = _native.default_context()
c1 = ctx.check_create(_native.pack_rows(hm1), 2048, 4096)
c2 = ctx.check_create(_native.pack_rows(hm2), 2047, 4096)
ref = None
for k in (21, 20, 19, 18, 22):
    ctx.set_option(_native.OPT_MC_CHUNK_LOG2, k)
    ctx.mc_run(c1, c2, 1, 0, 1 << 22, p, p, p, _native.HIST_WEIGHT)
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        hz, hx = ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    if ref is None:
        ref = (hz.copy(), hx.copy())
    assert np.array_equal(hz, ref[0]) and np.array_equal(hx, ref[1])
    print("chunk 2^%d samples: %.2f ms per 2^24 samples = %.3e samples/s" % (k, best * 1e3, count / best), flush=True)
