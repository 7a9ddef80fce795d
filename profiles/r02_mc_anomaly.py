#!/usr/bin/env python3
"""gf2_mc_run at n = 4096 from several contexts of one process (each context = its own three HIP streams): is the slow case
(about 20 instead of 12 ms per 2^24 samples, seen in about one process in four) a property of the process or of the streams?"""
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
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
diag = len(sys.argv) > 2 and sys.argv[2] == "diag"
ctxs = []
for k in range(nctx):
    ctx = _native.Context(0)
    ctxs.append(ctx)
    c1 = ctx.check_create(_native.pack_rows(hm1), 2048, 4096)
    c2 = ctx.check_create(_native.pack_rows(hm2), 2047, 4096)
    ctx.mc_run(c1, c2, 1, 0, 1 << 21, p, p, p, _native.HIST_WEIGHT)
    times = []
    if diag:
        ctx.set_flags(ctx.get_flags() | _native.F_DIAG_MC_TIMES)
    for rep in range(5):
        t0 = time.perf_counter()
        hz, hx = ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
        times.append((time.perf_counter() - t0) * 1e3)
    assert int(hz.sum()) == count
    print("pid %d context %d: %s ms per 2^24 samples" % (os.getpid(), k, " ".join("%.2f" % t for t in times)), flush=True)
