#!/usr/bin/env python3
"""Times gf2_normalize_dev (css_code.normalize_parity_check) on resident matrices of BASELINE config 4."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native, bin_matrix  # noqa: E402

ctx = _native.default_context()
h1 = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.uint8)
h2 = bin_matrix.nullspace(h1)[:2047].astype(np.uint8)
for name, mat, off in (("H1 2048x4096 offset 0", h1, 0), ("H2 2047x4096 offset 2048", h2, 2048)):
    a = _native.pack_rows(mat)
    r, n = mat.shape
    buf = ctx.alloc(a.nbytes)
    swaps, nsw, status = ctx.alloc(2 * r * 8), ctx.alloc(8), ctx.alloc(4)
    best = None
    for _ in range(3):
        buf.upload(a)
        ctx.timer_start()
        _native.check(_native.lib().gf2_normalize_dev(ctx.handle, buf.ptr, r, n, a.shape[1], off, swaps.ptr, nsw.ptr, status.ptr))
        ms = ctx.timer_stop()
        best = ms if best is None else min(best, ms)
    print("%s: %.3f ms, %d swaps, status %d  (sequential=%s)" % (name, best, int(nsw.download((1,), np.int64)[0]),
          int(status.download((1,), np.int32)[0]), os.environ.get("GF2_NORMALIZE_SEQUENTIAL")))
