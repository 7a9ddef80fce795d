#!/usr/bin/env python3
"""Times gf2_rref_batch_dev on resident random matrices (HIP events on the context's stream).
    python profiles/time_rref.py [m n batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native  # noqa: E402

m, n, batch = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2048, 4096, 1)
ctx = _native.default_context()
rng = np.random.default_rng(4096)
ld = (n + 63) // 64
a = (rng.integers(0, 2**63, (m, ld), dtype=np.int64).view(np.uint64) << np.uint64(1)) | \
    rng.integers(0, 2, (m, ld), dtype=np.int64).view(np.uint64)            # uniformly random packed words
if n % 64:
    a[:, -1] &= np.uint64((1 << (n % 64)) - 1)
buf = ctx.alloc(batch * a.nbytes)
piv, rk = ctx.alloc(batch * min(m, n) * 8), ctx.alloc(batch * 8)
best = None
for _ in range(2 if m > 8192 else 4):
    for b in range(batch):
        _native.check(_native.lib().gf2_h2d(ctx.handle, buf.ptr + b * a.nbytes, a.ctypes.data, a.nbytes))
    ctx.timer_start()
    _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, m, n, a.shape[1], piv.ptr, rk.ptr))
    ms = ctx.timer_stop()
    best = ms if best is None else min(best, ms)
ranks = rk.download((batch,), np.int64)
print("rank", int(ranks[0]))
print("rref %dx%d batch %d: %.3f ms  %.2f GB/s (2*m*ld*8 bytes per matrix)" % (m, n, batch, best, batch * 2 * a.nbytes / best / 1e6))
