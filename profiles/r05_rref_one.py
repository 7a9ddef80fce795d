#!/usr/bin/env python3
"""One shape of gf2_rref_batch_dev under the internal options, for rocprofv3:  python profiles/r05_rref_one.py m n batch K [rows_wg [variant]]
(K: panels per sweep, 4 / 2; 0 = the round-4 pair kernels; variant: GF2_OPT_RREF_STREAM_VARIANT, streamed sweeps above 4096 rows)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native  # noqa: E402

m, n, batch, k = (int(v) for v in sys.argv[1:5])
rows_wg = int(sys.argv[5]) if len(sys.argv) > 5 else -1
ctx = _native.default_context()
ctx.set_option(13, k)
ctx.set_option(12, rows_wg)
variant = int(sys.argv[6]) if len(sys.argv) > 6 else -1
ctx.set_option(11, variant)
if len(sys.argv) > 7:
    ctx.set_flags(ctx.get_flags() | int(sys.argv[7], 0))        # e.g. 0x10000 = GF2_F_RREF_NO_LOOKAHEAD
rng = np.random.default_rng(4096)
ld = (n + 63) // 64
mats = [(rng.integers(0, 2**63, (m, ld), dtype=np.int64).view(np.uint64) << np.uint64(1)) |
        rng.integers(0, 2, (m, ld), dtype=np.int64).view(np.uint64) for _ in range(min(batch, 64))]
nb = mats[0].nbytes
buf = ctx.alloc(batch * nb)
piv, rk = ctx.alloc(batch * min(m, n) * 8), ctx.alloc(batch * 8)
best = None
for _ in range(2 if m > 8192 else 4):
    for b in range(batch):
        _native.check(_native.lib().gf2_h2d(ctx.handle, buf.ptr + b * nb, mats[b % len(mats)].ctypes.data, nb))
    ctx.timer_start()
    _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, m, n, ld, piv.ptr, rk.ptr))
    ms = ctx.timer_stop()
    best = ms if best is None else min(best, ms)
print("rref %dx%d x%d K=%d rows_wg=%d variant=%d: %.3f ms  %.2f GB/s  rank %d" %
      (m, n, batch, k, rows_wg, variant, best, batch * 2 * nb / best / 1e6, int(rk.download((batch,), np.int64).min())))
