#!/usr/bin/env python3
"""The slab pipeline with the syndromes stored (bench.py's secondary.read_write_1536B): both components of 2^22 resident samples of the
benchmark's code, no histogram; used under rocprofv3 for the per-kernel split and the HBM-side traffic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ctx = _native.default_context()
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << 22
path = bench.Path(ctx, "sparse", chk1, chk2, batch, 0)
lde = _native.words_for(bench.N_QUBITS)
s1, s2 = ctx.alloc(batch * path.ls1 * 8), ctx.alloc(batch * path.ls2 * 8)
for _ in range(2):
    ctx.syndrome_sparse_dev(chk1, path.ez, batch, lde, s1, path.ls1)
    ctx.syndrome_sparse_dev(chk2, path.ex, batch, lde, s2, path.ls2)
ctx.sync()
ctx.timer_start()
for _ in range(6):
    ctx.syndrome_sparse_dev(chk1, path.ez, batch, lde, s1, path.ls1)
    ctx.syndrome_sparse_dev(chk2, path.ex, batch, lde, s2, path.ls2)
ms = ctx.timer_stop() / 6
rw = batch * (2 * bench.N_QUBITS / 8.0 + (bench.R1 + bench.R2) / 8.0)
print("slab pipeline, syndromes stored: %.3f ms per 2^22 samples (both components) = %.3e samples/s, %.0f GB/s of 1536 B per sample = %.3f of 8 TB/s"
      % (ms, batch / ms * 1e3, rw / ms / 1e6, rw / ms / 1e6 / 8000))
