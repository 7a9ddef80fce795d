#!/usr/bin/env python3
"""What if the identity half and the other half of every error row lay in arrays of their own (pitch 256 bytes each)?  The compact
kernel would read its 256 bytes per sample back to back instead of every other 256-byte block, the gather kernel its 64-byte
pieces at a pitch of 256 instead of 512 bytes.  Timing only (wrong results): the library built with -DWHATIF_HALF_PITCH accepts
lde = 32 for the n = 4096 check H1 = [I | A], which makes "word 32..63 of row j" the 256 bytes that follow row j's -- exactly that
access pattern on the resident buffers.  Both streams run H1 (H2's identity block is in the second half of a row).
    python3 profiles/r04_half_pitch.py [--steps 10]        (with scratch_ab/halfpitch.so copied over the library)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-log2", type=int, default=27)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

ctx = _native.default_context()
side = _native.Context(ctx.device)
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
path = bench.Path(ctx, "sparse", chk1, chk2, batch, 0, side)
hz2 = ctx.alloc((bench.R1 + 1) * 8).zero()
for rnd in range(3):
    for lde in (64, 32):
        def step():
            ctx.syndrome_sparse_dev(chk1, path.ez, batch, lde, None, 0, path.hz, bench.R1 + 1)
            side.syndrome_sparse_dev(chk1, path.ex, batch, lde, None, 0, hz2, bench.R1 + 1)
        step()
        ctx.sync(), side.sync()
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        side.sync()
        ms = ctx.timer_stop() / args.steps
        print("round %d  pitch %d words: %.3f ms  %.4f of the peak on 2 x 512 bytes per sample" % (rnd, lde, ms, 2 * batch * 512.0 / (ms * 1e-3) / 8.0e12), flush=True)
