#!/usr/bin/env python3
"""Rows of 512 bytes put a row part of every sample at the same offset modulo 512: does a row pitch of 576 bytes (lde = 72 words)
spread the gather kernel's 64-byte reads better over the HBM channels?  One process, two sets of resident errors of 2^26 samples
(pitch A allocated first, pitch B second; run the script both ways round: where a set lies matters too).
    python3 profiles/r04_lde.py 64,72 | 72,64 [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("pitches")
ap.add_argument("--batch-log2", type=int, default=26)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

ctx = _native.default_context()
side = _native.Context(ctx.device)
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
p = bench.P_TOTAL / 3
sets = []
for lde in [int(v) for v in args.pitches.split(",")]:
    ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
    for done in range(0, batch, 1 << 21):
        _native.check(_native.lib().gf2_sample_errors_dev(ctx.handle, bench.N_QUBITS, bench.SEED, done, min(1 << 21, batch - done), p, p, p,
                                                          ex.ptr + done * lde * 8, ez.ptr + done * lde * 8, lde, _native.LAYOUT_SAMPLE_MAJOR))
    sets.append((lde, ex, ez, ctx.alloc((bench.R1 + 1) * 8).zero(), ctx.alloc((bench.R2 + 1) * 8).zero()))
ctx.sync()
hists = []
for rnd in range(3):
    for lde, ex, ez, hz, hx in sets:
        def step():
            ctx.syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, bench.R1 + 1)
            side.syndrome_sparse_dev(chk2, ex, batch, lde, None, 0, hx, bench.R2 + 1)
        step()
        ctx.sync(), side.sync()
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        side.sync()
        ms = ctx.timer_stop() / args.steps
        print("round %d  pitch %d words: %.3f ms  %.4f" % (rnd, lde, ms, 2 * batch * bench.N_QUBITS / 8.0 / (ms * 1e-3) / 8.0e12), flush=True)
import numpy as np  # noqa: E402
a = sets[0][3].download((bench.R1 + 1,), np.uint64)
b = sets[1][3].download((bench.R1 + 1,), np.uint64)
print("same histograms:", bool(np.array_equal(a, b)))
