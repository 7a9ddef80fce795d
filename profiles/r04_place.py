#!/usr/bin/env python3
"""Consecutive bench processes alternate between a faster and a slower state.  Is it WHERE the resident errors lie in HBM?  One
process, three sets of resident errors of 2^26 samples (64 GiB each, allocated one after the other), the two-stream step timed on
each in turn.   python3 profiles/r04_place.py [--sets 3] [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-log2", type=int, default=26)
ap.add_argument("--sets", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

ctx = _native.default_context()
side = _native.Context(ctx.device)
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
paths = [bench.Path(ctx, "sparse", chk1, chk2, batch, 0, side) for _ in range(args.sets)]
for p in paths:
    print("set at ez %#x ex %#x" % (p.ez.ptr, p.ex.ptr), flush=True)
for rnd in range(3):
    for k, p in enumerate(paths):
        p.step()
        p.sync()
        ctx.timer_start()
        for _ in range(args.steps):
            p.step()
        side.sync()
        ms = ctx.timer_stop() / args.steps
        print("round %d  set %d: %.3f ms  %.4f" % (rnd, k, ms, 2 * batch * bench.N_QUBITS / 8.0 / (ms * 1e-3) / 8.0e12), flush=True)
