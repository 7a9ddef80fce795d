#!/usr/bin/env python3
"""Where does the run-to-run spread of the two-stream step come from?  One process, the resident errors made once; then several
SIDE contexts (= several HIP streams with workspaces of their own) in turn, each timed ROUNDS times.  If the side contexts differ
among themselves as much as processes do, the spread is a property of the stream / workspace a process happens to get.
    python3 profiles/r04_modes.py [--sides 6] [--rounds 3] [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-log2", type=int, default=27)
ap.add_argument("--sides", type=int, default=6)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

ctx = _native.default_context()
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
first = _native.Context(ctx.device)
path = bench.Path(ctx, "sparse", chk1, chk2, batch, 0, first)
lde = _native.words_for(bench.N_QUBITS)
peak = 8.0e12
sides = [first] + [_native.Context(ctx.device) for _ in range(args.sides - 1)]
for rnd in range(args.rounds):
    for k, side in enumerate(sides):
        def step():
            ctx.syndrome_sparse_dev(chk1, path.ez, batch, lde, None, 0, path.hz, bench.R1 + 1)
            side.syndrome_sparse_dev(chk2, path.ex, batch, lde, None, 0, path.hx, bench.R2 + 1)
        step()
        ctx.sync(), side.sync()
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        side.sync()
        ms = ctx.timer_stop() / args.steps
        frac = 2 * batch * bench.N_QUBITS / 8.0 / (ms * 1e-3) / peak
        print("round %d  side context %d: %.3f ms  %.4f" % (rnd, k, ms, frac), flush=True)
