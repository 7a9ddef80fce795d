#!/usr/bin/env python3
"""Which side context is fast?  One process = one configuration: DUMMIES contexts are created before the side context (their
streams take hardware queues first) and, with --close, destroyed again before anything runs.
    python3 profiles/r04_modes2.py --dummies K [--close] [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-log2", type=int, default=27)
ap.add_argument("--dummies", type=int, default=0)
ap.add_argument("--close", action="store_true")
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

ctx = _native.default_context()
dummies = [_native.Context(ctx.device) for _ in range(args.dummies)]
side = _native.Context(ctx.device)
if args.close:
    for d in dummies:
        d.close()
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
path = bench.Path(ctx, "sparse", chk1, chk2, batch, 0, side)
path.step()
path.sync()
out = []
for rnd in range(2):
    ctx.timer_start()
    for _ in range(args.steps):
        path.step()
    side.sync()
    ms = ctx.timer_stop() / args.steps
    out.append("%.4f" % (2 * batch * bench.N_QUBITS / 8.0 / (ms * 1e-3) / 8.0e12))
print("dummies %d close %d: %s" % (args.dummies, int(args.close), " ".join(out)), flush=True)
