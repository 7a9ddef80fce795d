#!/usr/bin/env python3
"""Does it help to allocate the resident errors FIRST in the process (before the code is built: its workspaces, tables and
temporaries)?   python3 profiles/r04_place2.py first|last [--steps 10]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("order", choices=("first", "last"))
ap.add_argument("--batch-log2", type=int, default=27)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

ctx = _native.default_context()
batch = 1 << args.batch_log2
lde = _native.words_for(bench.N_QUBITS)


def resident():
    ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
    p = bench.P_TOTAL / 3
    for done in range(0, batch, 1 << 21):
        _native.check(_native.lib().gf2_sample_errors_dev(ctx.handle, bench.N_QUBITS, bench.SEED, done, min(1 << 21, batch - done), p, p, p,
                                                          ex.ptr + done * lde * 8, ez.ptr + done * lde * 8, lde, _native.LAYOUT_SAMPLE_MAJOR))
    return ex, ez


if args.order == "first":
    ex, ez = resident()
side = _native.Context(ctx.device)
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
if args.order == "last":
    ex, ez = resident()
hz, hx = ctx.alloc((bench.R1 + 1) * 8).zero(), ctx.alloc((bench.R2 + 1) * 8).zero()


def step():
    ctx.syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, bench.R1 + 1)
    side.syndrome_sparse_dev(chk2, ex, batch, lde, None, 0, hx, bench.R2 + 1)


step()
ctx.sync(), side.sync()
out = []
for rnd in range(2):
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    side.sync()
    ms = ctx.timer_stop() / args.steps
    out.append("%.4f" % (2 * batch * bench.N_QUBITS / 8.0 / (ms * 1e-3) / 8.0e12))
print("resident errors allocated %s: %s   (ez %#x)" % (args.order, " ".join(out), ez.ptr), flush=True)
