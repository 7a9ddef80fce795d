#!/usr/bin/env python3
"""Same-process A/B of routing flags / options on the benchmark's two-stream step: the code, the contexts and the resident errors are
made once, then the variants alternate ROUNDS times, STEPS steps each (HIP events on the main stream, the side stream joined).
    python3 profiles/ab_inprocess.py [--batch-log2 27] [--steps 12] [--rounds 8] [--one-stream] VARIANT [VARIANT ...]
VARIANT = comma-separated settings: f=<flags, any base> and/or o<option number>=<value>, e.g.  f=0  f=0x40000  f=0,o0=23"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-log2", type=int, default=27)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--one-stream", action="store_true")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

ctx = _native.default_context()
ctx2 = None if args.one_stream else _native.Context(ctx.device)
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
path = bench.Path(ctx, "sparse", chk1, chk2, batch, 0, ctx2)


def apply(variant):
    flags, opts = 0, {}
    for item in variant.split(","):
        key, value = item.split("=")
        if key == "f":
            flags = int(value, 0)
        else:
            opts[int(key[1:])] = int(value)
    for c in path.contexts():
        c.set_flags(flags)
        for k in range(14):
            c.set_option(k, opts.get(k))


results = {v: [] for v in args.variants}
for v in args.variants:                                   # first touch of every variant (workspaces)
    apply(v)
    path.step()
    path.sync()
for rnd in range(args.rounds):
    for v in args.variants:
        apply(v)
        path.step()
        path.sync()
        ctx.timer_start()
        for _ in range(args.steps):
            path.step()
        if path.ctx2 is not None:
            path.ctx2.sync()
        ms = ctx.timer_stop() / args.steps
        results[v].append(2 * batch * 512 / ms / 1e6 / bench.HBM_PEAK_GBS)
    print("round %d: " % rnd + "  ".join("%s %.4f" % (v, results[v][-1]) for v in args.variants), flush=True)
for v in args.variants:
    r = results[v]
    print("%-24s median %.4f  mean %.4f  min %.4f  max %.4f  n %d" % (v, statistics.median(r), statistics.mean(r), min(r), max(r), len(r)))
