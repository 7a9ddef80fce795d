#!/usr/bin/env python3
"""Does it matter where the side context's records and partial weights lie in HBM relative to the main context's?  One process, one
side context whose workspace is allocated once (at the largest lead); then GF2_OPT_SLAB_WS_LEAD (4 KiB pages) of the side
context only is swept, the step timed at each value, twice over.   python3 profiles/r04_lead.py [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-log2", type=int, default=27)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--leads", default="0,1,2,3,4,8,16,32,64,128,256,512,1024,2048,4096,8192,16384,32768,65536")
args = ap.parse_args()
leads = [int(v) for v in args.leads.split(",")]

ctx = _native.default_context()
code, h1, h2 = bench.build_code()
chk1, chk2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
batch = 1 << args.batch_log2
side = _native.Context(ctx.device)
side.set_option(_native.OPT_SLAB_WS_LEAD, max(leads))
path = bench.Path(ctx, "sparse", chk1, chk2, batch, 0, side)
path.step()
path.sync()
for rnd in range(2):
    for lead in leads:
        side.set_option(_native.OPT_SLAB_WS_LEAD, lead)
        path.step()
        path.sync()
        ctx.timer_start()
        for _ in range(args.steps):
            path.step()
        side.sync()
        ms = ctx.timer_stop() / args.steps
        print("round %d  lead %6d pages: %.3f ms  %.4f" % (rnd, lead, ms, 2 * batch * bench.N_QUBITS / 8.0 / (ms * 1e-3) / 8.0e12), flush=True)
