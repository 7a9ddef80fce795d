#!/usr/bin/env python3
"""Soak of the streamed sweeps (more than 4096 rows) against the C oracle: random shapes, densities, batches and structure, many
seeds, with and without look-ahead.   python profiles/r05_soak.py [seeds] [first seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

ctx = _native.default_context()
seeds, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 10), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = cases = 0
t0 = time.time()
for seed in range(first, first + seeds):
    rng = np.random.default_rng(seed)
    for case in range(6):
        m = int(rng.integers(4097, 12000))
        n = int(rng.choice([int(rng.integers(1, 400)), int(rng.integers(400, 3000)), int(rng.integers(3000, 9000))]))
        batch = int(rng.integers(1, 4)) if m * n < 3e7 else 1
        mats = []
        for b in range(batch):
            a = (rng.random((m, n)) < float(rng.choice([0.002, 0.05, 0.5, 0.97]))).astype(np.uint8)
            kind = int(rng.integers(0, 6))
            if kind == 0:
                a[:, : int(rng.integers(1, n + 1))] = 0
            if kind == 1:
                a[m // 3:] = a[: m - m // 3]
            if kind == 2:
                a[:, ::2] = 0
            if kind == 3:
                a[rng.integers(0, m, 50)] = 0
            if kind == 4 and n > 130:
                lo = int(rng.integers(0, n - 128))
                a[:, lo:lo + 128] = 0
            mats.append(a)
        want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
        for flag in (0, 1 << 16):                                      # GF2_F_RREF_NO_LOOKAHEAD
            flags = ctx.get_flags()
            ctx.set_flags(flags | flag)
            packed = np.stack([_native.pack_rows(a) for a in mats])
            pivots, ranks = ctx.rref_batch(packed, batch, m, n)
            ctx.set_flags(flags)
            ok = all(ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]) and
                     list(pivots[b, :want[b][2]]) == list(want[b][1]) for b in range(batch))
            cases += 1
            if not ok:
                bad += 1
                print("MISMATCH seed %d case %d: %d x %d x %d flags %#x" % (seed, case, m, n, batch, flag), flush=True)
    print("seed %d done, %d cases, %d bad, %.0f s" % (seed, cases, bad, time.time() - t0), flush=True)
print("soak: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
