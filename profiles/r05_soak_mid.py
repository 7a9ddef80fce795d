#!/usr/bin/env python3
"""Soak of the blocked RREF up to 4096 rows (fused panel kernel, K = 4 / 2 by the shape or forced, snapshots or pivot rows read in
place, small and large batches) against the C oracle.   python profiles/r05_soak_mid.py [seeds] [first seed]"""
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
flags0 = ctx.get_flags()
for seed in range(first, first + seeds):
    rng = np.random.default_rng(seed)
    for case in range(8):
        m = int(rng.choice([int(rng.integers(65, 300)), int(rng.integers(300, 1100)), int(rng.integers(1100, 4097))]))
        n = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 2500)), int(rng.integers(2500, 9000))]))
        big_batch = case % 4 == 3
        distinct = int(rng.integers(1, 6))
        batch = int(rng.integers(1, 5)) if not big_batch else int(max(distinct, min(600, (300 * 1024 * 1024) // (m * ((n + 63) // 64) * 8 * 4))))
        mats = []
        for b in range(distinct):
            a = (rng.random((m, n)) < float(rng.choice([0.002, 0.05, 0.5, 0.97]))).astype(np.uint8)
            kind = int(rng.integers(0, 6))
            if kind == 0:
                a[:, : int(rng.integers(1, n + 1))] = 0
            if kind == 1:
                a[m // 3:] = a[: m - m // 3]
            if kind == 2:
                a[:, ::2] = 0
            if kind == 3:
                a[rng.integers(0, m, 20)] = 0
            if kind == 4 and n > 130:
                lo = int(rng.integers(0, n - 128))
                a[:, lo:lo + 128] = 0
            mats.append(a)
        if not big_batch:
            batch = max(batch, distinct)
        want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
        one = [_native.pack_rows(a) for a in mats]
        for k in (None, 4, 2):
            for variant in (None, 1):
                ctx.set_flags(flags0 | _native.F_RREF_NO_SMALL)
                ctx.set_option(_native.OPT_RREF_SWEEP_K, k)
                ctx.set_option(_native.OPT_RREF_STREAM_VARIANT, variant)
                packed = np.stack([one[b % distinct] for b in range(batch)])
                pivots, ranks = ctx.rref_batch(packed, batch, m, n)
                ok = all(ranks[b] == want[b % distinct][2] and np.array_equal(packed[b], want[b % distinct][0]) and
                         list(pivots[b, :want[b % distinct][2]]) == list(want[b % distinct][1]) for b in range(batch))
                cases += 1
                if not ok:
                    bad += 1
                    print("MISMATCH seed %d case %d: %d x %d x %d K=%s variant=%s" % (seed, case, m, n, batch, k, variant), flush=True)
    print("seed %d done, %d cases, %d bad, %.0f s" % (seed, cases, bad, time.time() - t0), flush=True)
ctx.set_flags(flags0)
ctx.set_option(_native.OPT_RREF_SWEEP_K, None)
ctx.set_option(_native.OPT_RREF_STREAM_VARIANT, None)
print("soak: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
