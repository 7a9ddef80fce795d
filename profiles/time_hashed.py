#!/usr/bin/env python3
"""Mid-size k = 1 CSS codes through the device hash tables (gf2_syndrome_table_hashed, gf2_mc_decode_hashed): wall time of the
CSSCode constructor (both syndrome tables) and the rate of the table decode + logical tally."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import bin_matrix, montecarlo  # noqa: E402
from quantum_css_codes_amd.css_code import CSSCode  # noqa: E402


def dual_pair(rng, n, r1):
    while True:
        h1 = rng.integers(0, 2, (r1, n))
        if bin_matrix.rank(h1) == r1:
            break
    null = bin_matrix.nullspace(h1)
    return h1, null[: null.shape[0] - 1]


for (n, r1, cap) in ((47, 23, None), (63, 31, None), (95, 47, 3), (127, 63, 2), (128, 64, 3)):
    rng = np.random.default_rng(n + r1)
    h1, h2 = dual_pair(rng, n, r1)
    CSSCode(h1, h2, max_table_weight=cap)                               # (first use: library, workspaces)
    t0 = time.perf_counter()
    code = CSSCode(h1, h2, max_table_weight=cap)
    build = time.perf_counter() - t0
    p = (0.003, 0.003, 0.003)
    count = 10**7
    montecarlo.decode_local(code, count, *p, seed=1)
    t0 = time.perf_counter()
    got = montecarlo.decode_local(code, count, *p, seed=1)
    dt = time.perf_counter() - t0
    print("n=%3d r=%d+%d cap=%s: CSSCode() %.1f ms (t=%d, %d + %d table entries); decode + tally of 10^7 samples %.1f ms = %.2e samples/s"
          % (n, code.r_1, code.r_2, cap, build * 1e3, code.t, len(code._c1_syndromes), len(code._c2_syndromes), dt * 1e3, count / dt), flush=True)
