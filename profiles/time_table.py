"""Times css_code.syndrome_table on the device (gf2_syndrome_table) against the oracle's Python loop on the same code."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quantum_css_codes_amd import _native, css_code
from oracle import cpu_ref

def main():
    ctx = _native.default_context()
    rng = np.random.default_rng(3)
    for (r, n, cap, with_oracle) in ((3, 7, None, True), (10, 15, None, True), (15, 31, None, True), (18, 63, None, False),
                                     (24, 64, None, False), (24, 64, 3, False)):
        h = rng.integers(0, 2, (r, n))
        if (r, n) == (3, 7):
            h = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
        packed = _native.pack_rows(h)
        ctx.syndrome_table(packed, r, n, cap)
        t0 = time.perf_counter()
        t, dense = ctx.syndrome_table(packed, r, n, cap)
        dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        t_api, table = css_code.syndrome_table(h, max_weight=cap)
        api = time.perf_counter() - t0
        line = "r=%d n=%d cap=%s: t=%d, %d entries, C ABI %.2f ms, css_code.syndrome_table (dict of vectors) %.1f ms" % (
            r, n, cap, t, len(table), dev * 1e3, api * 1e3)
        if with_oracle:
            t0 = time.perf_counter()
            want_t, want = cpu_ref.syndrome_table(h, max_weight=cap)
            line += ", reference-style Python loop %.1f ms" % ((time.perf_counter() - t0) * 1e3)
            assert want_t == t and len(want) == len(table)
        print(line)

main()
