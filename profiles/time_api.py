"""Wall time of the drop-in Python calls on the n = 4096 shapes (dense int NumPy arrays in and out, as the reference's API has
them): what a user of bin_matrix / css_code sees, conversions and transfers included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quantum_css_codes_amd import bin_matrix, css_code

def best_of(fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out

def main():
    rng = np.random.default_rng(4096)
    a = rng.integers(0, 2, (2048, 4096))
    bin_matrix.reduced_row_echelon_form(a[:64])
    t, red = best_of(lambda: bin_matrix.reduced_row_echelon_form(a))
    print("bin_matrix.reduced_row_echelon_form(2048 x 4096 int64): %.1f ms (reference: 8 s measured in SURVEY.md)" % (t * 1e3))
    t, ns = best_of(lambda: bin_matrix.nullspace(a))
    print("bin_matrix.nullspace(2048 x 4096): %.1f ms, %d basis rows" % (t * 1e3, ns.shape[0]))
    copies = [np.array(a) for _ in range(3)]                  # the call works in place: a fresh input per repetition, made outside the timed call
    t, (h, swaps) = best_of(lambda: css_code.normalize_parity_check(copies.pop(), 0))
    print("css_code.normalize_parity_check(2048 x 4096, 0): %.1f ms, %d swaps" % (t * 1e3, len(swaps)))
    h2 = ns[:2047]
    t, code = best_of(lambda: css_code.CSSCode(a, h2, max_table_weight=1), reps=2)
    print("css_code.CSSCode(H1 2048 x 4096, H2 2047 x 4096, tables capped at weight 1): %.0f ms, k = %d" % (t * 1e3, code.k))
    e = rng.integers(0, 2, (4096, 4096))
    t, s = best_of(lambda: css_code.syndrome_batch(a, e))
    print("css_code.syndrome_batch(H1, 4096 dense error vectors): %.1f ms (reference: 24 ms per vector)" % (t * 1e3))

main()
