import sys, time, cProfile, pstats
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quantum_css_codes_amd import css_code, bin_matrix, _native
rng = np.random.default_rng(4096)
a = rng.integers(0, 2, (2048, 4096))
css_code.normalize_parity_check(np.array(a), 0)
copies = [np.array(a) for _ in range(4)]
pr = cProfile.Profile()
pr.enable()
for c in copies:
    css_code.normalize_parity_check(c, 0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
ctx = _native.default_context()
p = _native.pack_rows(a)
for name, fn in (("pack", lambda: _native.pack_rows(a)), ("normalize packed", lambda: ctx.normalize(p.copy(), 2048, 4096, 0)),
                 ("unpack", lambda: _native.unpack_rows(p, 4096)), ("unpack_into", lambda: _native.unpack_rows_into(p, copies[0]))):
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    print(name, "%.2f ms" % (best * 1e3))
