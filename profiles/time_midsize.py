"""Mid-size regimes: resident syndrome batches and end-to-end Monte-Carlo for checks between the small-code kernels (n <= 64)
and the n = 4096 benchmark."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quantum_css_codes_amd import _native

def standard(rng, r, n, off):
    h = rng.integers(0, 2, (r, n), dtype=np.uint8)
    h[:, off:off + r] = np.identity(r, dtype=np.uint8)
    return _native.pack_rows(h)

def main():
    ctx = _native.default_context()
    rng = np.random.default_rng(3)
    for (n, r1, r2) in ((127, 63, 63), (255, 127, 127), (511, 255, 255), (1023, 511, 511), (2047, 1023, 1023)):
        c1 = ctx.check_create(standard(rng, r1, n, 0), r1, n)
        c2 = ctx.check_create(standard(rng, r2, n, n - r2), r2, n)
        p = 0.01 / 3
        count = 1 << 23
        ctx.mc_run(c1, c2, 1, 0, 1 << 18, p, p, p, _native.HIST_WEIGHT)
        t0 = time.perf_counter()
        hz, hx = ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
        dt = time.perf_counter() - t0
        assert int(hz.sum()) == count
        # resident batch, histogram only
        lde = _native.words_for(n)
        batch = 1 << 21
        ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
        ctx.sample_errors_dev(n, 5, 0, batch, p, p, p, ex, ez, lde)
        hist = ctx.alloc((r1 + 1) * 8).zero()
        for _ in range(2):
            ctx.syndrome_sparse_dev(c1, ez, batch, lde, None, 0, hist, r1 + 1)
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            ctx.syndrome_sparse_dev(c1, ez, batch, lde, None, 0, hist, r1 + 1)
        ms = ctx.timer_stop() / 10
        print("n=%d r=%d/%d: mc_run %.3e samples/s; resident H1.e_z histogram %.3f ms per 2^21 = %.3e syndromes/s, %.0f GB/s of errors"
              % (n, r1, r2, count / dt, ms, batch / ms * 1e3, batch * lde * 8 / ms / 1e6))
        ex.free(), ez.free(), hist.free()

main()
