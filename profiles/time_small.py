#!/usr/bin/env python3
"""Small-code measurements (BASELINE.json configs[1], configs[2]): Monte-Carlo pipeline throughput and the
streaming syndrome kernels on resident errors, sample-major and bit-sliced, against the HBM roofline."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native  # noqa: E402
from quantum_css_codes_amd.css_code import CSSCode  # noqa: E402

ctx = _native.default_context()
steane = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
cols = np.arange(1, 16)
h1 = np.array([(cols >> b) & 1 for b in range(4)])
rm_h2 = np.vstack([h1] + [h1[a] & h1[b] for a in range(4) for b in range(a + 1, 4)])
for name, code, count in (("steane", CSSCode(steane, steane), 10**8), ("rm15", CSSCode(h1, rm_h2), 10**8)):
    code.monte_carlo(10**6, 0.01 / 3, 0.01 / 3, 0.01 / 3, seed=1)
    t0 = time.perf_counter()
    res = code.monte_carlo(count, 0.01 / 3, 0.01 / 3, 0.01 / 3, seed=1)
    dt = time.perf_counter() - t0
    assert int(res['hist_z'].sum()) == count
    print("%s monte_carlo (sampler + syndromes + full histograms): %.3e samples/s" % (name, count / dt))
    n, r1, r2 = code.n, code.r_1, code.r_2
    batch = 1 << 27
    hp = _native.pack_rows(code.parity_check_c2)
    chk = ctx.check_create(hp, r2, n)
    words = batch // 64
    # sample-major: 8 B in, 8 B out per sample and component
    e = ctx.alloc(batch * 8)
    e2 = ctx.alloc(batch * 8)
    s = ctx.alloc(batch * 8)
    ctx.sample_errors_dev(n, 3, 0, batch, 0.1, 0.1, 0.1, e, e2, 1)
    ctx.syndrome_dev(chk, e, batch, 1, s, 1, _native.LAYOUT_SAMPLE_MAJOR)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        ctx.syndrome_dev(chk, e, batch, 1, s, 1, _native.LAYOUT_SAMPLE_MAJOR)
    ms = ctx.timer_stop() / 10
    print("  %s H2 (%dx%d) sample-major (one word per sample, 2 GiB working set): %.3f ms for 2^27 samples = %.3e syndromes/s, "
          "%.0f GB/s moved (%.1f%% of 8 TB/s); algorithmic %.2f B/sample -> %.0f GB/s"
          % (name, r2, n, ms, batch / ms * 1e3, batch * 16 / ms / 1e6, batch * 16 / ms / 1e6 / 80, (n + r2) / 8,
             batch * (n + r2) / 8 / ms / 1e6))
    e.free(), e2.free(), s.free()
    # bit-sliced: exactly n + r bits per sample; 2^31 samples so that the working set is far beyond the 256 MiB Infinity Cache
    big = 1 << 31
    bw = big // 64
    eb, sb = ctx.alloc(n * bw * 8), ctx.alloc(r2 * bw * 8)
    rng = np.random.default_rng(1)
    chunk = rng.integers(0, 2**63, 1 << 20, dtype=np.int64).view(np.uint64)
    for off in range(0, n * bw, chunk.size):                      # any bits will do for a bandwidth measurement
        _native.check(_native.lib().gf2_h2d(ctx.handle, eb.ptr + off * 8, chunk.ctypes.data, min(chunk.size, n * bw - off) * 8))
    ctx.syndrome_dev(chk, eb, big, bw, sb, bw, _native.LAYOUT_BIT_SLICED)
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        ctx.syndrome_dev(chk, eb, big, bw, sb, bw, _native.LAYOUT_BIT_SLICED)
    ms = ctx.timer_stop() / 5
    moved = (n + r2) * bw * 8
    print("  %s H2 (%dx%d) bit-sliced (%.2f GiB working set): %.3f ms for 2^31 samples = %.3e syndromes/s, %.0f GB/s "
          "= algorithmic (%.2f B/sample), %.1f%% of 8 TB/s"
          % (name, r2, n, moved / 2**30, ms, big / ms * 1e3, moved / ms / 1e6, (n + r2) / 8, moved / ms / 1e6 / 80))
    eb.free(), sb.free()
