"""The two components of a benchmark step (H1.e_z, H2.e_x) on one context (one HIP stream, back to back), on two contexts (two
streams, concurrently), and split in halves over four contexts: wall time per step over 100 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from quantum_css_codes_amd import _native

def main():
    ctxs = [_native.default_context()]
    ctxs += [_native.Context(ctxs[0].device) for _ in range(3)]
    code, h1, h2 = bench.build_code()
    chk1 = ctxs[0].check_create(h1, bench.R1, bench.N_QUBITS)
    chk2 = ctxs[0].check_create(h2, bench.R2, bench.N_QUBITS)
    batch, lde = 1 << 20, 64
    p = bench.P_TOTAL / 3
    a = ctxs[0]
    ex, ez = a.alloc(batch * 512), a.alloc(batch * 512)
    a.sample_errors_dev(bench.N_QUBITS, bench.SEED, 0, batch, p, p, p, ex, ez, lde)
    hz, hx = a.alloc((bench.R1 + 1) * 8).zero(), a.alloc((bench.R2 + 1) * 8).zero()
    a.sync()
    half = batch // 2

    class Part(object):                                      # a view of the second half of a resident buffer
        def __init__(self, buf):
            self.ptr = buf.ptr + half * 512

    def one_stream():
        a.syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, bench.R1 + 1)
        a.syndrome_sparse_dev(chk2, ex, batch, lde, None, 0, hx, bench.R2 + 1)

    def two_streams():
        ctxs[0].syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, bench.R1 + 1)
        ctxs[1].syndrome_sparse_dev(chk2, ex, batch, lde, None, 0, hx, bench.R2 + 1)

    def four_streams():
        ctxs[0].syndrome_sparse_dev(chk1, ez, half, lde, None, 0, hz, bench.R1 + 1)
        ctxs[1].syndrome_sparse_dev(chk2, ex, half, lde, None, 0, hx, bench.R2 + 1)
        ctxs[2].syndrome_sparse_dev(chk1, Part(ez), half, lde, None, 0, hz, bench.R1 + 1)
        ctxs[3].syndrome_sparse_dev(chk2, Part(ex), half, lde, None, 0, hx, bench.R2 + 1)

    for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("four streams", four_streams),
                     ("one stream", one_stream), ("two streams", two_streams), ("four streams", four_streams)):
        for _ in range(5):
            fn()
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) / 100
        print("%s: %.1f us per step = %.3e syndromes/s" % (name, dt * 1e6, batch / dt))

main()
