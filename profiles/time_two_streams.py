"""The two components of a benchmark step (H1.e_z, H2.e_x) on one context (one HIP stream, back to back) against two contexts
(two streams, concurrently): wall time per step over 100 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from quantum_css_codes_amd import _native

def main():
    ctx_a = _native.default_context()
    ctx_b = _native.Context(ctx_a.device)
    code, h1, h2 = bench.build_code()
    chk1 = ctx_a.check_create(h1, bench.R1, bench.N_QUBITS)
    chk2a = ctx_a.check_create(h2, bench.R2, bench.N_QUBITS)
    chk2b = ctx_b.check_create(h2, bench.R2, bench.N_QUBITS)
    batch, lde = 1 << 20, 64
    p = bench.P_TOTAL / 3
    ex, ez = ctx_a.alloc(batch * 512), ctx_a.alloc(batch * 512)
    ctx_a.sample_errors_dev(bench.N_QUBITS, bench.SEED, 0, batch, p, p, p, ex, ez, lde)
    hz, hx = ctx_a.alloc((bench.R1 + 1) * 8).zero(), ctx_a.alloc((bench.R2 + 1) * 8).zero()
    ctx_a.sync()

    def one_stream():
        ctx_a.syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, bench.R1 + 1)
        ctx_a.syndrome_sparse_dev(chk2a, ex, batch, lde, None, 0, hx, bench.R2 + 1)

    def two_streams():
        ctx_a.syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, bench.R1 + 1)
        ctx_b.syndrome_sparse_dev(chk2b, ex, batch, lde, None, 0, hx, bench.R2 + 1)

    for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("one stream", one_stream), ("two streams", two_streams)):
        for _ in range(5):
            fn()
        ctx_a.sync(), ctx_b.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        ctx_a.sync(), ctx_b.sync()
        dt = (time.perf_counter() - t0) / 100
        print("%s: %.1f us per step = %.3e syndromes/s" % (name, dt * 1e6, batch / dt))

main()
