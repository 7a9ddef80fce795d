"""PCIe-inclusive rate of the host-buffer entry point gf2_syndrome_batch (errors and syndromes in host NumPy arrays) on the
n = 4096 check of the benchmark: never `value`, reported in DESIGN.md as the boundary's host-side figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from quantum_css_codes_amd import _native

def main():
    ctx = _native.default_context()
    code, h1, h2 = bench.build_code()
    batch = 1 << 18
    rng = np.random.default_rng(0)
    for label, density in (("sparse errors (p = 0.0067 per bit)", 0.0067), ("dense errors (p = 0.5)", 0.5)):
        e = _native.pack_rows((rng.random((batch, bench.N_QUBITS)) < density).astype(np.uint8))
        ctx.syndrome_batch(h1, bench.R1, bench.N_QUBITS, e, batch)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            s = ctx.syndrome_batch(h1, bench.R1, bench.N_QUBITS, e, batch)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("%s: %d samples, %.1f MB in + %.1f MB out, %.1f ms = %.3e syndromes/s (%.1f GB/s over the host interface)"
              % (label, batch, e.nbytes / 1e6, s.nbytes / 1e6, best * 1e3, batch / best, (e.nbytes + s.nbytes) / best / 1e9))

main()
