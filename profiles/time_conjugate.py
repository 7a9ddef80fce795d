"""Times transform_stabilisers (gf2_conjugate_gates) on the encoders of the n=4096 benchmark code."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from quantum_css_codes_amd import _native, css_code

def main():
    ctx = _native.default_context()
    code, _, _ = bench.build_code()
    n = code.n
    for name, prog in (("noisy_encode_zero", code.noisy_encode_zero(range(n))), ("noisy_encode_plus", code.noisy_encode_plus(range(n)))):
        mat = np.concatenate((np.zeros((n, n), dtype=np.uint8), np.identity(n, dtype=np.uint8)), axis=1)
        packed = _native.pack_rows(mat)
        ctx.conjugate_gates(packed.copy(), n, n, prog)
        t0 = time.perf_counter()
        rc, stop = ctx.conjugate_gates(packed, n, n, prog)
        dt = time.perf_counter() - t0
        print("%s: %d gates on a %d x %d stabiliser matrix: %.1f ms through the C ABI (host matrix in and out), "
              "%.1f ns per gate; rc=%d" % (name, len(prog), n, 2 * n, dt * 1e3, dt / len(prog) * 1e9, rc))

main()
