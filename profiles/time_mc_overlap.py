"""End-to-end Monte-Carlo at n = 4096: sampler and both components on one context; sampler of chunk k + 1 on one context while
the two components of chunk k run on two others (host sync per chunk); gf2_mc_run (the same overlap inside the library,
events instead of host syncs); gf2_mc_run with GF2_MC_FUSED=1 (sampler fused into the column-gather kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from quantum_css_codes_amd import _native

def main():
    s, z, x = _native.default_context(), None, None
    z, x = _native.Context(s.device), _native.Context(s.device)
    code, h1, h2 = bench.build_code()
    chk1 = s.check_create(h1, bench.R1, bench.N_QUBITS)
    chk2 = s.check_create(h2, bench.R2, bench.N_QUBITS)
    chunk, lde, chunks = 1 << 20, 64, 16
    p = bench.P_TOTAL / 3
    bufs = [(s.alloc(chunk * 512), s.alloc(chunk * 512)) for _ in range(2)]
    hz, hx = s.alloc((bench.R1 + 1) * 8).zero(), s.alloc((bench.R2 + 1) * 8).zero()
    s.sync()

    def sequential():
        for k in range(chunks):
            ex, ez = bufs[0]
            s.sample_errors_dev(bench.N_QUBITS, bench.SEED, k * chunk, chunk, p, p, p, ex, ez, lde)
            s.syndrome_sparse_dev(chk1, ez, chunk, lde, None, 0, hz, bench.R1 + 1)
            s.syndrome_sparse_dev(chk2, ex, chunk, lde, None, 0, hx, bench.R2 + 1)
        s.sync()

    def overlapped():
        ex, ez = bufs[0]
        s.sample_errors_dev(bench.N_QUBITS, bench.SEED, 0, chunk, p, p, p, ex, ez, lde)
        s.sync()
        for k in range(chunks):
            ex, ez = bufs[k % 2]
            z.syndrome_sparse_dev(chk1, ez, chunk, lde, None, 0, hz, bench.R1 + 1)
            x.syndrome_sparse_dev(chk2, ex, chunk, lde, None, 0, hx, bench.R2 + 1)
            if k + 1 < chunks:
                nx, nz = bufs[(k + 1) % 2]
                s.sample_errors_dev(bench.N_QUBITS, bench.SEED, (k + 1) * chunk, chunk, p, p, p, nx, nz, lde)
            s.sync(), z.sync(), x.sync()

    def library():
        s.mc_run(chk1, chk2, bench.SEED, 0, chunks * chunk, p, p, p, _native.HIST_WEIGHT)

    def fused():
        os.environ["GF2_MC_FUSED"] = "1"
        try:
            s.mc_run(chk1, chk2, bench.SEED, 0, chunks * chunk, p, p, p, _native.HIST_WEIGHT)
        finally:
            del os.environ["GF2_MC_FUSED"]

    for name, fn in (("sampler + two calls on one stream", sequential), ("sampler || two components on three streams", overlapped),
                     ("gf2_mc_run (three streams inside the library)", library), ("gf2_mc_run, GF2_MC_FUSED=1", fused)) * 2:
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print("%s: %.2f ms per 2^20 samples = %.3e syndromes/s" % (name, dt / chunks * 1e3, chunks * chunk / dt))

main()
