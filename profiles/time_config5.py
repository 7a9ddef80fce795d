"""BASELINE.json configs[4] end to end on ONE GPU: 10^8 depolarising samples of the n = 4096 code through gf2_mc_run (sampler +
both syndromes + weight histograms), and the 1/8 share a GPU takes when the job is sharded over 8 (montecarlo.shard_range)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from quantum_css_codes_amd import _native
from quantum_css_codes_amd.montecarlo import shard_range

def main():
    ctx = _native.default_context()
    code, h1, h2 = bench.build_code()
    c1, c2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
    p = bench.P_TOTAL / 3
    ctx.mc_run(c1, c2, bench.SEED, 0, 1 << 20, p, p, p, _native.HIST_WEIGHT)
    total = 10**8
    t0 = time.perf_counter()
    hz, hx = ctx.mc_run(c1, c2, bench.SEED, 0, total, p, p, p, _native.HIST_WEIGHT)
    dt = time.perf_counter() - t0
    assert int(hz.sum()) == total and int(hx.sum()) == total
    print("10^8 samples on one GPU: %.3f s = %.3e syndromes/s; zero-syndrome counts %d (Z checks), %d (X checks)"
          % (dt, total / dt, int(hz[0]), int(hx[0])))
    acc_z = np.zeros_like(hz)
    worst = 0.0
    for rank in range(8):
        first, count = shard_range(0, total, rank, 8)
        t0 = time.perf_counter()
        sz, _ = ctx.mc_run(c1, c2, bench.SEED, first, count, p, p, p, _native.HIST_WEIGHT)
        worst = max(worst, time.perf_counter() - t0)
        acc_z += sz
    assert np.array_equal(acc_z, hz), "the 8 shards do not add up to the unsharded histogram"
    print("one of 8 shards (%d samples): %.1f ms; the 8 shard histograms add up to the unsharded one" % (count, worst * 1e3))

main()
