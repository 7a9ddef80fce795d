"""Experiment: the two components of successive steps round-robin over N contexts (streams)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from quantum_css_codes_amd import _native
ctx = _native.default_context()
code, h1, h2 = bench.build_code()
R1, R2, N = bench.R1, bench.R2, bench.N_QUBITS
chk1, chk2 = ctx.check_create(h1, R1, N), ctx.check_create(h2, R2, N)
batch = 1 << 20
lde = _native.words_for(N)
ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
p = 0.01 / 3
ctx.sample_errors_dev(N, 1, 0, batch, p, p, p, ex, ez, lde)
hz, hx = ctx.alloc((R1 + 1) * 8).zero(), ctx.alloc((R2 + 1) * 8).zero()
ctx.sync()
for nstreams in (2, 3, 4, 6):
    ctxs = [ctx] + [_native.Context(ctx.device) for _ in range(nstreams - 1)]
    k = 0
    def step():
        global k
        ctxs[k % nstreams].syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, R1 + 1)
        ctxs[(k + 1) % nstreams].syndrome_sparse_dev(chk2, ex, batch, lde, None, 0, hx, R2 + 1)
        k += 2
    for _ in range(300):
        step()
    for c in ctxs:
        c.sync()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) / 200
        best = dt if best is None else min(best, dt)
    print("streams %d: %.4f ms per step = %.3e syndromes/s" % (nstreams, best * 1e3, batch / best))
