import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from quantum_css_codes_amd import _native
def standard(rng, r, n, off):
    h = rng.integers(0, 2, (r, n), dtype=np.uint8)
    h[:, off:off + r] = np.identity(r, dtype=np.uint8)
    return _native.pack_rows(h)
ctx = _native.default_context()
rng = np.random.default_rng(3)
for (n, r1, r2) in ((127, 63, 63), (255, 127, 127), (511, 255, 255)):
    c1 = ctx.check_create(standard(rng, r1, n, 0), r1, n)
    c2 = ctx.check_create(standard(rng, r2, n, n - r2), r2, n)
    p = 0.01 / 3
    count = 1 << 23
    ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
    t0 = time.perf_counter()
    hz, hx = ctx.mc_run(c1, c2, 1, 0, count, p, p, p, _native.HIST_WEIGHT)
    dt = time.perf_counter() - t0
    print(os.environ.get("GF2_MC_FUSED","-"), os.environ.get("GF2_MC_UNFUSED","-"), "n=%d: %.3e samples/s" % (n, count / dt))
