"""Batch size at which the LDS-slab pipeline overtakes the column-gather kernel (histogram-only, n = 4096, p = 0.01)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quantum_css_codes_amd import _native

def main():
    ctx = _native.default_context()
    n, r = 4096, 2048
    rng = np.random.default_rng(1)
    hm = rng.integers(0, 2, (r, n), dtype=np.uint8)
    hm[:, :r] = np.identity(r, dtype=np.uint8)
    chk = ctx.check_create(_native.pack_rows(hm), r, n)
    big = 1 << 20
    ex, ez = ctx.alloc(big * 512), ctx.alloc(big * 512)
    ctx.sample_errors_dev(n, 7, 0, big, 0.01 / 3, 0.01 / 3, 0.01 / 3, ex, ez, 64)
    hist = ctx.alloc((r + 1) * 8).zero()
    for batch in (4096, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
        out = []
        for switch in ("GF2_SPARSE_SLABS", "GF2_SPARSE_GATHER"):
            os.environ[switch] = "1"
            for _ in range(3):
                ctx.syndrome_sparse_dev(chk, ex, batch, 64, None, 0, hist, r + 1)
            ctx.sync()
            ctx.timer_start()
            for _ in range(20):
                ctx.syndrome_sparse_dev(chk, ex, batch, 64, None, 0, hist, r + 1)
            out.append(ctx.timer_stop() / 20 * 1e3)
            del os.environ[switch]
        print("batch %8d: slab pipeline %8.1f us, column gather %8.1f us" % (batch, out[0], out[1]))

main()
