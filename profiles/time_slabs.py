"""Times the LDS row-slab sparse pipeline (gf2_slabs.hip) on the two check shapes of the benchmark; GF2_SPARSE_GATHER=1 times
the column-gather kernel instead.  Used under rocprofv3 --kernel-trace for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quantum_css_codes_amd import _native

def main():
    ctx = _native.default_context()
    n, r, batch = 4096, 2048, 1 << int(os.environ.get("SLAB_LOG2_BATCH", "20"))
    rng = np.random.default_rng(1)
    results = []
    for (rr, ioff) in ((2048, 0), (2047, 2049)):
        hm = rng.integers(0, 2, (rr, n), dtype=np.uint8)
        hm[:, ioff:ioff + rr] = np.identity(rr, dtype=np.uint8)
        chk = ctx.check_create(_native.pack_rows(hm), rr, n)
        ex, ez = ctx.alloc(batch * 512), ctx.alloc(batch * 512)
        ctx.sample_errors_dev(n, 7, 0, batch, 0.01 / 3, 0.01 / 3, 0.01 / 3, ex, ez, 64)
        hist = ctx.alloc((rr + 1) * 8).zero()
        for _ in range(3):
            ctx.syndrome_sparse_dev(chk, ex, batch, 64, None, 0, hist, rr + 1)
        ctx.sync()
        ctx.timer_start()
        for _ in range(20):
            ctx.syndrome_sparse_dev(chk, ex, batch, 64, None, 0, hist, rr + 1)
        ms = ctx.timer_stop() / 20
        results.append("r=%d off=%d: %.4f ms/launch of 2^%d (%.0f GB/s)" % (rr, ioff, ms, batch.bit_length() - 1, batch * 512 / ms / 1e6))
    print(("gather kernel: " if os.environ.get("GF2_SPARSE_GATHER") else "slab pipeline: ") + " | ".join(results))

main()
