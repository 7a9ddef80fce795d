#!/usr/bin/env python3
"""Batches of small matrices through the wavefront-per-matrix RREF kernel (gf2_elim.hip: rref_small_kernel): 256 MiB of 64 x 512,
128 x 512 and a few other shapes, read once and written once; GB/s = 2 * bytes / time.  Ranks checked against the oracle on a few."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_css_codes_amd import _native  # noqa: E402
from oracle import c_oracle  # noqa: E402

ctx = _native.default_context()
if os.environ.get("GF2_RREF_BCAST"):
    ctx.set_option(_native.OPT_RREF_SMALL_BCAST, int(os.environ["GF2_RREF_BCAST"]))
rng = np.random.default_rng(4096)


def random_packed(m, n):
    ld = (n + 63) // 64
    words = (rng.integers(0, 2**63, (m, ld), dtype=np.int64).view(np.uint64) << np.uint64(1)) | rng.integers(0, 2, (m, ld), dtype=np.int64).view(np.uint64)
    if n % 64:
        words[:, -1] &= np.uint64((1 << (n % 64)) - 1)
    return words


for (m, n, batch) in ((64, 512, 65536), (128, 512, 32768), (256, 512, 16384), (64, 128, 262144), (64, 1024, 32768), (128, 1024, 16384),
                      (100, 300, 40000),
                      (3, 7, 1000000)):
    ld = (n + 63) // 64
    some = random_packed(1024 * m, n).reshape(1024, m, ld)
    reps = -(-batch // 1024)
    host = np.ascontiguousarray(np.tile(some, (reps, 1, 1))[:batch])
    buf = ctx.alloc(host.nbytes)
    piv, rk = ctx.alloc(batch * min(m, n) * 8), ctx.alloc(batch * 8)
    best = None
    for _ in range(4):
        buf.upload(host)
        ctx.timer_start()
        _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, m, n, ld, piv.ptr, rk.ptr))
        ms = ctx.timer_stop()
        best = ms if best is None else min(best, ms)
    got = buf.download((batch, m, ld), "<u8")
    ranks = rk.download((batch,), np.int64)
    for b in (0, 1, 513, batch - 1):
        want, _, want_rank = c_oracle.rref(host[b], m, n)
        assert int(ranks[b]) == want_rank and np.array_equal(got[b], want), (m, n, b)
    print("rref %4d x %4d x %7d: %.3f ms  %.0f GB/s (2 x %d MiB)" % (m, n, batch, best, 2 * host.nbytes / best / 1e6, host.nbytes >> 20))
    buf.free(), piv.free(), rk.free()
