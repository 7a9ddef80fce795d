#!/usr/bin/env python3
"""Round 5 development check of the blocked RREF with K panels per sweep: parity against the C oracle over shapes that reach every
branch (several rounds per panel, pivot-free panels, rank-deficient, ragged chunks, small row blocks), then timings
of bench.py's three shapes under the internal options.   python profiles/r05_rref_dev.py [check] [time]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402
from quantum_css_codes_amd import _native  # noqa: E402

OPT_ROWS_WG, OPT_K = _native.OPT_RREF_ROWS_WG, _native.OPT_RREF_SWEEP_K
ctx = _native.default_context()


def mats_for(m, n, batch, seed):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        a = (rng.random((m, n)) < (0.5 if b % 3 != 1 else 0.03)).astype(np.uint8)
        if b % 4 == 2:
            a[:, :min(n, 200)] = 0
        if b % 5 == 3 and m >= 2:
            a[m // 2:] = a[: m - m // 2]
        if b % 7 == 6:
            a[:, ::2] = 0
        out.append(a)
    return out


def check():
    shapes = [(1, 1, 1), (3, 7, 2), (70, 150, 5), (300, 2500, 3), (2048, 4096, 2), (700, 4100, 5), (1500, 6000, 1), (256, 2112, 9),
              (2000, 2100, 2), (1025, 1030, 3), (3000, 1000, 2), (4100, 700, 1), (5000, 5100, 1), (520, 8200, 17), (129, 65, 33)]
    bad = 0
    for (m, n, batch) in shapes:
        mats = mats_for(m, n, batch, m * 11 + n + batch)
        want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
        for k in (4, 2, 0):
            for rows_wg in (-1, 128):
                if k == 0 and rows_wg > 0:
                    continue
                ctx.set_option(OPT_K, k)
                ctx.set_option(OPT_ROWS_WG, rows_wg)
                flags = ctx.get_flags()
                ctx.set_flags(flags | (1 << 9))                       # no wavefront-per-matrix kernel: the blocked path
                packed = np.stack([_native.pack_rows(a) for a in mats])
                pivots, ranks = ctx.rref_batch(packed, batch, m, n)
                ctx.set_flags(flags)
                ok = True
                for b in range(batch):
                    ok = ok and ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]) and \
                        list(pivots[b, :want[b][2]]) == list(want[b][1])
                print("check %5d x %5d x %3d  K=%d rows_wg=%d  %s" % (m, n, batch, k, rows_wg, "ok" if ok else "MISMATCH"), flush=True)
                bad += 0 if ok else 1
    ctx.set_option(OPT_K, -1)
    ctx.set_option(OPT_ROWS_WG, -1)
    print("check: %d mismatches" % bad)
    return bad


def timing():
    rng = np.random.default_rng(4096)

    def random_packed(m, n):
        ld = (n + 63) // 64
        return (rng.integers(0, 2**63, (m, ld), dtype=np.int64).view(np.uint64) << np.uint64(1)) | \
            rng.integers(0, 2, (m, ld), dtype=np.int64).view(np.uint64)

    for (m, n, batch) in ((2048, 4096, 1), (2048, 4096, 256), (2048, 4096, 16), (4096, 8192, 8), (1024, 2048, 64), (3000, 5000, 3)):
        mats = [random_packed(m, n) for _ in range(min(batch, 64))]
        nb = mats[0].nbytes
        buf = ctx.alloc(batch * nb)
        piv, rk = ctx.alloc(batch * min(m, n) * 8), ctx.alloc(batch * 8)
        for k in (0, 2, 4):
            ctx.set_option(OPT_K, k)
            best = None
            for _ in range(4):
                for b in range(batch):
                    _native.check(_native.lib().gf2_h2d(ctx.handle, buf.ptr + b * nb, mats[b % len(mats)].ctypes.data, nb))
                ctx.timer_start()
                _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, m, n, mats[0].shape[1], piv.ptr, rk.ptr))
                ms = ctx.timer_stop()
                best = ms if best is None else min(best, ms)
            ranks = rk.download((batch,), np.int64)
            print("time %5d x %5d x %3d  K=%d  %.3f ms  %.1f GB/s  rank %d" %
                  (m, n, batch, k, best, batch * 2 * nb / best / 1e6, int(ranks.min())), flush=True)
        buf.free(), piv.free(), rk.free()
    ctx.set_option(OPT_K, -1)


def check_tall():
    """The streamed sweeps (more than 4096 rows, K = 4 forced) against the oracle, with and without look-ahead."""
    shapes = [(4100, 700, 1), (5000, 5100, 1), (8200, 8300, 2), (9000, 2000, 1), (4500, 4600, 3)]
    bad = 0
    for (m, n, batch) in shapes:
        mats = mats_for(m, n, batch, m * 11 + n + batch)
        want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
        for flag in (0, 1 << 16):                                       # GF2_F_RREF_NO_LOOKAHEAD
            ctx.set_option(OPT_K, 4)
            flags = ctx.get_flags()
            ctx.set_flags(flags | flag)
            packed = np.stack([_native.pack_rows(a) for a in mats])
            pivots, ranks = ctx.rref_batch(packed, batch, m, n)
            ctx.set_flags(flags)
            ok = True
            for b in range(batch):
                ok = ok and ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]) and \
                    list(pivots[b, :want[b][2]]) == list(want[b][1])
            print("tall %5d x %5d x %3d  streamed sweeps, flags %#x  %s" % (m, n, batch, flag, "ok" if ok else "MISMATCH"), flush=True)
            bad += 0 if ok else 1
    ctx.set_option(OPT_K, -1)
    print("tall: %d mismatches" % bad)
    return bad


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "tall" in what:
        rc = rc or check_tall()
    if "time" in what and rc == 0:
        timing()
    sys.exit(1 if rc else 0)
