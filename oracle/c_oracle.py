"""
ORACLE -- test infrastructure only.  ctypes wrapper of oracle/gf2_oracle.c (packed-word restatement).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgf2oracle.so")
_lib = None

_i64, _u64, _p, _dbl = ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_double


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        sigs = {
            "orc_max_threads": [],
            "orc_rref": [_p, _i64, _i64, _i64, _p, _p],
            "orc_swap_columns": [_p, _i64, _i64, _i64, _i64],
            "orc_normalize": [_p, _i64, _i64, _i64, _i64, _p, _p],
            "orc_nullspace": [_p, _i64, _i64, _i64, _p, _i64, _p],
            "orc_syndrome_batch": [_p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64],
            "orc_histogram": [_p, _i64, _i64, _i64, ctypes.c_int, _p],
            "orc_sample_errors": [_i64, _u64, _i64, _i64, _dbl, _dbl, _dbl, _p, _p, _i64],
            "orc_mc_decode": [_p, _i64, _p, _i64, _i64, _p, _p, _u64, _u64, _u64, _i64, _i64, _dbl, _dbl, _dbl, _p],
            "orc_syndrome_table": [_p, _i64, _i64, _i64, _i64, _i64, _p, _p, _i64, _p, _p],
            "orc_mc_decode_wide": [_p, _i64, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _p, _i64, _p, _p, _u64, _i64, _i64, _dbl, _dbl,
                                   _dbl, _p],
            "orc_mc": [_p, _i64, _i64, _p, _i64, _i64, _i64, _u64, _i64, _i64, _dbl, _dbl, _dbl, ctypes.c_int,
                       _p, _i64, _p, _i64],
        }
        for name, args in sigs.items():
            getattr(_lib, name).argtypes = args
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def words_for(bits):
    return (int(bits) + 63) >> 6


def pack_rows(mat, ld=None):
    mat = np.asarray(mat)
    m, n = mat.shape
    width = max(1, words_for(n)) if ld is None else ld
    padded = np.zeros((m, width * 64), dtype=np.uint8)
    padded[:, :n] = (mat & 1).astype(np.uint8)
    return np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little").view("<u8").reshape(m, width))


def unpack_rows(words, n, dtype="int"):
    words = np.ascontiguousarray(words, dtype="<u8")
    m = words.shape[0]
    if m == 0 or n == 0:
        return np.zeros((m, n), dtype=dtype)
    bits = np.unpackbits(words.view(np.uint8).reshape(m, -1), axis=1, bitorder="little")
    return bits[:, :n].astype(dtype)


def rref(packed, m, n):
    a = np.array(packed, dtype="<u8", order="C")
    piv = np.zeros(max(1, min(m, n)), dtype=np.int64)
    rank = _i64(0)
    lib().orc_rref(_ptr(a), m, n, a.shape[1], _ptr(piv), ctypes.byref(rank))
    return a, piv[:rank.value], int(rank.value)


def normalize(packed, r, n, offset):
    h = np.array(packed, dtype="<u8", order="C")
    swaps = np.zeros((max(1, r), 2), dtype=np.int64)
    count = _i64(0)
    rc = lib().orc_normalize(_ptr(h), r, n, h.shape[1], offset, _ptr(swaps), ctypes.byref(count))
    return rc, h, [(int(a), int(b)) for a, b in swaps[:count.value]]


def nullspace(packed, m, n):
    a = np.ascontiguousarray(packed, dtype="<u8")
    out = np.zeros((max(1, n), max(1, words_for(n))), dtype="<u8")
    rows = _i64(0)
    lib().orc_nullspace(_ptr(a), m, n, a.shape[1], _ptr(out), out.shape[1], ctypes.byref(rows))
    return out[:rows.value]


def max_threads():
    """OpenMP threads the batch loops of the C restatement run on."""
    return int(lib().orc_max_threads())


def syndrome_batch(h, r, n, e, batch):
    h = np.ascontiguousarray(h, dtype="<u8")
    e = np.ascontiguousarray(e, dtype="<u8")
    out = np.zeros((max(1, batch), max(1, words_for(r))), dtype="<u8")
    lib().orc_syndrome_batch(_ptr(h), r, n, h.shape[1], _ptr(e), batch, e.shape[1], _ptr(out), out.shape[1])
    return out[:batch]


def histogram(s, batch, r, mode, nbins):
    s = np.ascontiguousarray(s, dtype="<u8")
    hist = np.zeros(nbins, dtype=np.uint64)
    lib().orc_histogram(_ptr(s), batch, s.shape[1], r, mode, _ptr(hist))
    return hist


def sample_errors(n, seed, first, count, p_x, p_y, p_z, lde=None):
    lde = max(1, words_for(n)) if lde is None else lde
    ex = np.zeros((max(1, count), lde), dtype="<u8")
    ez = np.zeros((max(1, count), lde), dtype="<u8")
    lib().orc_sample_errors(n, seed & 0xFFFFFFFFFFFFFFFF, first, count, p_x, p_y, p_z, _ptr(ex), _ptr(ez), lde)
    return ex[:count], ez[:count]


def mc(h1, r1, h2, r2, n, seed, first, count, p_x, p_y, p_z, mode):
    """mode 0 = full (2^r bins), 1 = weight (r+1 bins).  Returns (hist_z, hist_x)."""
    h1 = np.ascontiguousarray(h1, dtype="<u8")
    h2 = np.ascontiguousarray(h2, dtype="<u8")
    nz, nx = ((1 << r1), (1 << r2)) if mode == 0 else (r1 + 1, r2 + 1)
    hz = np.zeros(nz, dtype=np.uint64)
    hx = np.zeros(nx, dtype=np.uint64)
    lib().orc_mc(_ptr(h1), r1, h1.shape[1], _ptr(h2), r2, h2.shape[1], n, seed & 0xFFFFFFFFFFFFFFFF, first, count,
                 p_x, p_y, p_z, mode, _ptr(hz), nz, _ptr(hx), nx)
    return hz, hx


def mc_decode(h1, r1, h2, r2, n, t1, t2, xop, zop, seed, first, count, p_x, p_y, p_z):
    """h1/h2: packed rows (one word each, n <= 63); t1/t2: dense tables (2^r words, ~0 = missing)."""
    h1 = np.ascontiguousarray(np.asarray(h1, dtype="<u8").reshape(-1))
    h2 = np.ascontiguousarray(np.asarray(h2, dtype="<u8").reshape(-1))
    t1 = np.ascontiguousarray(t1, dtype=np.uint64)
    t2 = np.ascontiguousarray(t2, dtype=np.uint64)
    counts = np.zeros(5, dtype=np.uint64)
    lib().orc_mc_decode(_ptr(h1), r1, _ptr(h2), r2, n, _ptr(t1), _ptr(t2), int(xop), int(zop), seed & 0xFFFFFFFFFFFFFFFF,
                        first, count, p_x, p_y, p_z, _ptr(counts))
    return counts


def syndrome_table(h, r, n, max_weight=None):
    """css_code.syndrome_table (css_code.py:715-735) on packed rows h (r x ld), r <= 128.  Returns (t, keys, errors): keys as
    Python ints in the reference's insertion order, errors as packed rows (entries x ld)."""
    h = np.ascontiguousarray(h, dtype="<u8")
    lde = max(1, words_for(n))
    cap = 1 << 12
    while True:
        keys = np.zeros((cap, 2), dtype="<u8")
        errs = np.zeros((cap, lde), dtype="<u8")
        t, entries = _i64(), _i64()
        rc = lib().orc_syndrome_table(_ptr(h), r, n, h.shape[1], -1 if max_weight is None else max_weight, cap, _ptr(keys), _ptr(errs),
                                      lde, ctypes.byref(t), ctypes.byref(entries))
        assert rc == 0, rc
        if entries.value <= cap:
            break
        cap = int(entries.value)
    count = int(entries.value)
    return int(t.value), [int(lo) | (int(hi) << 64) for lo, hi in keys[:count].tolist()], errs[:count]


def mc_decode_wide(h1, r1, h2, r2, n, keys1, corr1, keys2, corr2, xop, zop, seed, first, count, p_x, p_y, p_z):
    """Table decode + logical tally for n <= 128: tables as (keys: Python ints, corr: packed errors, entries x ld)."""
    ld = max(1, words_for(n))
    h1 = np.ascontiguousarray(h1, dtype="<u8").reshape(r1, ld)
    h2 = np.ascontiguousarray(h2, dtype="<u8").reshape(r2, ld)

    def two(keys):
        out = np.zeros((len(keys), 2), dtype="<u8")
        for i, k in enumerate(keys):
            out[i, 0], out[i, 1] = int(k) & 0xFFFFFFFFFFFFFFFF, int(k) >> 64
        return out

    def corr2words(c):
        out = np.zeros((len(c), 2), dtype="<u8")
        c = np.asarray(c, dtype="<u8").reshape(len(c), -1)
        out[:, :c.shape[1]] = c[:, :2]
        return out
    k1, k2, c1, c2 = two(keys1), two(keys2), corr2words(corr1), corr2words(corr2)
    ops = [np.zeros(2, dtype="<u8") for _ in range(2)]
    for o, src in zip(ops, (xop, zop)):
        src = np.asarray(src, dtype="<u8").reshape(-1)
        o[:src.size] = src[:2]
    counts = np.zeros(5, dtype=np.uint64)
    rc = lib().orc_mc_decode_wide(_ptr(h1), r1, _ptr(h2), r2, n, ld, _ptr(k1), _ptr(c1), len(k1), _ptr(k2), _ptr(c2), len(k2),
                                  _ptr(ops[0]), _ptr(ops[1]), seed & 0xFFFFFFFFFFFFFFFF, first, count, p_x, p_y, p_z, _ptr(counts))
    assert rc == 0, rc
    return counts
