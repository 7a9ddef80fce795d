"""
ctypes binding of libgf2hip.so (include/gf2hip.h) and the packed-word helpers the host side uses.

There is no CPU fallback.  If the shared library is missing, or no gfx950 device is usable, the
first compute call raises GF2Error -- the product never routes through oracle/ or NumPy arithmetic.
"""
import contextlib
import ctypes
import math
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgf2hip.so")

GF2_OK, GF2_E_ARG, GF2_E_COLUMNS, GF2_E_DEPENDENT, GF2_E_HIP, GF2_E_NOMEM, GF2_E_NOTCSS, GF2_E_RCCL = 0, -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128
LAYOUT_SAMPLE_MAJOR, LAYOUT_BIT_SLICED, LAYOUT_TILED = 0, 1, 2
HIST_FULL, HIST_WEIGHT = 0, 1
K_SYNDROME, K_HIST, K_SAMPLER, K_ELIM = 0, 1, 2, 3
# routing flags of a context and its tunables: the few a caller needs are in include/gf2hip.h (F_MC_DENSE, F_RREF_SEQUENTIAL,
# F_NORMALIZE_SEQUENTIAL, OPT_SLAB_PASS_LOG2, OPT_MC_CHUNK_LOG2), the rest -- routes for the parity tests and the A/B scripts -- in
# csrc/gf2_tuning.h
(F_SPARSE_GATHER, F_SPARSE_SLABS, F_NO_REDO, F_GATHER_GENERIC, F_MC_UNFUSED, F_MC_DENSE, F_MC_FUSED, F_MC_PIPELINE,
 F_RREF_SEQUENTIAL, F_RREF_NO_SMALL, F_NORMALIZE_SEQUENTIAL, F_SAMPLER_GENERIC, F_DIAG_CLOCKS,
 F_DIAG_MC_TIMES, F_MC_ROWS, F_COMBINE_FOLDED, F_RREF_NO_LOOKAHEAD, F_RREF_LOOKAHEAD, F_COMBINE_SEPARATE) = (1 << k for k in range(19))
(OPT_SLAB_PASS_LOG2, OPT_COMBINE_BLOCKS, OPT_GATHER_REVERSE, OPT_REDO_BLOCKS_PER_CU, OPT_MC_CHUNK_LOG2, OPT_COMBINE_THREADS,
 OPT_GATHER_CROSS, OPT_GATHER_OVER, OPT_RREF_SMALL_BCAST, OPT_MC_SAMPLER_WAVES,
 OPT_MC_TAIL_CAP, OPT_RREF_STREAM_VARIANT, OPT_RREF_ROWS_WG, OPT_RREF_SWEEP_K) = range(14)


class GF2Error(RuntimeError):
    """A libgf2hip.so call failed (code and message from gf2_last_error)."""

    def __init__(self, code, message):
        super().__init__("libgf2hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


_c_i64 = ctypes.c_int64
_c_u64 = ctypes.c_uint64
_p = ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> argument types; every function returns int unless listed in _RESTYPES.  This table is the
# Python statement of include/gf2hip.h; tests/test_abi.py checks the two against each other.
SIGNATURES = {
    "gf2_version": [],
    "gf2_last_error": [],
    "gf2_device_count": [ctypes.POINTER(ctypes.c_int)],
    "gf2_ctx_create": [ctypes.c_int, _pp],
    "gf2_ctx_destroy": [_p],
    "gf2_ctx_sync": [_p],
    "gf2_ctx_set_flags": [_p, ctypes.c_uint32],
    "gf2_ctx_get_flags": [_p, ctypes.POINTER(ctypes.c_uint32)],
    "gf2_ctx_set_option": [_p, ctypes.c_int, _c_i64],
    "gf2_dev_alloc": [_p, ctypes.c_size_t, _pp],
    "gf2_dev_free": [_p, _p],
    "gf2_dev_zero": [_p, _p, ctypes.c_size_t],
    "gf2_h2d": [_p, _p, _p, ctypes.c_size_t],
    "gf2_d2h": [_p, _p, _p, ctypes.c_size_t],
    "gf2_timer_start": [_p],
    "gf2_timer_stop": [_p, ctypes.POINTER(ctypes.c_float)],
    "gf2_profile_enable": [_p, ctypes.c_int],
    "gf2_profile_reset": [_p],
    "gf2_profile_get": [_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_i64)],
    "gf2_membw_probe_dev": [_p, _p, _p, ctypes.c_size_t, _p],
    "gf2_pack_rows_u8": [_p, _c_i64, _c_i64, _c_i64, _p, _c_i64],
    "gf2_pack_rows_i64": [_p, _c_i64, _c_i64, _c_i64, _p, _c_i64],
    "gf2_pack_rows_binary_u8": [_p, _c_i64, _c_i64, _c_i64, _p, _c_i64, ctypes.POINTER(ctypes.c_int)],
    "gf2_pack_rows_binary_i64": [_p, _c_i64, _c_i64, _c_i64, _p, _c_i64, ctypes.POINTER(ctypes.c_int)],
    "gf2_unpack_rows_u8": [_p, _c_i64, _c_i64, _c_i64, _p, _c_i64],
    "gf2_unpack_rows_i64": [_p, _c_i64, _c_i64, _c_i64, _p, _c_i64],
    "gf2_rref": [_p, _p, _c_i64, _c_i64, _c_i64, _p, _p],
    "gf2_rref_batch": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p, _p],
    "gf2_rref_batch_dev": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p, _p],
    "gf2_normalize_dev": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p, _p, _p],
    "gf2_nullspace": [_p, _p, _c_i64, _c_i64, _c_i64, _p, _c_i64, _p],
    "gf2_normalize": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p, _p],
    "gf2_swap_columns": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64],
    "gf2_matmul_abt": [_p, _p, _c_i64, _c_i64, _p, _c_i64, _c_i64, _c_i64, _p, _c_i64],
    "gf2_row_weights": [_p, _p, _c_i64, _c_i64, _c_i64, _p],
    "gf2_conjugate_gates": [_p, _p, _c_i64, _c_i64, _c_i64, _p, _c_i64, ctypes.POINTER(_c_i64)],
    "gf2_syndrome_table": [_p, _p, _c_i64, _c_i64, _c_i64, _p, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)],
    "gf2_syndrome_table_wide": [_p, _p, _c_i64, _c_i64, _c_i64, _p, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)],
    "gf2_syndrome_table_cols": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)],
    "gf2_syndrome_table_hashed": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p, _p, _c_i64, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)],
    "gf2_check_create": [_p, _p, _c_i64, _c_i64, _c_i64, _pp],
    "gf2_check_destroy": [_p, _p],
    "gf2_syndrome_batch": [_p, _p, _c_i64, _c_i64, _c_i64, _p, _c_i64, _c_i64, ctypes.c_int, _p, _c_i64],
    "gf2_syndrome_dev": [_p, _p, _p, _c_i64, _c_i64, ctypes.c_int, _p, _c_i64],
    "gf2_syndrome_sparse_dev": [_p, _p, _p, _c_i64, _c_i64, _p, _c_i64, _p, _c_i64],
    "gf2_histogram_dev": [_p, _p, _c_i64, _c_i64, ctypes.c_int, _c_i64, ctypes.c_int, _p, _c_i64],
    "gf2_sample_errors_dev": [_p, _c_i64, _c_u64, _c_i64, _c_i64, ctypes.c_double, ctypes.c_double,
                              ctypes.c_double, _p, _p, _c_i64, ctypes.c_int],
    "gf2_tiled_ld": [_c_i64],
    "gf2_tiled_words": [_c_i64, _c_i64],
    "gf2_retile_dev": [_p, _p, _c_i64, _c_i64, _c_i64, _p],
    "gf2_mc_decode": [_p, _p, _p, _p, _p, _c_u64, _c_u64, _c_u64, _c_i64, _c_i64, ctypes.c_double, ctypes.c_double,
                      ctypes.c_double, _p],
    "gf2_mc_decode_hashed": [_p, _c_i64, _c_i64, _p, _c_i64, _p, _p, _c_i64, _p, _c_i64, _p, _p, _c_i64, _p, _p, _c_u64, _c_i64, _c_i64,
                             ctypes.c_double, ctypes.c_double, ctypes.c_double, _p],
    "gf2_mc_run": [_p, _p, _p, _c_u64, _c_i64, _c_i64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                   ctypes.c_int, _p, _c_i64, _p, _c_i64],
    "gf2_comm_unique_id": [_p, ctypes.c_size_t],
    "gf2_comm_create": [_p, _p, ctypes.c_int, ctypes.c_int, _pp],
    "gf2_comm_create_all": [_pp, ctypes.c_int, _pp],
    "gf2_comm_size": [_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "gf2_comm_destroy": [_p],
    "gf2_hist_allreduce": [_p, _pp, _c_i64],
    "gf2_rccl_version": [ctypes.POINTER(ctypes.c_int)],
}
_RESTYPES = {"gf2_last_error": ctypes.c_char_p, "gf2_tiled_ld": _c_i64, "gf2_tiled_words": _c_i64}

_lib = None
_lib_lock = threading.Lock()


def lib():
    """The loaded library.  Raises GF2Error when it has not been built."""
    global _lib
    if _lib is None:
        with _lib_lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise GF2Error(GF2_E_HIP, "%s not found: build it with `python -c 'import "
                                   "__graft_entry__ as g; g.build()'` (there is no CPU fallback)" % LIB_PATH)
                handle = ctypes.CDLL(LIB_PATH)
                for name, argtypes in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.argtypes = argtypes
                    fn.restype = _RESTYPES.get(name, ctypes.c_int)
                _lib = handle
    return _lib


def check(code):
    if code != GF2_OK:
        raise GF2Error(code, lib().gf2_last_error().decode("utf-8", "replace"))


def device_count():
    count = ctypes.c_int(0)
    rc = lib().gf2_device_count(ctypes.byref(count))
    return count.value if rc == GF2_OK else 0


# ---- packed words ---------------------------------------------------------------------------------------

def words_for(bits):
    return (int(bits) + 63) >> 6


def pack_rows(mat, ld=None):
    """Dense 2-D integer array -> packed uint64 rows (column j at word j>>6, bit j&63).  Entries are
    reduced with `& 1`, which is what the reference's lazy np.mod(., 2) amounts to for integers."""
    mat = np.asarray(mat)
    if mat.ndim != 2:
        raise ValueError("expected a 2-D array")
    m, n = mat.shape
    width = max(1, words_for(n)) if ld is None else int(ld)
    if m and n and mat.flags.c_contiguous and mat.dtype in (np.int64, np.uint8):
        # one pass in C (gf2_pack_rows_*): 2.5x faster than the NumPy route below on a 2048 x 4096 int64 array
        out = np.zeros((m, width), dtype="<u8")
        fn = lib().gf2_pack_rows_i64 if mat.dtype == np.int64 else lib().gf2_pack_rows_u8
        check(fn(_ptr(mat), m, n, n, _ptr(out), width))
        return out
    if mat.dtype == np.bool_:
        bits = mat.astype(np.uint8)
    elif np.issubdtype(mat.dtype, np.integer):
        bits = (mat & 1).astype(np.uint8)
    else:
        bits = np.mod(mat, 2).astype(np.uint8)
    padded = np.zeros((m, width * 64), dtype=np.uint8)
    padded[:, :n] = bits
    return np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little").view("<u8").reshape(m, width))


def pack_rows_binary(mat):
    """(packed rows, every entry is 0 or 1) -- the test of css_code.py:39-44 made while packing: for C-contiguous int64 / uint8
    arrays one threaded pass in C; otherwise the reference's own expression."""
    mat = np.asarray(mat)
    if mat.ndim != 2:
        raise ValueError("expected a 2-D array")
    m, n = mat.shape
    if m and n and mat.flags.c_contiguous and mat.dtype in (np.int64, np.uint8):
        out = np.zeros((m, max(1, words_for(n))), dtype="<u8")
        other = ctypes.c_int(0)
        fn = lib().gf2_pack_rows_binary_i64 if mat.dtype == np.int64 else lib().gf2_pack_rows_binary_u8
        check(fn(_ptr(mat), m, n, n, _ptr(out), out.shape[1], ctypes.byref(other)))
        return out, other.value == 0
    reduced = np.mod(np.array(mat, dtype='int'), 2)
    return pack_rows(reduced), bool(np.array_equal(reduced, mat))


def unpack_rows(words, n, dtype="int"):
    """Packed uint64 rows -> dense m x n array of `dtype`."""
    words = np.ascontiguousarray(words, dtype="<u8")
    m = words.shape[0]
    if m == 0 or n == 0:
        return np.zeros((m, n), dtype=dtype)
    if np.dtype(dtype) in (np.dtype(np.int64), np.dtype(np.uint8)):
        out = np.empty((m, n), dtype=dtype)
        fn = lib().gf2_unpack_rows_i64 if np.dtype(dtype) == np.dtype(np.int64) else lib().gf2_unpack_rows_u8
        check(fn(_ptr(words), m, n, words.shape[1], _ptr(out), n))
        return out
    bits = np.unpackbits(words.view(np.uint8).reshape(m, -1), axis=1, bitorder="little")
    return bits[:, :n].astype(dtype)


def unpack_rows_into(words, out):
    """Packed uint64 rows -> the existing dense array `out` (m x n), in place; one pass in C when `out` is a C-contiguous
    int64 or uint8 array."""
    m, n = out.shape
    words = np.ascontiguousarray(words, dtype="<u8")
    if m and n and out.flags.c_contiguous and out.dtype in (np.int64, np.uint8):
        fn = lib().gf2_unpack_rows_i64 if out.dtype == np.int64 else lib().gf2_unpack_rows_u8
        check(fn(_ptr(words), m, n, words.shape[1], _ptr(out), n))
    else:
        out[...] = unpack_rows(words, n, dtype=out.dtype)
    return out


def unrank_supports(ranks, n, w):
    """Supports (count x w, ascending positions) of the weight-w errors with the given ranks in the combinatorial number system:
    positions c_1 < ... < c_w have rank C(c_1, 1) + ... + C(c_w, w) (the order gf2_syndrome_table_wide enumerates in)."""
    import math
    ranks = np.array(ranks, dtype=np.uint64)
    out = np.zeros((ranks.size, w), dtype=np.int64)
    for k in range(w, 0, -1):
        column = np.array([min(math.comb(c, k), 1 << 63) for c in range(n + 1)], dtype=np.uint64)   # C(c, k), ascending in c
        c = np.searchsorted(column, ranks, side='right').astype(np.int64) - 1                        # largest c with C(c, k) <= rank
        out[:, k - 1] = c
        ranks = ranks - column[c]
    return out


def tiled_ld(n):
    return int(lib().gf2_tiled_ld(int(n)))


def tiled_words(n, batch):
    return int(lib().gf2_tiled_words(int(n), int(batch)))


def tile_rows(packed, n):
    """Host model of GF2_LAYOUT_TILED: sample-major packed rows (B x ld) -> flat tiled words."""
    packed = np.ascontiguousarray(packed, dtype="<u8")
    batch, ld = packed.shape
    ldt = tiled_ld(n)
    tiles = max(1, (batch + 63) // 64)
    full = np.zeros((tiles * 64, ldt), dtype="<u8")
    full[:batch, :min(ld, ldt)] = packed[:, :min(ld, ldt)]
    # (tile, lane, pair, half) -> (tile, pair, lane, half)
    return np.ascontiguousarray(full.reshape(tiles, 64, ldt // 2, 2).transpose(0, 2, 1, 3)).reshape(-1)


def untile_rows(flat, n, batch):
    flat = np.ascontiguousarray(flat, dtype="<u8")
    ldt = tiled_ld(n)
    tiles = flat.size // (64 * ldt)
    rows = flat.reshape(tiles, ldt // 2, 64, 2).transpose(0, 2, 1, 3).reshape(tiles * 64, ldt)
    return np.ascontiguousarray(rows[:batch])


def _ptr(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)


# ---- context ----------------------------------------------------------------------------------------------

class DeviceBuffer(object):
    """Device memory owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        out = ctypes.c_void_p()
        check(lib().gf2_dev_alloc(ctx.handle, self.nbytes, ctypes.byref(out)))
        self.ptr = out.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("upload larger than the buffer")
        check(lib().gf2_h2d(self.ctx.handle, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than the buffer")
        check(lib().gf2_d2h(self.ctx.handle, _ptr(out), self.ptr, out.nbytes))
        return out

    def zero(self):
        check(lib().gf2_dev_zero(self.ctx.handle, self.ptr, self.nbytes))
        return self

    def view(self, offset, nbytes):
        """A window of this buffer (same upload / download / zero interface; the parent keeps the memory)."""
        if offset < 0 or nbytes < 0 or offset + nbytes > self.nbytes:
            raise ValueError("view outside the buffer")
        return DeviceView(self, int(offset), int(nbytes))

    def free(self):
        if self.ptr:
            check(lib().gf2_dev_free(self.ctx.handle, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                lib().gf2_dev_free(self.ctx.handle, self.ptr)
        except Exception:
            pass


class DeviceView(DeviceBuffer):
    """Part of a DeviceBuffer: owns nothing."""

    def __init__(self, parent, offset, nbytes):
        self.ctx, self.parent, self.nbytes = parent.ctx, parent, nbytes
        self.ptr = parent.ptr + offset

    def free(self):
        self.ptr = None

    def __del__(self):
        pass


class Check(object):
    """A parity-check matrix prepared on the device (gf2_check_create)."""

    def __init__(self, ctx, packed, r, n):
        self.ctx = ctx
        self.r, self.n = int(r), int(n)
        packed = np.ascontiguousarray(packed, dtype="<u8")
        out = ctypes.c_void_p()
        ld = packed.shape[1] if packed.ndim == 2 and packed.shape[0] else max(1, words_for(n))
        check(lib().gf2_check_create(ctx.handle, _ptr(packed), self.r, self.n, ld, ctypes.byref(out)))
        self.handle = out.value

    @property
    def slabs(self):
        return max(1, words_for(self.r))

    def free(self):
        if self.handle:
            check(lib().gf2_check_destroy(self.ctx.handle, self.handle))
            self.handle = None

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                lib().gf2_check_destroy(self.ctx.handle, self.handle)
        except Exception:
            pass


class Context(object):
    """One HIP stream on one MI355X (gf2_ctx)."""

    def __init__(self, device=0):
        out = ctypes.c_void_p()
        check(lib().gf2_ctx_create(int(device), ctypes.byref(out)))
        self.handle = out.value
        self.device = int(device)

    def close(self):
        if self.handle:
            lib().gf2_ctx_destroy(self.handle)
            self.handle = None

    def sync(self):
        check(lib().gf2_ctx_sync(self.handle))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # -- routing ----------------------------------------------------------------------------------------
    def get_flags(self):
        out = ctypes.c_uint32(0)
        check(lib().gf2_ctx_get_flags(self.handle, ctypes.byref(out)))
        return int(out.value)

    def set_flags(self, flags):
        check(lib().gf2_ctx_set_flags(self.handle, int(flags)))

    @contextlib.contextmanager
    def flags(self, extra):
        """`with ctx.flags(F_SPARSE_SLABS): ...` forces a route (GF2_F_*) for the calls inside."""
        before = self.get_flags()
        self.set_flags(before | int(extra))
        try:
            yield self
        finally:
            self.set_flags(before)

    def set_option(self, option, value):
        check(lib().gf2_ctx_set_option(self.handle, int(option), -1 if value is None else int(value)))

    # -- timing ---------------------------------------------------------------------------------------
    def timer_start(self):
        check(lib().gf2_timer_start(self.handle))

    def timer_stop(self):
        ms = ctypes.c_float(0)
        check(lib().gf2_timer_stop(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def profile(self, on):
        check(lib().gf2_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        check(lib().gf2_profile_reset(self.handle))

    def profile_get(self, family):
        ms, count = ctypes.c_double(0), _c_i64(0)
        check(lib().gf2_profile_get(self.handle, int(family), ctypes.byref(ms), ctypes.byref(count)))
        return float(ms.value), int(count.value)

    def membw_probe(self, src_buf, nbytes, dst_buf=None, reps=5):
        """GB/s of a plain streaming read (or copy) of `nbytes` of src_buf: the measured ceiling beside the 8 TB/s spec."""
        sink = self.alloc(8).zero()
        call = lambda: check(lib().gf2_membw_probe_dev(self.handle, src_buf.ptr, dst_buf.ptr if dst_buf is not None else None,
                                                       int(nbytes), sink.ptr))
        call()
        self.sync()
        self.timer_start()
        for _ in range(reps):
            call()
        ms = self.timer_stop() / reps
        sink.free()
        return (2 if dst_buf is not None else 1) * nbytes / ms / 1e6

    # -- linear algebra on packed host arrays ------------------------------------------------------------
    def rref(self, packed, m, n):
        """In place.  Returns (pivot columns, rank)."""
        cap = max(1, min(m, n))
        pivots = np.zeros(cap, dtype=np.int64)
        rank = _c_i64(0)
        check(lib().gf2_rref(self.handle, _ptr(packed), m, n, packed.shape[1] if m else max(1, words_for(n)),
                             _ptr(pivots), ctypes.byref(rank)))
        return pivots[:rank.value], int(rank.value)

    def rref_batch(self, packed, batch, m, n):
        cap = max(1, min(m, n))
        pivots = np.zeros((batch, cap), dtype=np.int64)
        ranks = np.zeros(max(1, batch), dtype=np.int64)
        check(lib().gf2_rref_batch(self.handle, _ptr(packed), batch, m, n, packed.shape[-1], _ptr(pivots),
                                   _ptr(ranks)))
        return pivots, ranks[:batch]

    def nullspace(self, packed, m, n):
        ld = packed.shape[1] if m else max(1, words_for(n))
        out = np.zeros((max(1, n), max(1, words_for(n))), dtype="<u8")
        rows = _c_i64(0)
        check(lib().gf2_nullspace(self.handle, _ptr(packed), m, n, ld, _ptr(out), out.shape[1], ctypes.byref(rows)))
        return out[:rows.value]

    def normalize(self, packed, r, n, offset):
        """In place.  Returns the list of (column, column) swaps."""
        swaps = np.zeros((max(1, r), 2), dtype=np.int64)
        count = _c_i64(0)
        ld = packed.shape[1] if r else max(1, words_for(n))
        check(lib().gf2_normalize(self.handle, _ptr(packed), r, n, ld, offset, _ptr(swaps), ctypes.byref(count)))
        return [(int(a), int(b)) for a, b in swaps[:count.value]]

    def swap_columns(self, packed, m, n, i, j):
        ld = packed.shape[1] if m else max(1, words_for(n))
        check(lib().gf2_swap_columns(self.handle, _ptr(packed), m, n, ld, int(i), int(j)))

    def matmul_abt(self, a, ra, b, rb, n):
        out = np.zeros((max(1, ra), max(1, words_for(rb))), dtype="<u8")
        if ra and rb:
            check(lib().gf2_matmul_abt(self.handle, _ptr(a), ra, a.shape[1], _ptr(b), rb, b.shape[1], n, _ptr(out),
                                       out.shape[1]))
        return out[:ra]

    def row_weights(self, packed, m, n):
        out = np.zeros(max(1, m), dtype=np.uint32)
        ld = packed.shape[1] if m else max(1, words_for(n))
        check(lib().gf2_row_weights(self.handle, _ptr(packed), m, n, ld, _ptr(out)))
        return out[:m]

    def conjugate_gates(self, packed, k, n, gates):
        """css_code.transform_stabilisers on a packed k x 2n matrix, in place.  gates: (g, 3) int32 rows (kind, a, b).
        Returns (code, stop): (GF2_OK, -1), or (GF2_E_NOTCSS | GF2_E_ARG, index of the gate that stopped the walk) with
        `packed` holding the result of the gates before it."""
        gates = np.ascontiguousarray(gates, dtype=np.int32).reshape(-1, 3)
        stop = _c_i64(-1)
        ld = packed.shape[1] if k else max(1, words_for(2 * n))
        rc = lib().gf2_conjugate_gates(self.handle, _ptr(packed) if k else None, k, n, ld,
                                       _ptr(gates) if len(gates) else None, len(gates), ctypes.byref(stop))
        if rc not in (GF2_OK, GF2_E_NOTCSS, GF2_E_ARG) or (rc == GF2_E_ARG and stop.value < 0):
            check(rc)
        return rc, int(stop.value)

    TABLE_MAX_N, TABLE_MAX_R = 64, 24
    TABLE_EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)

    def syndrome_table(self, packed, r, n, max_weight=None):
        """css_code.syndrome_table on the device (n <= 64, r <= 24): returns (t, dense) where dense[key] is the packed
        error with syndrome key = vec_to_int(syndrome), or TABLE_EMPTY."""
        rows = np.ascontiguousarray(packed[:, 0] if r else np.zeros(0, dtype="<u8"), dtype="<u8")
        dense = np.empty(1 << r, dtype="<u8")
        t, entries = _c_i64(), _c_i64()
        check(lib().gf2_syndrome_table(self.handle, _ptr(rows) if r else None, r, n,
                                       -1 if max_weight is None else max_weight, _ptr(dense), ctypes.byref(t),
                                       ctypes.byref(entries)))
        return int(t.value), dense

    TABLE_WIDE_MAX_N = 128

    def syndrome_table_wide(self, packed, r, n, max_weight=None):
        """The same for 64 < n <= 128: dense[key] = (weight << 32) | rank of the error inside its weight class (combinatorial
        number system), or TABLE_EMPTY; unrank with unrank_supports."""
        rows = np.zeros((max(1, r), 2), dtype="<u8")
        rows[:r, :packed.shape[1]] = packed[:r, :2]
        dense = np.empty(1 << r, dtype="<u8")
        t, entries = _c_i64(), _c_i64()
        check(lib().gf2_syndrome_table_wide(self.handle, _ptr(rows) if r else None, r, n,
                                            -1 if max_weight is None else max_weight, _ptr(dense), ctypes.byref(t),
                                            ctypes.byref(entries)))
        return int(t.value), dense

    TABLE_COLS_MAX_N = 8192

    def syndrome_table_cols(self, packed, r, n, max_weight=None):
        """The same for 128 < n <= 8192 (errors as position lists on the device): dense[key] = (weight << 32) | rank, or
        TABLE_EMPTY; unrank with unrank_supports."""
        rows = np.ascontiguousarray(packed, dtype="<u8") if r else np.zeros((1, max(1, words_for(n))), dtype="<u8")
        dense = np.empty(1 << r, dtype="<u8")
        t, entries = _c_i64(), _c_i64()
        check(lib().gf2_syndrome_table_cols(self.handle, _ptr(rows) if r else None, r, n, rows.shape[1],
                                            -1 if max_weight is None else max_weight, _ptr(dense), ctypes.byref(t),
                                            ctypes.byref(entries)))
        return int(t.value), dense

    TABLE_HASH_MAX_R, TABLE_HASH_MAX_N = 127, 8192

    def syndrome_table_hashed(self, packed, r, n, max_weight=None):
        """css_code.syndrome_table beyond 24 checks (1 <= r <= 127, n <= 8192): a hash table on the device.  Returns
        (t, keys, weights, ranks): keys as a uint64 array (r <= 63) or a list of Python ints (two-word keys), the entries'
        weights and their ranks inside their weight class (unrank with unrank_supports), in no particular order."""
        rows = np.ascontiguousarray(packed, dtype="<u8")
        kw = 1 if r <= 63 else 2
        # The C call writes the entries only into a buffer that holds them all (it counts the classes, it does not keep them), so
        # a buffer that is too small costs a second search.  With a weight bound the table cannot exceed the classes up to it; without
        # one, 2^22 entries (100 MB on the host) cover every table met so far -- the first collision of a code with r checks comes
        # after some 2^((r+1)/2) errors -- and only larger tables are searched twice.
        cap = 1 << 22
        if max_weight is not None:
            total, w = 0, 0
            while w <= min(max_weight, n) and total <= (1 << 28):
                total += math.comb(n, w)
                w += 1
            cap = max(1, min(total, 1 << 28))
        t, entries = _c_i64(), _c_i64()
        while True:
            keys = np.empty((cap, kw), dtype="<u8")
            vals = np.empty(cap, dtype="<u8")
            check(lib().gf2_syndrome_table_hashed(self.handle, _ptr(rows), r, n, rows.shape[1], -1 if max_weight is None else max_weight,
                                                  _ptr(keys), _ptr(vals), cap, ctypes.byref(t), ctypes.byref(entries)))
            if entries.value <= cap:
                break
            cap = int(entries.value)                    # (the search runs once more: the classes were counted, not kept)
        count = int(entries.value)
        keys, vals = keys[:count], vals[:count]
        weights = (vals >> np.uint64(32)).astype(np.int64)
        ranks = vals & np.uint64(0xFFFFFFFF)
        if kw == 1:
            return int(t.value), keys[:, 0].copy(), weights, ranks
        return int(t.value), [int(lo) | (int(hi) << 64) for lo, hi in keys.tolist()], weights, ranks

    def mc_decode_hashed(self, n, h1, r1, keys1, corr1, h2, r2, keys2, corr2, x_operator, z_operator, seed, first, count,
                         p_x, p_y, p_z):
        """gf2_mc_decode_hashed: h1 / h2 packed rows (ld words), keys (entries x 1 or 2 words), corr (entries x 2 words),
        operators (2 words).  Returns the five counts."""
        h1, h2 = np.ascontiguousarray(h1, dtype="<u8"), np.ascontiguousarray(h2, dtype="<u8")
        arrays = [np.ascontiguousarray(a, dtype="<u8") for a in (keys1, corr1, keys2, corr2, x_operator, z_operator)]
        counts = np.zeros(5, dtype=np.uint64)
        check(lib().gf2_mc_decode_hashed(self.handle, n, h1.shape[1], _ptr(h1), r1, _ptr(arrays[0]), _ptr(arrays[1]), len(arrays[1]),
                                         _ptr(h2), r2, _ptr(arrays[2]), _ptr(arrays[3]), len(arrays[3]), _ptr(arrays[4]),
                                         _ptr(arrays[5]), seed & 0xFFFFFFFFFFFFFFFF, first, count, p_x, p_y, p_z, _ptr(counts)))
        return counts

    # -- syndromes ----------------------------------------------------------------------------------------
    def check_create(self, packed, r, n):
        return Check(self, packed, r, n)

    def syndrome_batch(self, h, r, n, e, batch):
        """Sample-major host arrays: e is batch x words(n); returns batch x words(r)."""
        out = np.zeros((max(1, batch), max(1, words_for(r))), dtype="<u8")
        if batch and r:
            check(lib().gf2_syndrome_batch(self.handle, _ptr(h), r, n, h.shape[1], _ptr(e), batch, e.shape[1],
                                           LAYOUT_SAMPLE_MAJOR, _ptr(out), out.shape[1]))
        return out[:batch]

    def syndrome_batch_sliced(self, h, r, n, e, batch):
        """Bit-sliced host arrays: e is n x words(batch); returns r x words(batch)."""
        width = max(1, words_for(batch))
        out = np.zeros((max(1, r), width), dtype="<u8")
        if batch and r:
            check(lib().gf2_syndrome_batch(self.handle, _ptr(h), r, n, h.shape[1], _ptr(e), batch, e.shape[1],
                                           LAYOUT_BIT_SLICED, _ptr(out), width))
        return out[:r]

    def syndrome_dev(self, chk, e_buf, batch, lde, s_buf, lds, layout=LAYOUT_SAMPLE_MAJOR):
        check(lib().gf2_syndrome_dev(self.handle, chk.handle, e_buf.ptr, batch, lde, layout, s_buf.ptr, lds))

    def syndrome_sparse_dev(self, chk, e_buf, batch, lde, s_buf=None, lds=0, hist_buf=None, nbins=0):
        """Sparse-error kernel: sample-major errors; writes syndromes and/or accumulates the weight histogram."""
        check(lib().gf2_syndrome_sparse_dev(self.handle, chk.handle, e_buf.ptr, batch, lde,
                                            s_buf.ptr if s_buf is not None else None, lds,
                                            hist_buf.ptr if hist_buf is not None else None, nbins))

    def histogram_dev(self, s_buf, batch, lds, r, mode, hist_buf, nbins, layout=LAYOUT_SAMPLE_MAJOR):
        check(lib().gf2_histogram_dev(self.handle, s_buf.ptr, batch, lds, layout, r, mode, hist_buf.ptr, nbins))

    def sample_errors_dev(self, n, seed, first, count, p_x, p_y, p_z, ex_buf, ez_buf, lde,
                          layout=LAYOUT_SAMPLE_MAJOR):
        check(lib().gf2_sample_errors_dev(self.handle, n, seed & 0xFFFFFFFFFFFFFFFF, first, count, p_x, p_y, p_z,
                                          ex_buf.ptr, ez_buf.ptr, lde, layout))

    def retile_dev(self, e_buf, batch, lde, n, tiled_buf):
        check(lib().gf2_retile_dev(self.handle, e_buf.ptr, batch, lde, n, tiled_buf.ptr))

    def mc_run(self, chk1, chk2, seed, first, count, p_x, p_y, p_z, mode):
        """Returns (hist_z, hist_x) as uint64 arrays."""
        if mode == HIST_FULL:
            nz, nx = 1 << chk1.r, 1 << chk2.r
        else:
            nz, nx = chk1.r + 1, chk2.r + 1
        hist_z = np.zeros(nz, dtype=np.uint64)
        hist_x = np.zeros(nx, dtype=np.uint64)
        check(lib().gf2_mc_run(self.handle, chk1.handle, chk2.handle, seed & 0xFFFFFFFFFFFFFFFF, first, count,
                               p_x, p_y, p_z, mode, _ptr(hist_z), nz, _ptr(hist_x), nx))
        return hist_z, hist_x


    def mc_decode(self, chk1, chk2, table_c1, table_c2, x_operator, z_operator, seed, first, count, p_x, p_y, p_z):
        """Returns the five counts of gf2_mc_decode as a uint64 array."""
        t1 = np.ascontiguousarray(table_c1, dtype=np.uint64)
        t2 = np.ascontiguousarray(table_c2, dtype=np.uint64)
        counts = np.zeros(5, dtype=np.uint64)
        check(lib().gf2_mc_decode(self.handle, chk1.handle, chk2.handle, _ptr(t1), _ptr(t2), int(x_operator),
                                  int(z_operator), seed & 0xFFFFFFFFFFFFFFFF, first, count, p_x, p_y, p_z, _ptr(counts)))
        return counts


class Comm(object):
    """An RCCL communicator for the histogram all-reduce (gf2_comm_*).

    Comm.unique_id()                     bytes to hand to every rank out of band (rank 0 makes them)
    Comm(ctx, id, nranks, rank)          one process per GPU
    Comm.all_local([ctx0, ctx1, ...])    one process, one context per device
    """

    def __init__(self, ctx, unique_id, nranks, rank):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("the communicator id has %d bytes" % COMM_ID_BYTES)
        out = ctypes.c_void_p()
        ident = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        check(lib().gf2_comm_create(ctx.handle, ident, int(nranks), int(rank), ctypes.byref(out)))
        self.handle, self.contexts, self.nranks = out.value, [ctx], int(nranks)

    @staticmethod
    def unique_id():
        ident = ctypes.create_string_buffer(COMM_ID_BYTES)
        check(lib().gf2_comm_unique_id(ident, COMM_ID_BYTES))
        return ident.raw

    @classmethod
    def all_local(cls, contexts):
        self = cls.__new__(cls)
        handles = (ctypes.c_void_p * len(contexts))(*[c.handle for c in contexts])
        out = ctypes.c_void_p()
        check(lib().gf2_comm_create_all(handles, len(contexts), ctypes.byref(out)))
        self.handle, self.contexts, self.nranks = out.value, list(contexts), len(contexts)
        return self

    def allreduce(self, bufs, nbins):
        """In-place sum of `nbins` uint64 bins over all ranks; bufs: one DeviceBuffer per context of this communicator."""
        bufs = list(bufs) if isinstance(bufs, (list, tuple)) else [bufs]
        if len(bufs) != len(self.contexts):
            raise ValueError("one buffer per context")
        if any(b.nbytes < 8 * nbins for b in bufs):
            raise ValueError("buffer smaller than nbins words")
        ptrs = (ctypes.c_void_p * len(bufs))(*[b.ptr for b in bufs])
        check(lib().gf2_hist_allreduce(self.handle, ptrs, int(nbins)))

    def allreduce_host(self, hists):
        """Sums a list of host uint64 arrays over the ranks (one all-reduce of the concatenation, through device memory of
        this process's context)."""
        if len(self.contexts) != 1:
            raise ValueError("allreduce_host is for one context per process")
        sizes = [int(np.asarray(h).size) for h in hists]
        flat = np.ascontiguousarray(np.concatenate([np.asarray(h, dtype=np.uint64).ravel() for h in hists]))
        buf = self.contexts[0].alloc(flat.nbytes).upload(flat)
        self.allreduce(buf, flat.size)
        total = buf.download((flat.size,), np.uint64)
        buf.free()
        out, pos = [], 0
        for size in sizes:
            out.append(total[pos:pos + size].copy())
            pos += size
        return out

    def close(self):
        if self.handle:
            handle, self.handle = self.handle, None
            check(lib().gf2_comm_destroy(handle))


def rccl_version():
    out = ctypes.c_int(0)
    check(lib().gf2_rccl_version(ctypes.byref(out)))
    return int(out.value)


_default = None
_default_lock = threading.Lock()


def default_context():
    """Process-wide context on GF2_DEVICE, else LOCAL_RANK, else device 0."""
    global _default
    if _default is None:
        with _default_lock:
            if _default is None:
                device = int(os.environ.get("GF2_DEVICE", os.environ.get("LOCAL_RANK", "0")))
                _default = Context(device)
    return _default
