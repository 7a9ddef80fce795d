"""
Utilities for binary NumPy matrices -- drop-in for the reference's bin_matrix.py, with the GF(2)
elimination running on the MI355X through libgf2hip.so.

Same names, argument meaning, return types and exceptions as bin_matrix.py:8-72.  Two documented
differences: vec_to_int returns an exact Python int (the reference
silently wraps at 64 bits), and bool input is treated as bits (the reference's `+=` on bool arrays is
a logical OR and gives wrong answers).
"""
import numpy as np

from . import _native


def reduced_row_echelon_form(mat):
    """
    Returns a new copy of a binary matrix in reduced row echelon form (bin_matrix.py:8-34): same
    shape and dtype, input untouched.  Entries are taken modulo 2.  The RREF is unique, so the packed
    row-swapping elimination on the GPU returns exactly what the reference's add-not-swap loop does.
    """
    mat = np.asarray(mat)
    m, n = mat.shape
    if m == 0 or n == 0:
        return np.mod(np.copy(mat), 2)
    packed = _native.pack_rows(mat)
    _native.default_context().rref(packed, m, n)
    return _native.unpack_rows(packed, n, dtype=mat.dtype)


def rank(mat):
    """Rank over GF(2) [build-defined convenience; by-product of the RREF kernel]."""
    mat = np.asarray(mat)
    m, n = mat.shape
    if m == 0 or n == 0:
        return 0
    packed = _native.pack_rows(mat)
    return _native.default_context().rref(packed, m, n)[1]


def nullspace(mat):
    """
    [build-defined, SURVEY.md 8a x1]  Canonical basis of the GF(2) nullspace read off the RREF R of
    `mat`: with pivot columns P and free columns F (both ascending), row t has a 1 at F[t] and
    R[i, F[t]] at P[i].  For H = [I A] this is [A^T I], the construction css_code.py:124-161 uses for
    the logical operators.  Returns an (n - rank) x n array of dtype 'int'.
    """
    mat = np.asarray(mat)
    m, n = mat.shape
    if n == 0:
        return np.zeros((0, 0), dtype='int')
    packed = _native.pack_rows(mat)
    basis = _native.default_context().nullspace(packed, m, n)
    return _native.unpack_rows(basis, n, dtype='int')


def vec_to_int(vec):
    """
    Convert a big-endian bit vector to an integer (bin_matrix.py:36-43); vec[0] is the most
    significant bit.  Exact for any length.
    """
    vec = np.asarray(vec)
    flat = vec.reshape(-1)
    if flat.size and flat.dtype.kind in 'biu' and bool(np.all((flat == 0) | (flat == 1))):
        # bits: the same sum through one big-endian byte string (the Python loop costs a millisecond per 2048 bits)
        packed = np.packbits(flat.astype(np.uint8), bitorder='big')
        return int.from_bytes(packed.tobytes(), 'big') >> ((-flat.size) % 8)
    result = 0
    for i in range(vec.size):
        result = (result << 1) + int(vec[i])
    return result


def int_to_vec(int_repr, n):
    """
    Convert an int to its big-endian bit vector representation (bin_matrix.py:45-55).
    """
    int_repr = int(int_repr)
    vec = np.zeros(n, dtype='int')
    for i in reversed(range(n)):
        vec[i] = int_repr & 1
        int_repr >>= 1
    if int_repr != 0:
        raise ValueError("n is too small")
    return vec


def weight_w_vectors(n, w):
    """
    Generate all length n binary vectors with Hamming weight w (bin_matrix.py:57-72): supports in
    lexicographic order, a fresh array per item.
    """
    def extend(vec, remaining, start):
        if remaining == 0:
            yield np.copy(vec)
            return
        for i in range(start, vec.size):
            vec[i] = 1
            yield from extend(vec, remaining - 1, i + 1)
            vec[i] = 0

    yield from extend(np.zeros(n, dtype='int'), w, 0)


def weight_w_supports(n, w):
    """All weight-w supports as a C(n,w) x w index array, in the order weight_w_vectors yields them
    [build-defined helper for the batched syndrome table]."""
    import itertools
    combos = list(itertools.combinations(range(n), w))
    return np.array(combos, dtype=np.int64).reshape(len(combos), w)
