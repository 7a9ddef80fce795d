"""
ORACLE -- test infrastructure only.  NOT part of the product.

CPU (NumPy) restatement of the GF(2) hot path of jimpo/quantum-css-codes.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (quantum_css_codes_amd) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function below that restates a reference function is checked in
tests/test_oracle_golden.py against (a) the reference's own known-answer tests
(test/test_bin_matrix.py:8-31, test/test_css_code.py:13-59,108-143) and (b) fixtures produced by
importing the reference itself in the build container (tests/golden/make_golden.py, outputs in
tests/golden/*.npz).

Each restated function cites the reference lines it follows (paths relative to the reference
repo).  The arithmetic mirrors the reference: dense integer NumPy arrays, additions that are
only reduced modulo 2 at the end, first-match pivot searches, add-instead-of-swap row moves.
That is deliberate -- this file is also the "reference's NumPy CPU path" that bench.py times on
the GPU box's host cores.

Functions marked [build-defined] have no counterpart in the reference; SURVEY.md section 8 (x1-x3)
anchors their definitions to reference lines and DESIGN.md states them in full.
"""

import itertools

import numpy as np


class InvalidCodeError(Exception):
    """errors.py:5-6"""


# --------------------------------------------------------------------------------------------
# bin_matrix.py
# --------------------------------------------------------------------------------------------

def reduced_row_echelon_form(mat):
    """bin_matrix.py:8-34.  Gauss-Jordan over GF(2) on a copy; a pivot row is *added* into
    position (never swapped), every other odd row in the pivot column is cleared, and the
    single reduction modulo 2 happens at the very end, so intermediate entries grow."""
    work = np.copy(mat)
    n_rows, n_cols = work.shape
    lead = 0
    for col in range(n_cols):
        found = None
        for cand in range(lead, n_rows):
            if work[cand, col] % 2 == 1:
                found = cand
                break
        if found is None:
            continue
        if work[lead, col] % 2 == 0:
            work[lead, :] += work[found, :]
        for other in range(n_rows):
            if other != lead and work[other, col] % 2 == 1:
                work[other, :] += work[lead, :]
        lead += 1
    return np.mod(work, 2)


def vec_to_int(vec):
    """bin_matrix.py:36-43.  Big-endian (vec[0] is the most significant bit).  Like the
    reference the accumulator takes NumPy's integer type after the first addition, so vectors of
    64 bits or more wrap -- that is reference behaviour and is reproduced here on purpose."""
    acc = 0
    for pos in range(vec.size):
        acc = (acc << 1) + vec[pos]
    return acc


def int_to_vec(int_repr, n):
    """bin_matrix.py:45-55."""
    out = np.zeros(n, dtype='int')
    for pos in range(n - 1, -1, -1):
        out[pos] = int_repr & 1
        int_repr = int_repr >> 1
    if int_repr != 0:
        raise ValueError("n is too small")
    return out


def weight_w_vectors(n, w):
    """bin_matrix.py:57-72.  All length-n vectors of Hamming weight w, supports in
    lexicographic order, a fresh array per item."""
    for support in itertools.combinations(range(n), w):
        item = np.zeros(n, dtype='int')
        item[list(support)] = 1
        yield item


# --------------------------------------------------------------------------------------------
# css_code.py -- free functions
# --------------------------------------------------------------------------------------------

def swap_columns(mat, indices):
    """css_code.py:783-785.  In place."""
    a, b = indices
    col_a = np.array(mat[:, a])
    mat[:, a] = mat[:, b]
    mat[:, b] = col_a


def normalize_parity_check(h, offset):
    """css_code.py:809-836.  Sequential elimination that leaves an identity block in columns
    offset..offset+r-1.  Mutates h, returns (h mod 2, list of (column, column) swaps).  The
    result depends on the order of operations, so the order here is the
    reference's: first odd row at or below the diagonal is added to the diagonal row; when
    there is none, the first odd column of the diagonal row is swapped in."""
    r, n = h.shape
    if n < offset + r:
        raise ValueError("not enough columns")
    swaps = []
    for diag in range(r):
        col = diag + offset
        donor = None
        for cand in range(diag, r):
            if h[cand, col] % 2 == 1:
                donor = cand
                break
        if donor is not None:
            if h[diag, col] % 2 == 0:
                h[diag, :] += h[donor, :]
        else:
            other_col = None
            for cand in range(col, n):
                if h[diag, cand] % 2 == 1:
                    other_col = cand
                    break
            if other_col is None:
                raise InvalidCodeError("rows are not independent")
            swaps.append((col, other_col))
            swap_columns(h, swaps[-1])
        for other in range(r):
            if other != diag and h[other, col] % 2 == 1:
                h[other, :] += h[diag, :]
    return np.mod(h, 2), swaps


def syndrome_product(parity_check, e):
    """css_code.py:728 (also :47 and test/test_css_code.py:116): np.mod(np.matmul(H, e), 2)."""
    return np.mod(np.matmul(parity_check, e), 2)


def syndrome_table(parity_check, max_weight=None):
    """css_code.py:715-735.  max_weight=None is the reference behaviour (runs until the first
    collision).  max_weight=w [build-defined cap] stops after weight class w when no collision
    has been seen by then and returns (w, table)."""
    _, n = parity_check.shape
    table = {}
    for w in range(n + 1):
        if max_weight is not None and w > max_weight:
            return max_weight, table
        layer = {}
        for e in weight_w_vectors(n, w):
            key = vec_to_int(syndrome_product(parity_check, e))
            if key in table or key in layer:
                return w - 1, table
            layer[key] = e
        table = {**table, **layer}
    return n, table


def codes_equal(parity_check_1, parity_check_2):
    """css_code.py:838-844."""
    if parity_check_1.shape != parity_check_2.shape:
        return False
    return np.array_equal(reduced_row_echelon_form(parity_check_1),
                          reduced_row_echelon_form(parity_check_2))


def is_doubly_even(mat):
    """css_code.py:846-850."""
    return not np.any(np.mod(np.sum(mat, axis=1), 4))


def pauli_label_for_row(x_check, z_check):
    """Text form of css_code.py:787-807 (pauli_term_for_row) without pyQuil: factors in qubit
    order, 'Y' where both checks are set, 'I' for the empty product.  test/test_css_code.py:32-59
    pins the Steane values (sX(0)*sX(3)*... == 'X0*X3*...')."""
    n = x_check.size
    if not x_check.shape == (n,):
        raise ValueError("x_check has the wrong dimensions")
    if not z_check.shape == (n,):
        raise ValueError("z_check has the wrong dimensions")
    factors = []
    for q in range(n):
        if x_check[q] and z_check[q]:
            factors.append("Y%d" % q)
        elif x_check[q]:
            factors.append("X%d" % q)
        elif z_check[q]:
            factors.append("Z%d" % q)
    return "*".join(factors) if factors else "I"


# --------------------------------------------------------------------------------------------
# css_code.py -- CSSCode numeric core
# --------------------------------------------------------------------------------------------

class CSSCode(object):
    """Numeric part of css_code.py:32-75, 124-136, 149-161, 174-201 (no Quil emission)."""

    def __init__(self, parity_check_c1, parity_check_c2, max_table_weight=None):
        r_1, n_1 = parity_check_c1.shape
        r_2, n_2 = parity_check_c2.shape
        if n_1 != n_2:
            raise ValueError("C_1 and C_2 must have the same code word length")

        h_1 = np.mod(np.array(parity_check_c1, dtype='int'), 2)
        h_2 = np.mod(np.array(parity_check_c2, dtype='int'), 2)
        if not np.array_equal(h_1, parity_check_c1):
            raise ValueError("C_1 parity check matrix must be binary")
        if not np.array_equal(h_2, parity_check_c2):
            raise ValueError("C_2 parity check matrix must be binary")

        if np.any(np.mod(np.matmul(h_1, np.transpose(h_2)), 2)):
            raise ValueError("C_2 dual code must be a subspace of C_1")

        h_1, swaps = normalize_parity_check(h_1, offset=0)
        for pair in swaps:
            swap_columns(h_2, pair)
        self.swaps_1 = list(swaps)
        h_2, swaps = normalize_parity_check(h_2, offset=r_1)
        for pair in swaps:
            swap_columns(h_1, pair)
        self.swaps_2 = list(swaps)

        self._n = n_1
        self._k = n_1 - r_1 - r_2
        self.r_1 = r_1
        self.r_2 = r_2
        self.parity_check_c1 = h_1
        self.parity_check_c2 = h_2
        t_1, self._c1_syndromes = syndrome_table(h_1, max_table_weight)
        t_2, self._c2_syndromes = syndrome_table(h_2, max_table_weight)
        self._t = min(t_1, t_2)
        self._transversal_gates = self._determine_transversal_gates(h_1, h_2)

        if self.k != 1:
            raise InvalidCodeError("currently only supports CSS codes for a single logical qubit")

    n = property(lambda self: self._n)
    k = property(lambda self: self._k)
    t = property(lambda self: self._t)

    def z_operator_matrix(self):
        """css_code.py:124-136: [A2^T 0 I]."""
        n, r_1, r_2, k = self.n, self.r_1, self.r_2, self.k
        out = np.zeros((k, n), dtype='int')
        out[:, 0:r_1] = np.transpose(self.parity_check_c1[:, (r_1 + r_2):n])
        out[:, (r_1 + r_2):n] = np.identity(k)
        return out

    def x_operator_matrix(self):
        """css_code.py:149-161: [0 E^T I]."""
        n, r_1, r_2, k = self.n, self.r_1, self.r_2, self.k
        out = np.zeros((k, n), dtype='int')
        out[:, r_1:(r_1 + r_2)] = np.transpose(self.parity_check_c2[:, (r_1 + r_2):n])
        out[:, (r_1 + r_2):n] = np.identity(k)
        return out

    def stabiliser_labels(self):
        """css_code.py:98-111 rendered as text: rows of H1 as X-type first, then rows of H2 as
        Z-type."""
        zeros = np.zeros(self.n, dtype='int')
        xs = [pauli_label_for_row(self.parity_check_c1[i, :], zeros) for i in range(self.r_1)]
        zs = [pauli_label_for_row(zeros, self.parity_check_c2[i, :]) for i in range(self.r_2)]
        return xs + zs

    def is_transversal(self, gate_name):
        """css_code.py:174-180."""
        return gate_name in self._transversal_gates

    @staticmethod
    def _determine_transversal_gates(parity_check_c1, parity_check_c2):
        """css_code.py:182-201."""
        names = ['I', 'CNOT']
        if codes_equal(parity_check_c1, parity_check_c2):
            names += ['H', 'CZ']
            if is_doubly_even(parity_check_c1):
                names.append('S')
        return frozenset(names)


# --------------------------------------------------------------------------------------------
# [build-defined] x1: nullspace, x2: batched syndromes, x3: Monte-Carlo sampler + histograms
# --------------------------------------------------------------------------------------------

def nullspace(mat):
    """[build-defined, SURVEY.md 8a x1]  Canonical basis of {v : mat.v = 0 mod 2} read off the
    RREF R of mat (bin_matrix.py:8-34) the way css_code.py:124-161 reads logical operators off a
    standard form: with pivot columns P (ascending, pivot i in row i) and free columns F
    (ascending), basis row t has a 1 at F[t] and R[i, F[t]] at P[i]."""
    red = np.mod(reduced_row_echelon_form(np.array(mat, dtype='int')), 2)
    m, n = red.shape
    pivots = []
    for i in range(m):
        nz = np.flatnonzero(red[i])
        if nz.size == 0:
            break
        pivots.append(int(nz[0]))
    free = [c for c in range(n) if c not in set(pivots)]
    basis = np.zeros((len(free), n), dtype='int')
    for t, fc in enumerate(free):
        basis[t, fc] = 1
        for i, pc in enumerate(pivots):
            basis[t, pc] = red[i, fc]
    return basis


def syndrome_batch(parity_check, errors):
    """[build-defined, x2]  css_code.py:728 applied to B error vectors at once.  errors is B x n
    (one error per row); returns B x r."""
    return np.mod(np.matmul(errors, np.transpose(parity_check)), 2)


_M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
STREAM_MULT = 0xD1B54A32D192ED03


def mix64(z):
    """splitmix64 output function (Steele, Lea, Flood 2014) on Python ints."""
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def quantise_probability(x):
    """Threshold T in [0, 2^32]: a 32-bit uniform u maps to 1 iff u < T."""
    t = int(np.floor(float(x) * 4294967296.0 + 0.5))
    return max(0, min(t, 1 << 32))


def pauli_thresholds(p_x, p_y, p_z):
    """(t_any, t_1, t_2): a qubit errs with probability t_any / 2^32; the kind uniform c of an erroneous qubit gives
    X for c < t_1, Y for t_1 <= c < t_2, Z otherwise."""
    p_t = p_x + p_y + p_z
    t_any = quantise_probability(p_t)
    t_1 = quantise_probability(p_x / p_t) if p_t > 0 else 0
    t_2 = quantise_probability((p_x + p_y) / p_t) if p_t > 0 else 0
    return t_any, t_1, t_2


SEGMENT = 512                # qubits per segment of the sampler


def _binomial_cdf_direct(t_any, nb):
    cdf = [1 << 32] * (max(nb, 64) + 1)
    if nb <= 0:
        return cdf
    if t_any >= (1 << 32):
        for k in range(nb):
            cdf[k] = 0
        return cdf
    q = float(t_any) / 4294967296.0
    om = 1.0 - q
    pmf = 1.0
    for _ in range(nb):
        pmf *= om
    cum = 0.0
    for k in range(nb):
        cum += pmf
        c = float(np.floor(cum * 4294967296.0 + 0.5))
        if c > 4294967296.0:
            c = 4294967296.0
        cdf[k] = int(c)
        pmf = pmf * float(nb - k) / float(k + 1) * q / om
    return cdf


_cdf_cache = {}


def binomial_cdf_table(t_any, nb):
    """cdf[k] = 2^32 * P(Bin(nb, q) <= k) rounded to nearest and clamped to 2^32, q = t_any / 2^32, in IEEE doubles with the
    operation order of DESIGN.md "Sampler".  The number of errors of a segment is K = #{k < nb : u >= cdf[k]}.  Segments of more
    than 64 qubits at q > 1/2 take the table from the complementary count nb - K ~ Bin(nb, 1 - q) ((1 - q)^nb underflows)."""
    key = (t_any, nb)
    if key not in _cdf_cache:
        if nb <= 64 or t_any <= (1 << 31) or t_any >= (1 << 32):
            cdf = _binomial_cdf_direct(t_any, nb)
        else:
            other = _binomial_cdf_direct((1 << 32) - t_any, nb)
            cdf = [1 << 32] * (nb + 1)
            for k in range(nb):
                cdf[k] = (1 << 32) - other[nb - k - 1]
        _cdf_cache[key] = cdf
    return _cdf_cache[key]


def sample_pauli_error(seed, sample, n, p_x, p_y, p_z):
    """[build-defined, x3]  Error of global sample index `sample`: a pure function of (seed, sample).  Returns
    (e_x, e_z) as length-n int arrays.  Per segment of 512 qubits one draw d: its high half gives the number of erroneous
    qubits by inverse binomial CDF; each erroneous qubit takes one further draw mix64(d + G (k + 1)), whose high half
    picks its position inside the segment (Floyd's algorithm) and whose low half its kind (DESIGN.md "Sampler")."""
    t_any, t_1, t_2 = pauli_thresholds(p_x, p_y, p_z)
    ks = mix64(seed + GOLDEN * (sample + 1))
    e_x = np.zeros(n, dtype='int')
    e_z = np.zeros(n, dtype='int')
    segments = (n + SEGMENT - 1) // SEGMENT
    for s in range(segments):
        nb = SEGMENT if s < segments - 1 else n - SEGMENT * (segments - 1)
        cdf = binomial_cdf_table(t_any, nb)
        d = mix64(ks + STREAM_MULT * (s + 1))
        u = d >> 32
        k_err = 0
        while k_err < nb and u >= cdf[k_err]:
            k_err += 1
        chosen = set()
        for k in range(k_err):                                 # Floyd: k_err distinct positions in range(nb)
            v = mix64(d + GOLDEN * (k + 1))
            j = nb - k_err + k
            t = ((v >> 32) * (j + 1)) >> 32
            pos = j if t in chosen else t
            chosen.add(pos)
            kind = v & 0xFFFFFFFF
            e_x[SEGMENT * s + pos] = 1 if kind < t_2 else 0
            e_z[SEGMENT * s + pos] = 1 if kind >= t_1 else 0
    return e_x, e_z


def monte_carlo_histograms(h_1, h_2, seed, first_sample, num_samples, p_x, p_y, p_z, mode):
    """[build-defined, x3]  X errors are caught by parity_check_c2, Z errors by parity_check_c1
    (css_code.py:457-470).  mode 'full': bins indexed by vec_to_int(syndrome) (css_code.py:729),
    2^r bins; mode 'weight': bins indexed by the syndrome's Hamming weight, r+1 bins.
    Returns (hist_z, hist_x) as uint64 arrays (hist_z from H1.e_z, hist_x from H2.e_x)."""
    r_1, n = h_1.shape
    r_2, _ = h_2.shape
    if mode == 'full':
        hist_z = np.zeros(1 << r_1, dtype=np.uint64)
        hist_x = np.zeros(1 << r_2, dtype=np.uint64)
    else:
        hist_z = np.zeros(r_1 + 1, dtype=np.uint64)
        hist_x = np.zeros(r_2 + 1, dtype=np.uint64)
    for i in range(first_sample, first_sample + num_samples):
        e_x, e_z = sample_pauli_error(seed, i, n, p_x, p_y, p_z)
        s_z = syndrome_product(h_1, e_z)
        s_x = syndrome_product(h_2, e_x)
        if mode == 'full':
            hist_z[int(vec_to_int(s_z))] += np.uint64(1)
            hist_x[int(vec_to_int(s_x))] += np.uint64(1)
        else:
            hist_z[int(s_z.sum())] += np.uint64(1)
            hist_x[int(s_x.sum())] += np.uint64(1)
    return hist_z, hist_x


def decode_and_tally(code, seed, first_sample, num_samples, p_x, p_y, p_z):
    """[build-defined, SURVEY.md 8f item 1]  Classical content of quil_classical_correct (css_code.py:649-685) and
    noisy_measure (css_code.py:640-646) on sampled errors, one sample at a time with the reference's own data
    structures: the syndrome tables are the dicts built by syndrome_table, keys by vec_to_int.  `code` is a CSSCode
    of this module.  Returns [logical X flips, logical Z flips, either, X syndrome not in table, Z syndrome not in
    table]."""
    z_op = code.z_operator_matrix()[0]
    x_op = code.x_operator_matrix()[0]
    counts = [0, 0, 0, 0, 0]
    for i in range(first_sample, first_sample + num_samples):
        e_x, e_z = sample_pauli_error(seed, i, code.n, p_x, p_y, p_z)
        flips = []
        for err, check, table, op, miss_slot in ((e_x, code.parity_check_c2, code._c2_syndromes, z_op, 3),
                                                 (e_z, code.parity_check_c1, code._c1_syndromes, x_op, 4)):
            key = int(vec_to_int(syndrome_product(check, err)))
            errors = np.zeros(code.n, dtype='int')             # the block's known-error register starts at zero
            if key in table:
                errors = np.mod(errors + table[key], 2)         # conditional_xor of the matching correction
            else:
                counts[miss_slot] += 1                          # no match: errors left unchanged
            residual = np.mod(err + errors, 2)                  # codeword XOR errors, codeword = true codeword + err
            flips.append(int(np.mod(np.dot(op, residual), 2)))
        counts[0] += flips[0]
        counts[1] += flips[1]
        counts[2] += 1 if (flips[0] or flips[1]) else 0
    return counts


# --------------------------------------------------------------------------------------------
# css_code.py -- encoder gate lists and stabiliser conjugation (pyquil-free forms)
# --------------------------------------------------------------------------------------------
# Gates are rows (kind, a, b) of an int array: kind 0 = H on qubit a, kind 1 = CNOT control a target b.
# Pinned by tests/golden/conjugation_golden.npz (the reference's own conjugate_* functions run on seeded inputs) and
# by the reference's known answers for the Steane encoders (test/test_css_code.py:61-106).

GATE_H, GATE_CNOT = 0, 1


def conjugate_h_with_check_mat(mat, qubit):
    """css_code.py:757-767: H swaps the X and Z entry of the qubit in every row; a row with both set is refused."""
    k, cols = mat.shape
    n = cols // 2
    for i in range(k):
        if mat[i, qubit] == 1 and mat[i, n + qubit] == 1:
            raise NotImplementedError("only handles CSS codes")
        mat[i, qubit], mat[i, n + qubit] = mat[i, n + qubit], mat[i, qubit]


def conjugate_cnot_with_check_mat(mat, control, target):
    """css_code.py:769-781: X spreads from control to target, Z from target to control."""
    k, cols = mat.shape
    n = cols // 2
    for i in range(k):
        if mat[i, control] == 1:
            mat[i, target] = (mat[i, target] + 1) % 2
        if mat[i, n + target] == 1:
            mat[i, n + control] = (mat[i, n + control] + 1) % 2


def transform_stabilisers(mat, gates):
    """css_code.py:737-755 on a gate array: in place, in order; ValueError for a qubit outside [0, n) or an unknown gate."""
    _, cols = mat.shape
    n = cols // 2
    for kind, a, b in np.asarray(gates).reshape(-1, 3):
        qubits = (a,) if kind == GATE_H else (a, b)
        if any(q < 0 or q >= n for q in qubits):
            raise ValueError("qubit index must be within [0, n)")
        if kind == GATE_H:
            conjugate_h_with_check_mat(mat, int(a))
        elif kind == GATE_CNOT:
            conjugate_cnot_with_check_mat(mat, int(a), int(b))
        else:
            raise ValueError("cannot conjugate gate {}".format(kind))


def encode_zero_gates(code, qubits=None):
    """Gate sequence of CSSCode.noisy_encode_zero (css_code.py:203-259): H on the first r_1 qubits, then
    CNOT(i, j) for every 1 of parity_check_c1[i, j], j >= r_1, rows in order, columns in order."""
    n, r_1 = code.n, code.r_1
    qubits = list(range(n)) if qubits is None else list(qubits)
    gates = [(GATE_H, qubits[i], 0) for i in range(r_1)]
    for i in range(r_1):
        for j in range(r_1, n):
            if code.parity_check_c1[i, j] == 1:
                gates.append((GATE_CNOT, qubits[i], qubits[j]))
    return np.array(gates, dtype=np.int32).reshape(-1, 3)


def encode_plus_gates(code, qubits=None):
    """Gate sequence of CSSCode.noisy_encode_plus (css_code.py:261-312): H on the first r_1 and the last n - r_1 - r_2
    qubits, CNOT(j, i) for every 1 of parity_check_c2[i - r_1, j] with j >= r_1 + r_2, then the CNOTs of encode_zero."""
    n, r_1, r_2 = code.n, code.r_1, code.r_2
    qubits = list(range(n)) if qubits is None else list(qubits)
    gates = [(GATE_H, qubits[i], 0) for i in range(r_1)]
    gates += [(GATE_H, qubits[i], 0) for i in range(r_1 + r_2, n)]
    for i in range(r_1, r_1 + r_2):
        for j in range(r_1 + r_2, n):
            if code.parity_check_c2[i - r_1, j] == 1:
                gates.append((GATE_CNOT, qubits[j], qubits[i]))
    for i in range(r_1):
        for j in range(r_1, n):
            if code.parity_check_c1[i, j] == 1:
                gates.append((GATE_CNOT, qubits[i], qubits[j]))
    return np.array(gates, dtype=np.int32).reshape(-1, 3)
