"""
Calderbank-Shor-Steane codes: the numeric core of the reference's css_code.py with every GF(2)
product and elimination running on the MI355X (libgf2hip.so).

Drop-in surface (same names, arguments, mutation and exception behaviour):
    CSSCode(parity_check_c1, parity_check_c2)         css_code.py:32-75
      .n .k .t .r_1 .r_2 .parity_check_c1 .parity_check_c2 ._c1_syndromes ._c2_syndromes
      .z_operator_matrix() .x_operator_matrix()       css_code.py:124-136, 149-161
      .stabilisers() .x_operators() .z_operators() .y_operators()   css_code.py:98-172
      .is_transversal(gate_name)                      css_code.py:174-201
    syndrome_table, normalize_parity_check, swap_columns, codes_equal, is_doubly_even,
    pauli_term_for_row                                css_code.py:715-735, 783-850

    CSSCode.noisy_encode_zero / noisy_encode_plus    css_code.py:203-312   (gate arrays, see below)
    transform_stabilisers, conjugate_h_with_check_mat, conjugate_cnot_with_check_mat   css_code.py:737-781

The two encoders are returned as gate arrays -- rows (kind, a, b), GATE_H on qubit a or GATE_CNOT control a target b, in
the reference's instruction order -- which is what transform_stabilisers takes here [SURVEY.md 8f item 3].  The other
emitters (encode_zero / encode_plus, apply_gate, error_correct, measure, noisy_measure, quil_classical_correct /
quil_classical_detect; css_code.py:314-713) build pyQuil-free instruction tuples (quil.py, css_emit.py) [8f item 4].
pyQuil is not required: Pauli operators are returned as text labels ("X0*X3*X4*X5") unless pyQuil is
importable, in which case PauliTerm objects are returned as in the reference.

Build-defined additions (8f item 1): CSSCode(..., max_table_weight=), CSSCode.syndromes,
CSSCode.monte_carlo, CSSCode.logical_error_rates, syndrome_batch.
"""
import itertools

import numpy as np

from . import _native
from . import bin_matrix
from . import css_emit
from .css_emit import apply_transversally, quil_classical_correct, quil_classical_detect  # noqa: F401  (module-level in the reference)
from .errors import InvalidCodeError
from .qecc import QECC

try:                                                # optional, as in SURVEY.md 7.3 item 7
    from pyquil.paulis import ID as _ID, sX as _sX, sY as _sY, sZ as _sZ
    _HAVE_PYQUIL = True
except Exception:                                   # pragma: no cover - pyquil is absent in this image
    _HAVE_PYQUIL = False


GATE_H, GATE_CNOT = 0, 1                            # kinds of a gate-array row (kind, a, b)


class CSSCode(QECC):
    """
    A CSS code defined by two binary linear codes C_1, C_2 with the dual of C_2 inside C_1, given by
    their parity check matrices.  Measured in the X basis a codeword is a codeword of C_1, in the Z
    basis a codeword of C_2 (the reference's convention, css_code.py:22-31).
    """

    def __init__(self, parity_check_c1, parity_check_c2, max_table_weight=None):
        parity_check_c1 = np.asarray(parity_check_c1)
        parity_check_c2 = np.asarray(parity_check_c2)
        r_1, n_1 = parity_check_c1.shape
        r_2, n_2 = parity_check_c2.shape
        if n_1 != n_2:
            raise ValueError("C_1 and C_2 must have the same code word length")

        # css_code.py:39-44: entries must be 0 or 1 (tested while the rows are packed: one pass over each 64 MiB array at n = 4096
        # instead of copy + np.mod + np.array_equal); from here to the end of the normalisations the matrices stay packed
        p_1, binary = _native.pack_rows_binary(parity_check_c1)
        if not binary:
            raise ValueError("C_1 parity check matrix must be binary")
        p_2, binary = _native.pack_rows_binary(parity_check_c2)
        if not binary:
            raise ValueError("C_2 parity check matrix must be binary")

        ctx = _native.default_context()
        # css_code.py:47-49 -- H1 . H2^T must vanish over GF(2)
        product = ctx.matmul_abt(p_1, r_1, p_2, r_2, n_1)
        if np.any(product):
            raise ValueError("C_2 dual code must be a subspace of C_1")

        # css_code.py:51-61 -- standard forms H1 = [I A1 A2], H2 = [D I E]; each normalisation's qubit
        # swaps are applied to the other matrix as well.
        for i, j in _normalize_packed(ctx, p_1, r_1, n_1, 0):
            if r_2 and i != j:
                ctx.swap_columns(p_2, r_2, n_1, i, j)
        for i, j in _normalize_packed(ctx, p_2, r_2, n_1, r_1):
            if r_1 and i != j:
                ctx.swap_columns(p_1, r_1, n_1, i, j)
        h_1 = _native.unpack_rows(p_1[:r_1], n_1, dtype='int')
        h_2 = _native.unpack_rows(p_2[:r_2], n_1, dtype='int')

        self._n = n_1
        self._k = n_1 - r_1 - r_2
        self.r_1 = r_1
        self.r_2 = r_2
        self.parity_check_c1 = h_1
        self.parity_check_c2 = h_2
        t_1, self._c1_syndromes = syndrome_table(h_1, max_weight=max_table_weight, _packed=p_1[:r_1])
        t_2, self._c2_syndromes = syndrome_table(h_2, max_weight=max_table_weight, _packed=p_2[:r_2])
        self._t = min(t_1, t_2)
        self._transversal_gates = self._determine_transversal_gates(h_1, h_2)
        self._checks = None

        if self.k != 1:
            raise InvalidCodeError("currently only supports CSS codes for a single logical qubit")

    @property
    def n(self):
        """Number of physical qubits per code block."""
        return self._n

    @property
    def k(self):
        """Number of logical qubits per code block."""
        return self._k

    @property
    def t(self):
        """Maximum number of errors per code block that can be corrected."""
        return self._t

    # -- operators ----------------------------------------------------------------------------------------
    def stabilisers(self):
        """Generators of the stabiliser group: the rows of H1 as X-type operators, then the rows of H2
        as Z-type operators (css_code.py:98-111)."""
        zeros = np.zeros(self.n, dtype='int')
        x_type = (pauli_term_for_row(self.parity_check_c1[i, :], zeros) for i in range(self.r_1))
        z_type = (pauli_term_for_row(zeros, self.parity_check_c2[i, :]) for i in range(self.r_2))
        return list(itertools.chain(x_type, z_type))

    def stabiliser_matrix(self):
        """[build-defined]  The same generators as a (r_1 + r_2) x 2n check matrix [X part | Z part]."""
        out = np.zeros((self.r_1 + self.r_2, 2 * self.n), dtype='int')
        out[:self.r_1, :self.n] = self.parity_check_c1
        out[self.r_1:, self.n:] = self.parity_check_c2
        return out

    def z_operator_matrix(self):
        """Check matrix of the logical Z operators, the row [A2^T 0 I] (css_code.py:124-136)."""
        n, r_1, r_2, k = self.n, self.r_1, self.r_2, self.k
        check_mat = np.zeros((k, n), dtype='int')
        check_mat[:, 0:r_1] = np.transpose(self.parity_check_c1[:, (r_1 + r_2):n])
        check_mat[:, (r_1 + r_2):n] = np.identity(k)
        return check_mat

    def x_operator_matrix(self):
        """Check matrix of the logical X operators, the row [0 E^T I] (css_code.py:149-161)."""
        n, r_1, r_2, k = self.n, self.r_1, self.r_2, self.k
        check_mat = np.zeros((k, n), dtype='int')
        check_mat[:, r_1:(r_1 + r_2)] = np.transpose(self.parity_check_c2[:, (r_1 + r_2):n])
        check_mat[:, (r_1 + r_2):n] = np.identity(k)
        return check_mat

    def z_operators(self):
        """css_code.py:113-122."""
        z_matrix = self.z_operator_matrix()
        zeros = np.zeros_like(z_matrix, dtype='int')
        return [pauli_term_for_row(zeros[i, :], z_matrix[i, :]) for i in range(self.k)]

    def x_operators(self):
        """css_code.py:138-147."""
        x_matrix = self.x_operator_matrix()
        zeros = np.zeros_like(x_matrix, dtype='int')
        return [pauli_term_for_row(x_matrix[i, :], zeros[i, :]) for i in range(self.k)]

    def y_operators(self):
        """Y = iXZ (css_code.py:163-172): qubits carrying both X and Z carry Y."""
        x_matrix, z_matrix = self.x_operator_matrix(), self.z_operator_matrix()
        return [pauli_term_for_row(x_matrix[i, :], z_matrix[i, :]) for i in range(self.k)]

    def is_transversal(self, gate_name):
        """Whether the gate is known to be fault tolerant when applied transversally (css_code.py:174-180)."""
        return gate_name in self._transversal_gates

    def _determine_transversal_gates(self, parity_check_c1, parity_check_c2):
        # css_code.py:182-201 (Steane 1998, Lemmas 2 and 3)
        names = ['I', 'CNOT']
        if codes_equal(parity_check_c1, parity_check_c2):
            names += ['H', 'CZ']
            if is_doubly_even(parity_check_c1):
                names.append('S')
        return frozenset(names)

    # -- build-defined: batched syndromes and Monte-Carlo ---------------------------------------------------
    def encode_zero_gates(self, qubits=None):
        """
        The instruction sequence of noisy_encode_zero (css_code.py:203-259) as an (g, 3) int32 array: H on the first
        r_1 qubits, then CNOT(i, j) for every 1 of parity_check_c1[i, j] with j >= r_1, rows in order, columns in order.
        """
        n, r_1 = self.n, self.r_1
        qubits = np.arange(n, dtype=np.int32) if qubits is None else np.asarray(list(qubits), dtype=np.int32)
        ctrl, targ = np.nonzero(self.parity_check_c1[:, r_1:] == 1)          # row-major: i ascending, then j
        gates = np.zeros((r_1 + len(ctrl), 3), dtype=np.int32)
        gates[:r_1, 0], gates[:r_1, 1] = GATE_H, qubits[:r_1]
        gates[r_1:, 0], gates[r_1:, 1], gates[r_1:, 2] = GATE_CNOT, qubits[ctrl], qubits[targ + r_1]
        return gates

    def encode_plus_gates(self, qubits=None):
        """
        The instruction sequence of noisy_encode_plus (css_code.py:261-312): H on the first r_1 and on the last
        n - r_1 - r_2 qubits, CNOT(j, i) for every 1 of parity_check_c2[i - r_1, j] with j >= r_1 + r_2, then the CNOTs of
        noisy_encode_zero.
        """
        n, r_1, r_2 = self.n, self.r_1, self.r_2
        qubits = np.arange(n, dtype=np.int32) if qubits is None else np.asarray(list(qubits), dtype=np.int32)
        had = np.concatenate((qubits[:r_1], qubits[r_1 + r_2:]))
        rows, cols = np.nonzero(self.parity_check_c2[:, r_1 + r_2:] == 1)
        ctrl, targ = np.nonzero(self.parity_check_c1[:, r_1:] == 1)
        gates = np.zeros((len(had) + len(rows) + len(ctrl), 3), dtype=np.int32)
        a, b = len(had), len(had) + len(rows)
        gates[:a, 0], gates[:a, 1] = GATE_H, had
        gates[a:b, 0], gates[a:b, 1], gates[a:b, 2] = GATE_CNOT, qubits[cols + r_1 + r_2], qubits[rows + r_1]
        gates[b:, 0], gates[b:, 1], gates[b:, 2] = GATE_CNOT, qubits[ctrl], qubits[targ + r_1]
        return gates

    def noisy_encode_zero(self, qubits):
        """css_code.py:203-259 as a gate array (see encode_zero_gates); qubits must be integers."""
        return self.encode_zero_gates(qubits)

    def noisy_encode_plus(self, qubits):
        """css_code.py:261-312 as a gate array (see encode_plus_gates); qubits must be integers."""
        return self.encode_plus_gates(qubits)

    # -- program emission (css_code.py:314-646), pyQuil-free: quantum_css_codes_amd/css_emit.py ------------------------------
    def noisy_encode_zero_program(self, qubits):
        """noisy_encode_zero (css_code.py:203-259) as instruction tuples on `qubits` (integers or placeholders)."""
        return css_emit.noisy_encode_zero_program(self, qubits)

    def noisy_encode_plus_program(self, qubits):
        """noisy_encode_plus (css_code.py:261-312) as instruction tuples."""
        return css_emit.noisy_encode_plus_program(self, qubits)

    def encode_zero(self, prog, block, ancilla, scratch):
        css_emit.encode_zero(self, prog, block, ancilla, scratch)

    def encode_plus(self, prog, block, ancilla, scratch):
        css_emit.encode_plus(self, prog, block, ancilla, scratch)

    def apply_gate(self, prog, gate_name, *blocks):
        css_emit.apply_gate(self, prog, gate_name, *blocks)

    def error_correct(self, prog, data, ancilla_1, ancilla_2, scratch):
        css_emit.error_correct(self, prog, data, ancilla_1, ancilla_2, scratch)

    def _error_detect_x(self, prog, data, ancilla, outcome, scratch, include_operators):
        css_emit.error_detect_x(self, prog, data, ancilla, outcome, scratch, include_operators)

    def _error_detect_z(self, prog, data, ancilla, outcome, scratch, include_operators):
        css_emit.error_detect_z(self, prog, data, ancilla, outcome, scratch, include_operators)

    def measure(self, prog, data, index, outcome, ancilla_1, ancilla_2, scratch, scratch_int):
        return css_emit.measure(self, prog, data, index, outcome, ancilla_1, ancilla_2, scratch, scratch_int)

    def noisy_measure(self, prog, data, index, outcome, ancilla_1, ancilla_2, scratch):
        css_emit.noisy_measure(self, prog, data, index, outcome, ancilla_1, ancilla_2, scratch)

    @property
    def encode_scratch_size(self):
        return css_emit.encode_scratch_size(self)

    @property
    def error_correct_scratch_size(self):
        return css_emit.error_correct_scratch_size(self)

    @property
    def measure_scratch_size(self):
        return css_emit.measure_scratch_size(self)

    def _device_checks(self):
        if self._checks is None:
            ctx = _native.default_context()
            self._checks = (ctx.check_create(_native.pack_rows(self.parity_check_c1), self.r_1, self.n),
                            ctx.check_create(_native.pack_rows(self.parity_check_c2), self.r_2, self.n))
        return self._checks

    def syndromes(self, x_errors, z_errors):
        """[build-defined, x2]  Syndromes of B errors at once.  x_errors and z_errors are B x n 0/1
        arrays.  X errors are detected by parity_check_c2 and Z errors by parity_check_c1
        (css_code.py:457-470).  Returns (s_x: B x r_2, s_z: B x r_1)."""
        return (syndrome_batch(self.parity_check_c2, x_errors), syndrome_batch(self.parity_check_c1, z_errors))

    def monte_carlo(self, num_samples, p_x, p_y, p_z, seed=0, first_sample=0, mode=None):
        """[build-defined, x3]  Histograms of the syndromes of `num_samples` independent Pauli errors
        (X, Y, Z with probability p_x, p_y, p_z per qubit); sample i of the global stream is a pure
        function of (seed, i).  mode 'full': bins indexed by bin_matrix.vec_to_int(syndrome), needs
        r_1, r_2 <= 24; mode 'weight': bins indexed by syndrome weight.  Default: 'full' when possible.
        Returns {'hist_z': from parity_check_c1 . e_z, 'hist_x': from parity_check_c2 . e_x, 'mode'}."""
        from . import montecarlo
        return montecarlo.run_local(self, num_samples, p_x, p_y, p_z, seed=seed, first_sample=first_sample,
                                    mode=mode)


    def logical_error_rates(self, num_samples, p_x, p_y, p_z, seed=0, first_sample=0):
        """[build-defined, SURVEY.md 8f item 1]  Monte-Carlo of the classical side of error correction followed by
        a logical measurement: for every sampled error the syndrome is looked up in this code's tables
        (quil_classical_correct, css_code.py:649-685: the correction is applied when the syndrome is in the table,
        otherwise the error stays), and the logical Z (X) measurement flips iff z_operator . residual_x
        (x_operator . residual_z) is odd (noisy_measure, css_code.py:640-646).  Codes of at most 128 qubits (dense tables of
        2^r words for n <= 63 and r_1, r_2 <= 20, hash tables on the device beyond).  Returns counts: logical_x, logical_z, logical_any, uncorrectable_x, uncorrectable_z, samples."""
        from . import montecarlo
        return montecarlo.decode_local(self, num_samples, p_x, p_y, p_z, seed=seed, first_sample=first_sample)


# -- free functions -----------------------------------------------------------------------------------------

def syndrome_batch(parity_check, errors):
    """[build-defined, x2]  np.mod(np.matmul(parity_check, e), 2) (css_code.py:728) for every row e of
    `errors` (B x n).  Returns a B x r array of dtype 'int'."""
    parity_check = np.asarray(parity_check)
    errors = np.asarray(errors)
    r, n = parity_check.shape
    if errors.ndim != 2 or errors.shape[1] != n:
        raise ValueError("errors must be B x n")
    batch = errors.shape[0]
    if batch == 0 or r == 0:
        return np.zeros((batch, r), dtype='int')
    ctx = _native.default_context()
    out = ctx.syndrome_batch(_native.pack_rows(parity_check), r, n, _native.pack_rows(errors), batch)
    return _native.unpack_rows(out, r, dtype='int')


def syndrome_table(parity_check, max_weight=None, _packed=None):
    """
    Given a parity check matrix of a binary linear code, determine the unique decoding threshold t and
    return it along with a lookup table from syndromes (as bin_matrix.vec_to_int keys) to error vectors
    of weight at most t (css_code.py:715-735).

    Codes of at most 8192 bits are searched entirely on the device (gf2_table.hip): up to 24 checks against a dense table of
    2^r slots, up to 127 checks against a hash table sized from the weight classes it holds.  Otherwise (r > 127) each
    weight class is enumerated in bin_matrix.weight_w_vectors order, in chunks, and its syndromes are computed on the GPU; a class containing a syndrome already seen (in an earlier class or earlier in the
    same class) ends the search and is discarded as a whole, exactly as the reference's loop does.  Keys are
    formed and compared as machine words when r <= 63 (the reference's own keys are only meaningful there,
    SURVEY.md 7.3 item 2) and as Python ints beyond.  max_weight [build-defined] caps the search for large
    codes, where the reference cannot finish.
    """
    parity_check = np.asarray(parity_check)
    r, n = parity_check.shape
    ctx = _native.default_context()
    packed_h = _native.pack_rows(parity_check) if _packed is None else np.ascontiguousarray(_packed)     # (the constructor has them)
    if 0 < n <= ctx.TABLE_MAX_N and r <= ctx.TABLE_MAX_R:
        # whole search on the device (gf2_syndrome_table): one kernel per weight class, first collision ends it
        t, dense = ctx.syndrome_table(packed_h, r, n, max_weight)
        keys = np.nonzero(dense != ctx.TABLE_EMPTY)[0]
        words = dense[keys]
        # the reference's insertion order: by weight, then bin_matrix.weight_w_vectors order (supports ascending
        # lexicographically = the word read with qubit 0 as its most significant bit, descending)
        rev = np.unpackbits(words.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little")
        rev = np.packbits(rev, axis=1, bitorder="big").view(">u8").reshape(-1).astype(np.uint64)
        order = np.lexsort((~rev, np.bitwise_count(words)))
        errs = _native.unpack_rows(words[order].reshape(-1, 1), n, dtype=np.uint8).astype('int')
        return t, dict(zip(keys[order].tolist(), errs))
    if ctx.TABLE_MAX_N < n <= ctx.TABLE_COLS_MAX_N and r <= ctx.TABLE_MAX_R:
        # two-word errors (n up to 127) or, beyond 128 bits, errors as position lists: the device table
        # holds (weight, rank in the class)
        if n <= ctx.TABLE_WIDE_MAX_N:
            t, dense = ctx.syndrome_table_wide(packed_h, r, n, max_weight)
        else:
            t, dense = ctx.syndrome_table_cols(packed_h, r, n, max_weight)
        keys = np.nonzero(dense != ctx.TABLE_EMPTY)[0]
        weight = (dense[keys] >> np.uint64(32)).astype(np.int64)
        rank = dense[keys] & np.uint64(0xFFFFFFFF)
        table = {}
        for w in range(int(weight.max()) + 1 if keys.size else 0):
            sel = np.flatnonzero(weight == w)
            supports = _native.unrank_supports(rank[sel], n, w)
            # the reference's insertion order inside a class: supports ascending lexicographically (bin_matrix.py:57-72)
            order = np.lexsort(tuple(supports[:, k] for k in range(w - 1, -1, -1))) if w else np.arange(sel.size)
            errs = np.zeros((sel.size, n), dtype='int')
            if w:
                errs[np.arange(sel.size)[:, None], supports] = 1
            table.update(zip(keys[sel][order].tolist(), errs[order]))
        return t, table
    if ctx.TABLE_MAX_R < r <= ctx.TABLE_HASH_MAX_R and 0 < n <= ctx.TABLE_HASH_MAX_N:
        # more than 24 checks (round 4; every k = 1 CSS code from n = 51 on has such a check): an open-addressing hash table on
        # the device, sized from the weight classes it holds (gf2_syndrome_table_hashed); exact keys of up to 127 bits
        try:
            t, keys, weight, rank = ctx.syndrome_table_hashed(packed_h, r, n, max_weight)
        except _native.GF2Error as err:
            if err.code != _native.GF2_E_NOMEM:
                raise
            # 2^28 errors enumerated and no two with one syndrome yet: where the reference's loop (css_code.py:722-733) would go on
            # for days (the first collision of a random code with r checks is expected after some 2^((r+1)/2) errors: from about
            # r = 56 on without a bound), this says so and names the way out
            raise ValueError("syndrome_table: more than 2^28 errors without two of one syndrome (r = %d checks, n = %d); pass "
                             "max_weight (CSSCode: max_table_weight) to bound the table" % (r, n)) from err
        keys = np.asarray(keys, dtype=object if r > 63 else np.uint64)
        table = {}
        for w in range(int(weight.max()) + 1 if weight.size else 0):
            sel = np.flatnonzero(weight == w)
            supports = _native.unrank_supports(rank[sel], n, w)
            order = np.lexsort(tuple(supports[:, k] for k in range(w - 1, -1, -1))) if w else np.arange(sel.size)
            errs = np.zeros((sel.size, n), dtype='int')
            if w:
                errs[np.arange(sel.size)[:, None], supports] = 1
            table.update(zip(keys[sel][order].tolist(), errs[order]))
        return t, table
    chk = ctx.check_create(packed_h, r, n) if r else None
    table = {}
    seen = np.zeros(0, dtype=np.uint64)                       # keys of `table`, sorted (r <= 63)
    chunk_size = 1 << 18
    weights = (np.uint64(1) << np.arange(r - 1, -1, -1, dtype=np.uint64)) if 0 < r <= 63 else None
    for w in range(n + 1):
        if max_weight is not None and w > max_weight:
            return max_weight, table
        layer_keys, layer_errs = [], []
        layer_seen = np.zeros(0, dtype=np.uint64)             # keys of this weight class so far (r <= 63)
        layer_seen_big = set()                                # the same as Python ints (r > 63)
        combos = itertools.combinations(range(n), w)
        while True:
            batch = list(itertools.islice(combos, chunk_size))
            count = len(batch)
            if count == 0:
                break
            supports = np.array(batch, dtype=np.int64).reshape(count, w)
            # the error vectors as the table hands them out (dtype 'int'; np.zeros maps untouched pages lazily) and, packed
            # straight from the supports, as the syndrome kernel reads them
            errors = np.zeros((count, n), dtype='int')
            packed_e = np.zeros((count, max(1, _native.words_for(n))), dtype=np.uint64)
            if w:
                rows = np.arange(count)[:, None]
                errors[rows, supports] = 1
                np.bitwise_or.at(packed_e, (np.broadcast_to(rows, supports.shape), supports >> 6),
                                 np.uint64(1) << (supports & 63).astype(np.uint64))
            if r:
                syn = _native.unpack_rows(_syndromes_of(ctx, chk, packed_h, r, n, packed_e, count), r, dtype=np.uint8)
            else:
                syn = np.zeros((count, 0), dtype=np.uint8)
            if weights is not None or r == 0:
                keys = (syn.astype(np.uint64) @ weights) if r else np.zeros(count, dtype=np.uint64)
                collide = (np.unique(keys).size != count or np.isin(keys, seen).any() or
                           np.isin(keys, layer_seen).any())
                if collide:
                    return w - 1, table
                layer_seen = np.union1d(layer_seen, keys)
                layer_keys.extend(int(k) for k in keys)
            else:
                # bin_matrix.vec_to_int of every row (bits, row 0 most significant), as exact Python ints
                big = np.packbits(syn, axis=1, bitorder='big')
                shift = (-r) % 8
                keys = [int.from_bytes(row.tobytes(), 'big') >> shift for row in big]
                fresh = set(keys)
                if len(fresh) != count or any(k in table for k in keys) or not fresh.isdisjoint(layer_seen_big):
                    return w - 1, table
                layer_seen_big |= fresh
                layer_keys.extend(keys)
            layer_errs.append(errors)
        errs = (layer_errs[0] if len(layer_errs) == 1 else np.concatenate(layer_errs)) if layer_errs else np.zeros((0, n), dtype='int')
        table.update(zip(layer_keys, errs))
        if weights is not None:
            seen = np.union1d(seen, layer_seen)
    return n, table


def _syndromes_of(ctx, chk, packed_h, r, n, packed_e, count):
    """Packed syndromes (count x words(r)) of packed errors through a prepared check."""
    if chk.handle is None:
        return ctx.syndrome_batch(packed_h, r, n, packed_e, count)
    lde, lds = packed_e.shape[1], max(1, _native.words_for(r))
    e_buf = ctx.alloc(packed_e.nbytes).upload(packed_e)
    s_buf = ctx.alloc(count * lds * 8).zero()
    ctx.syndrome_dev(chk, e_buf, count, lde, s_buf, lds)
    out = s_buf.download((count, lds), "<u8")
    e_buf.free()
    s_buf.free()
    return out


def _gate_array(prog):
    """Gate array of a program: an (g, 3) array as is; a pyQuil Program (when pyQuil is importable) or any sequence
    of (name, qubits...) tuples is converted, with the reference's checks (css_code.py:741-755)."""
    if isinstance(prog, np.ndarray):
        return np.ascontiguousarray(prog, dtype=np.int32).reshape(-1, 3)
    gates = []
    for inst in getattr(prog, "instructions", prog):
        if isinstance(inst, (tuple, list)):
            name, qubits = inst[0], [int(q) for q in inst[1:]]
        else:
            if not hasattr(inst, "qubits") or not hasattr(inst, "name"):
                raise ValueError("program must only contain gates")
            if any(not hasattr(qubit, "index") for qubit in inst.qubits):
                raise ValueError("gate cannot have placeholders")
            name, qubits = inst.name, [qubit.index for qubit in inst.qubits]
        if name in ('H', GATE_H) and len(qubits) == 1:
            gates.append((GATE_H, qubits[0], 0))
        elif name in ('CNOT', GATE_CNOT) and len(qubits) == 2:
            gates.append((GATE_CNOT, qubits[0], qubits[1]))
        else:
            gates.append((-1, qubits[0] if qubits else 0, 0))        # refused when the walk reaches it
    return np.array(gates, dtype=np.int32).reshape(-1, 3)


def transform_stabilisers(mat, prog):
    """
    Conjugate the k x 2n stabiliser matrix [X | Z] through a Clifford program of H and CNOT gates, in place
    (css_code.py:737-755).  prog is a gate array as returned by CSSCode.noisy_encode_zero / noisy_encode_plus (or a
    pyQuil Program, or tuples ('H', q) / ('CNOT', c, t)).  Gates apply in order on the GPU (gf2_conjugate_gates).
    As in the reference, a gate is checked when it is reached: ValueError for a qubit outside [0, n) or a gate other
    than H / CNOT, NotImplementedError("only handles CSS codes") for an H on a qubit where a row has both X and Z; mat
    then holds what the reference leaves behind (all earlier gates, and the rows before the offending one swapped).
    """
    k, cols = mat.shape
    n = cols // 2
    gates = _gate_array(prog)
    packed = _native.pack_rows(mat)
    rc, stop = _native.default_context().conjugate_gates(packed, k, n, gates)
    _native.unpack_rows_into(packed, mat)
    if rc == _native.GF2_E_NOTCSS:
        q = int(gates[stop, 1])
        bad = int(np.flatnonzero((mat[:, q] == 1) & (mat[:, n + q] == 1))[0])
        upper = mat[:bad, [q, n + q]].copy()                       # css_code.py:761-767: rows above were swapped already
        mat[:bad, q], mat[:bad, n + q] = upper[:, 1], upper[:, 0]
        raise NotImplementedError("only handles CSS codes")
    if rc == _native.GF2_E_ARG:
        kind, a, b = (int(v) for v in gates[stop])
        if kind in (GATE_H, GATE_CNOT):
            raise ValueError("qubit index must be within [0, n)")
        raise ValueError("cannot conjugate gate {}".format(kind))


def conjugate_h_with_check_mat(mat, qubit):
    """css_code.py:757-767."""
    transform_stabilisers(mat, np.array([[GATE_H, qubit, 0]], dtype=np.int32))


def conjugate_cnot_with_check_mat(mat, control, target):
    """css_code.py:769-781."""
    transform_stabilisers(mat, np.array([[GATE_CNOT, control, target]], dtype=np.int32))


def swap_columns(mat, indices):
    """In-place swap of two columns (css_code.py:783-785).  The two columns travel to the GPU as packed
    bits (gf2_swap_columns), so their entries come back reduced modulo 2; the reference's callers
    (css_code.py:56-61) only pass binary matrices."""
    i, j = indices
    m, n = mat.shape
    if m == 0 or i == j:
        return
    packed = _native.pack_rows(mat[:, [i, j]])
    _native.default_context().swap_columns(packed, m, 2, 0, 1)
    mat[:, [i, j]] = _native.unpack_rows(packed, 2, dtype=mat.dtype)


def pauli_term_for_row(x_check, z_check):
    """
    The Pauli operator of a check-matrix row (css_code.py:787-807; Nielsen & Chuang 10.5.1): Y where
    both checks are set, else X or Z.  A pyQuil PauliTerm when pyQuil is importable, otherwise the text
    label with factors in qubit order ("X0*X3*X4*X5"; "I" for the identity).
    """
    x_check = np.asarray(x_check)
    z_check = np.asarray(z_check)
    n = x_check.size
    if not x_check.shape == (n,):
        raise ValueError("x_check has the wrong dimensions")
    if not z_check.shape == (n,):
        raise ValueError("z_check has the wrong dimensions")
    if _HAVE_PYQUIL:                                # pragma: no cover
        result = _ID()
        for i in range(n):
            if x_check[i] and z_check[i]:
                result *= _sY(i)
            elif x_check[i]:
                result *= _sX(i)
            elif z_check[i]:
                result *= _sZ(i)
        return result
    factors = []
    for i in range(n):
        if x_check[i] and z_check[i]:
            factors.append("Y%d" % i)
        elif x_check[i]:
            factors.append("X%d" % i)
        elif z_check[i]:
            factors.append("Z%d" % i)
    return "*".join(factors) if factors else "I"


def _normalize_packed(ctx, packed, r, n, offset):
    """gf2_normalize on packed rows, in place; returns the swaps; the reference's two exceptions (css_code.py:811-812, 825-826)."""
    if n < offset + r:
        raise ValueError("not enough columns")
    if r == 0:
        return []
    try:
        return ctx.normalize(packed, r, n, offset)
    except _native.GF2Error as err:
        if err.code == _native.GF2_E_DEPENDENT:
            raise InvalidCodeError("rows are not independent") from None
        if err.code == _native.GF2_E_COLUMNS:
            raise ValueError("not enough columns") from None
        raise


def normalize_parity_check(h, offset):
    """
    Put h into the form with an identity block in columns offset..offset+r-1 (css_code.py:809-836),
    swapping qubits (columns) where a pivot is missing.  Like the reference it works in place AND returns
    (h mod 2, list of (column, column) swaps).  The swaps and the result are bit-identical to the
    reference's: the GPU kernel performs the same operations in the same order.
    After the call `h` holds the reduced matrix (the reference leaves un-reduced integers congruent to
    it modulo 2).
    """
    r, n = h.shape
    if n < offset + r:
        raise ValueError("not enough columns")
    if r == 0:
        return np.mod(h, 2), []
    packed = _native.pack_rows(h)
    swaps = _normalize_packed(_native.default_context(), packed, r, n, offset)
    _native.unpack_rows_into(packed, h)                       # h is reduced from here on: np.mod(h, 2) is the same matrix again,
    # unpacked a second time (on several host threads: a 64 MiB h.copy() on one takes three times as long)
    if isinstance(h, np.ndarray) and h.dtype in (np.int64, np.uint8):
        return _native.unpack_rows(packed, n, dtype=h.dtype), swaps
    return h.copy(), swaps


def codes_equal(parity_check_1, parity_check_2):
    """Two parity checks define the same code iff their RREFs agree (css_code.py:838-844)."""
    parity_check_1 = np.asarray(parity_check_1)
    parity_check_2 = np.asarray(parity_check_2)
    if parity_check_1.shape != parity_check_2.shape:
        return False
    return np.array_equal(bin_matrix.reduced_row_echelon_form(parity_check_1),
                          bin_matrix.reduced_row_echelon_form(parity_check_2))


def is_doubly_even(mat):
    """Whether every row weight is a multiple of 4 (css_code.py:846-850).  Row weights by popcount on
    the GPU; for 0/1 matrices this is np.sum(mat, axis=1)."""
    mat = np.asarray(mat)
    m, n = mat.shape
    if m == 0:
        return True
    if np.issubdtype(mat.dtype, np.integer) and mat.size and (mat.min() < 0 or mat.max() > 1):
        # the reference sums the raw entries; outside {0,1} that is not a popcount
        return not np.any(np.mod(np.sum(mat, axis=1), 4))
    weights = _native.default_context().row_weights(_native.pack_rows(mat), m, n)
    return not np.any(weights % 4)
