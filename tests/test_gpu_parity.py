"""
Parity of the HIP path (through the C ABI) against the oracle and the reference-generated fixtures.
Needs a real MI355X: run with `pytest -m gpu`.  Bit-exact everywhere (integer work only).
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import c_oracle, cpu_ref
from quantum_css_codes_amd import _native, bin_matrix, css_code
from quantum_css_codes_amd.css_code import CSSCode
from quantum_css_codes_amd.errors import InvalidCodeError

pytestmark = pytest.mark.gpu


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    return _native.default_context()


class Route(object):
    """Forces one of the bit-identical implementations behind an entry point: routing flags of the process-wide context
    (gf2_ctx_set_flags; "GF2_MC_DENSE" names GF2_F_MC_DENSE of include/gf2hip.h)."""

    def __init__(self, context):
        self.context = context

    def force(self, name):
        self.context.set_flags(self.context.get_flags() | getattr(_native, "F_" + name[4:]))

    def release(self, name):
        self.context.set_flags(self.context.get_flags() & ~getattr(_native, "F_" + name[4:]))


@pytest.fixture
def route():
    context = _native.default_context()
    yield Route(context)
    context.set_flags(0)


@pytest.fixture
def sweep_k(ctx, request):
    """Route of the blocked RREF above 4096 rows: None = the default (round 5: four panels per sweep, rows streamed,
    launch_rref_sweeps_streamed), 0 = round 4's pair kernels (launch_rref_blocked)."""
    ctx.set_option(_native.OPT_RREF_SWEEP_K, request.param)
    yield request.param
    ctx.set_option(_native.OPT_RREF_SWEEP_K, None)


TALL_ROUTES = pytest.mark.parametrize("sweep_k", [None, 0], indirect=True, ids=["streamed-sweeps", "pair-kernels"])


def test_native_library_is_loaded(ctx):
    assert _native.lib().gf2_version() >= 100
    assert _native.device_count() >= 1
    with open("/proc/self/maps") as maps:
        assert "libgf2hip.so" in maps.read()


# ---- bin_matrix ------------------------------------------------------------------------------------------

def test_rref_kat():
    # test/test_bin_matrix.py:8-20
    mat = np.array([[1, 0, 1, 1, 0, 1, 0], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]], dtype='int')
    expected = np.array([[1, 0, 1, 0, 1, 0, 1], [0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 1, 1, 1, 1]], dtype='int')
    before = mat.copy()
    assert np.array_equal(bin_matrix.reduced_row_echelon_form(mat), expected)
    assert np.array_equal(mat, before)


def test_rref_golden(golden):
    tags = [str(i) for i in golden["rref_shape_ids"]] + ["def", "zero", "norows", "nonbin", "u8", "i8"]
    for tag in tags:
        a = golden["rref_in_" + tag]
        out = bin_matrix.reduced_row_echelon_form(a)
        assert out.dtype == golden["rref_out_" + tag].dtype, tag
        assert out.shape == a.shape
        assert np.array_equal(out, golden["rref_out_" + tag]), tag


@pytest.mark.parametrize("shape", [(1, 1), (2, 130), (64, 64), (65, 64), (130, 70), (300, 1000), (1025, 200),
                                   (1100, 1100)])
def test_rref_random_vs_oracle(shape, ctx):
    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    a = rng.integers(0, 2, shape)
    if shape[0] > 4:
        a[3] = a[0] ^ a[1]                      # some dependence
        a[:, shape[1] // 2] = 0                 # a pivot-free column
    packed = _native.pack_rows(a)
    pivots, rank = ctx.rref(packed, *shape)
    want, want_piv, want_rank = c_oracle.rref(c_oracle.pack_rows(a), *shape)
    assert rank == want_rank and list(pivots) == list(want_piv)
    assert np.array_equal(packed, want)


@pytest.mark.parametrize("shape", [(1500, 300, 1), (3000, 700, 1), (700, 256, 5), (5000, 640, 1)])
def test_rref_panels_that_need_several_rounds(shape, ctx):
    # the first column of every 64-column panel has its only entries far down the matrix, beyond the 128 rows a window holds:
    # the panel's first round resolves the other columns, a second one (with the coefficients of the first carried along)
    # this one; sparse upper rows make some windows short of pivots altogether
    m, n, batch = shape
    rng = np.random.default_rng(m + n)
    mats = []
    for b in range(batch):
        a = (rng.random((m, n)) < (0.5 if b % 2 == 0 else 0.03)).astype(np.uint8)
        a[: (2 * m) // 3, ::64] = 0
        a[m // 3: m // 2, :] = 0                         # a stretch of empty rows
        mats.append(a)
    packed = np.stack([_native.pack_rows(a) for a in mats])
    if batch == 1:
        pivots, rank = ctx.rref(packed[0], m, n)
        want, want_piv, want_rank = c_oracle.rref(c_oracle.pack_rows(mats[0]), m, n)
        assert rank == want_rank and list(pivots) == list(want_piv)
        assert np.array_equal(packed[0], want)
    else:
        pivots, ranks = ctx.rref_batch(packed, batch, m, n)
        for b in range(batch):
            want, want_piv, want_rank = c_oracle.rref(c_oracle.pack_rows(mats[b]), m, n)
            assert ranks[b] == want_rank and np.array_equal(packed[b], want)
            assert list(pivots[b, :want_rank]) == list(want_piv)


@pytest.mark.parametrize("shape", [(300, 700, 1), (200, 500, 4), (1100, 2500, 2), (2048, 4096, 3)])
def test_rref_first_pass_moves_rows_it_has_nothing_to_add_to(shape, ctx):
    # The blocked RREF's first trailing pass writes every row into the workspace copy (the row gather at the end then writes
    # straight into the caller's buffer): also where the first PAIR of panels finds no pivot at all (128 empty columns in front:
    # the pass has nothing to add and must still move the rows), where one matrix of a batch is like that and its neighbour is
    # not, where a matrix is all zeros, and where the last 32-word chunk of a row is a partial one.
    m, n, batch = shape
    rng = np.random.default_rng(m * 3 + n)
    mats = []
    for b in range(batch):
        a = rng.integers(0, 2, (m, n)).astype(np.uint8)
        if b % 2 == 0:
            a[:, :130] = 0
        if b == 3:
            a[:] = 0
        if b == 2:
            a[:, :1000] = 0                               # several pairs without a pivot, then the rest
        mats.append(a)
    packed = np.stack([_native.pack_rows(a) for a in mats])
    if batch == 1:
        pivots, rank = ctx.rref(packed[0], m, n)
        want, want_piv, want_rank = c_oracle.rref(c_oracle.pack_rows(mats[0]), m, n)
        assert rank == want_rank and list(pivots) == list(want_piv)
        assert np.array_equal(packed[0], want)
    else:
        pivots, ranks = ctx.rref_batch(packed, batch, m, n)
        for b in range(batch):
            want, want_piv, want_rank = c_oracle.rref(c_oracle.pack_rows(mats[b]), m, n)
            assert ranks[b] == want_rank and np.array_equal(packed[b], want), b
            assert list(pivots[b, :want_rank]) == list(want_piv)


@pytest.mark.parametrize("shape", [(300, 2500, 3), (2048, 4096, 2), (700, 4100, 5), (1500, 6000, 1), (256, 2112, 9), (2000, 2100, 2)])
def test_rref_mixed_batches_under_either_look_ahead_flag(shape, ctx, route):
    # Batches of matrices with at most 2048 rows (panels held in registers), under both look-ahead flags (they only change the
    # route of matrices with more than 8192 rows; a look-ahead for these batches was built and measured slower):
    # same reduced forms, pivots and ranks as the oracle's -- dense and sparse matrices in one batch, empty leading columns
    # (pairs without a pivot), a rank-deficient matrix, a last chunk of 32 words that is partial.
    m, n, batch = shape
    rng = np.random.default_rng(m * 11 + n + batch)
    mats = []
    for b in range(batch):
        a = (rng.random((m, n)) < (0.5 if b % 3 != 1 else 0.02)).astype(np.uint8)
        if b % 4 == 2:
            a[:, :200] = 0
        if b % 5 == 3:
            a[m // 2:] = a[: m - m // 2]                     # every row twice: rank at most m / 2
        mats.append(a)
    want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
    for flag in ("GF2_RREF_LOOKAHEAD", "GF2_RREF_NO_LOOKAHEAD"):
        route.force(flag)
        packed = np.stack([_native.pack_rows(a) for a in mats])
        pivots, ranks = ctx.rref_batch(packed, batch, m, n)
        route.release(flag)
        for b in range(batch):
            assert ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]), (flag, b)
            assert list(pivots[b, :want[b][2]]) == list(want[b][1]), (flag, b)


def test_rref_big512(golden):
    a = np.random.default_rng(1024).integers(0, 2, (512, 1024)).astype(np.int64)
    out = bin_matrix.reduced_row_echelon_form(a)
    assert sha(_native.pack_rows(out)) == str(golden["big512_rref_sha"])
    assert bin_matrix.rank(a) == int(golden["big512_rank"])


def test_rref_full_size_config4(golden, ctx):
    # BASELINE.json configs[3]: random 2048 x 4096; digest of the reference's own output
    a = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.int64)
    packed = _native.pack_rows(a)
    pivots, rank = ctx.rref(packed, 2048, 4096)
    assert rank == int(golden["big4096_rank"])
    assert sha(packed) == str(golden["big4096_rref_sha"])
    # idempotence: the RREF of an RREF is itself
    again = packed.copy()
    ctx.rref(again, 2048, 4096)
    assert np.array_equal(again, packed)


def test_rref_batch(ctx):
    rng = np.random.default_rng(3)
    mats = rng.integers(0, 2, (5, 70, 150))
    packed = np.stack([_native.pack_rows(m) for m in mats])
    pivots, ranks = ctx.rref_batch(packed, 5, 70, 150)
    for b in range(5):
        want, want_piv, want_rank = c_oracle.rref(c_oracle.pack_rows(mats[b]), 70, 150)
        assert ranks[b] == want_rank and np.array_equal(packed[b], want)
        assert list(pivots[b, :want_rank]) == list(want_piv)


def test_nullspace(golden, steane_h, ctx):
    for mat in (steane_h, golden["rref_in_4"], golden["rref_in_def"], golden["rref_in_5"], golden["rref_in_7"]):
        mat = np.asarray(mat) & 1
        got = bin_matrix.nullspace(mat)
        assert np.array_equal(got, cpu_ref.nullspace(mat))
        assert not np.any(np.mod(mat @ got.T, 2))
    a = np.random.default_rng(5).integers(0, 2, (4, 6))
    h = np.hstack([np.identity(4, dtype=int), a])
    assert np.array_equal(bin_matrix.nullspace(h), np.hstack([a.T, np.identity(6, dtype=int)]))
    assert bin_matrix.nullspace(np.zeros((0, 5), dtype=int)).shape == (5, 5)


def test_nullspace_full_size(ctx):
    a = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.uint8)
    packed = _native.pack_rows(a)
    basis = ctx.nullspace(packed, 2048, 4096)
    assert np.array_equal(basis, c_oracle.nullspace(packed, 2048, 4096))
    # H . N^T = 0 at full size, on the GPU
    assert not np.any(ctx.matmul_abt(packed, 2048, basis, basis.shape[0], 4096))


# ---- css_code free functions -------------------------------------------------------------------------------

def test_normalize_golden(golden):
    for tag in golden["norm_tags"]:
        tag = str(tag)
        work = np.array(golden["norm_in_" + tag])
        out, swaps = css_code.normalize_parity_check(work, int(golden["norm_off_" + tag]))
        assert np.array_equal(out, golden["norm_out_" + tag]), tag
        assert np.array_equal(work, np.mod(golden["norm_mut_" + tag], 2)), tag      # in place, mod 2
        assert [tuple(s) for s in swaps] == [tuple(s) for s in golden["norm_swaps_" + tag]], tag
    with pytest.raises(InvalidCodeError, match="rows are not independent"):
        css_code.normalize_parity_check(np.array(golden["norm_dep_in"]), 0)
    with pytest.raises(ValueError, match="not enough columns"):
        css_code.normalize_parity_check(np.zeros((3, 5), dtype=np.int64), 3)


@pytest.mark.parametrize("case", [(40, 100, 0, 11), (40, 110, 60, 12), (130, 300, 64, 13), (1030, 1200, 100, 14)])
def test_normalize_random_vs_oracle(case, ctx):
    r, n, off, seed = case
    rng = np.random.default_rng(seed)
    for _ in range(50):
        h = rng.integers(0, 2, (r, n))
        h[:, rng.integers(off, off + r, 4)] = 0          # force column swaps
        packed = _native.pack_rows(h)
        rc, want, want_swaps = c_oracle.normalize(packed, r, n, off)
        if rc == 0:
            break
    assert rc == 0 and len(want_swaps) > 0
    swaps = ctx.normalize(packed, r, n, off)
    assert swaps == want_swaps
    assert np.array_equal(packed, want)


@pytest.fixture(scope="module")
def config4_golden():
    """Digests of the reference's own CSSCode(H1, H2) at n = 4096 (tests/golden/make_golden_config4.py)."""
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "config4_golden.npz"), allow_pickle=False) as data:
        return {k: data[k] for k in data.files}


def test_css_code_config4_against_the_reference_constructor(config4_golden, ctx):
    # BASELINE.json configs[3]/[4]: the code the benchmark times.  H2 = first 2047 rows of nullspace(H1); both standard
    # forms, both swap lists (css_code.py:51-61), the logical operators and the gate set as the reference's constructor
    # produced them
    g = config4_golden
    h1 = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.int64)
    assert sha(_native.pack_rows(h1)) == str(g["h1_in_sha"])
    h2 = bin_matrix.nullspace(h1)[:2047]
    assert sha(_native.pack_rows(h2)) == str(g["h2_in_sha"])
    code = CSSCode(h1, h2, max_table_weight=0)
    assert [code.n, code.k, code.r_1, code.r_2] == [int(v) for v in g["nk_r1_r2"]]
    assert sha(_native.pack_rows(code.parity_check_c1)) == str(g["c1_sha"])
    assert sha(_native.pack_rows(code.parity_check_c2)) == str(g["c2_sha"])
    assert np.array_equal(code.z_operator_matrix(), g["zop"]) and np.array_equal(code.x_operator_matrix(), g["xop"])
    assert sorted(code._transversal_gates) == [str(v) for v in g["gates"]]
    # the constructor's steps one by one: swap lists of both normalisations
    w1, w2 = np.array(h1), np.array(h2)
    out1, swaps1 = css_code.normalize_parity_check(w1, 0)
    for pair in swaps1:
        css_code.swap_columns(w2, pair)
    out2, swaps2 = css_code.normalize_parity_check(w2, 2048)
    for pair in swaps2:
        css_code.swap_columns(out1, pair)
    assert [tuple(p) for p in swaps1] == [tuple(int(v) for v in p) for p in g["swaps_c1"]]
    assert [tuple(p) for p in swaps2] == [tuple(int(v) for v in p) for p in g["swaps_c2"]]
    assert np.array_equal(out1, code.parity_check_c1) and np.array_equal(out2, code.parity_check_c2)
    # syndrome products of the standard forms (css_code.py:728)
    e = _native.pack_rows(np.random.default_rng(78).integers(0, 2, (16, 4096)).astype(np.uint8))
    s1 = ctx.syndrome_batch(_native.pack_rows(code.parity_check_c1), 2048, 4096, e, 16)
    s2 = ctx.syndrome_batch(_native.pack_rows(code.parity_check_c2), 2047, 4096, e, 16)
    assert sha(s1) == str(g["syn_c1_sha"]) and sha(s2) == str(g["syn_c2_sha"])


def test_normalize_full_size_config4(golden, ctx):
    a = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.int64)
    packed = _native.pack_rows(a)
    swaps = ctx.normalize(packed, 2048, 4096, 0)
    assert swaps == [tuple(s) for s in golden["big4096_norm_swaps"]]
    assert sha(packed) == str(golden["big4096_norm_sha"])


def test_swap_columns(ctx):
    rng = np.random.default_rng(9)
    mat = rng.integers(0, 2, (37, 200))
    want = mat.copy()
    for pair in ((0, 199), (63, 64), (5, 6), (70, 70), (130, 2)):
        css_code.swap_columns(mat, pair)
        cpu_ref.swap_columns(want, pair)
        assert np.array_equal(mat, want)
    packed = _native.pack_rows(mat)
    ctx.swap_columns(packed, 37, 200, 64, 191)
    cpu_ref.swap_columns(want, (64, 191))
    assert np.array_equal(_native.unpack_rows(packed, 200), want)


def test_codes_equal_doubly_even(golden):
    a, b, c = golden["ceq_a"], golden["ceq_b"], golden["ceq_c"]
    got = [css_code.codes_equal(a, b), css_code.codes_equal(a, c), css_code.codes_equal(a, a[:5])]
    assert got == [bool(v) for v in golden["ceq_res"]]
    de = golden["de_in"]
    assert [css_code.is_doubly_even(de[i:i + 1]) for i in range(10)] == [bool(v) for v in golden["de_rows"]]
    # test/test_css_code.py:120-143
    a = np.array([[0] * 8, [0, 0, 1, 1, 0, 1, 1, 0], [1, 1, 1, 0, 0, 0, 0, 1], [1] * 8])
    b = np.array([[0] * 8, [0, 0, 1, 1, 0, 1, 1, 0], [0, 1, 1, 0, 0, 0, 0, 1], [1] * 8])
    assert css_code.is_doubly_even(a) and not css_code.is_doubly_even(b)
    wide = np.random.default_rng(2).integers(0, 2, (70, 1000))
    assert css_code.is_doubly_even(wide) == cpu_ref.is_doubly_even(wide)


def test_matmul_abt(ctx):
    rng = np.random.default_rng(21)
    for (ra, rb, n) in ((3, 3, 7), (70, 130, 200), (5, 300, 4100)):
        a, b = rng.integers(0, 2, (ra, n)), rng.integers(0, 2, (rb, n))
        got = _native.unpack_rows(ctx.matmul_abt(_native.pack_rows(a), ra, _native.pack_rows(b), rb, n), rb)
        assert np.array_equal(got, np.mod(a @ b.T, 2))


# ---- syndromes -------------------------------------------------------------------------------------------------

def test_syndromes_golden(golden):
    for tag in ("steane", "rm15", "r64x128", "r70x200"):
        h, e, s = golden["syn_h_" + tag], golden["syn_e_" + tag], golden["syn_s_" + tag]
        assert np.array_equal(css_code.syndrome_batch(h, e), s), tag


@pytest.mark.parametrize("shape", [(1, 65, 3), (64, 64, 100), (65, 127, 1000), (200, 1000, 4097), (130, 4096, 300),
                                   (70, 4100, 50), (64, 9000, 20), (3, 7, 5000), (10, 15, 5000), (64, 64, 5000)])
def test_syndrome_batch_vs_oracle(shape, ctx):
    r, n, batch = shape
    rng = np.random.default_rng(r * 31 + n)
    h = _native.pack_rows(rng.integers(0, 2, (r, n)))
    e = _native.pack_rows(rng.integers(0, 2, (batch, n)))
    got = ctx.syndrome_batch(h, r, n, e, batch)
    assert np.array_equal(got, c_oracle.syndrome_batch(h, r, n, e, batch))


def test_syndrome_batch_full_size(golden, ctx):
    a = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.int64)
    e = np.random.default_rng(77).integers(0, 2, (32, 4096)).astype(np.int64)
    h = _native.pack_rows(a)
    got = ctx.syndrome_batch(h, 2048, 4096, _native.pack_rows(e), 32)
    assert sha(got) == str(golden["big4096_syn_sha"])          # the reference's own products
    # linearity at full size: S(e1 ^ e2) = S(e1) ^ S(e2)
    rng = np.random.default_rng(5)
    e1 = rng.integers(0, 2**63, (5000, 64), dtype=np.int64).view(np.uint64)
    e2 = rng.integers(0, 2**63, (5000, 64), dtype=np.int64).view(np.uint64)
    s1, s2 = ctx.syndrome_batch(h, 2048, 4096, e1, 5000), ctx.syndrome_batch(h, 2048, 4096, e2, 5000)
    assert np.array_equal(ctx.syndrome_batch(h, 2048, 4096, e1 ^ e2, 5000), s1 ^ s2)
    assert np.array_equal(s1[:200], c_oracle.syndrome_batch(h, 2048, 4096, e1[:200], 200))


@pytest.mark.parametrize("shape", [(3, 7, 1000), (10, 15, 64 * 300 + 5), (4, 15, 63), (20, 31, 4096), (64, 64, 1000)])
def test_syndrome_bit_sliced(shape, ctx):
    r, n, batch = shape
    rng = np.random.default_rng(r + n)
    hm, em = rng.integers(0, 2, (r, n)), rng.integers(0, 2, (batch, n))
    h = _native.pack_rows(hm)
    e_sliced = _native.pack_rows(em.T)                 # n x words(batch)
    got = ctx.syndrome_batch_sliced(h, r, n, e_sliced, batch)
    want = np.mod(em @ hm.T, 2)                        # batch x r
    assert np.array_equal(_native.unpack_rows(got, batch).T, want)


# ---- CSSCode ---------------------------------------------------------------------------------------------------

def test_steane_kat(steane_h):
    # test/test_css_code.py:13-59, 108-118
    code = CSSCode(steane_h, steane_h)
    assert np.array_equal(code.parity_check_c1[:, 0:3], np.identity(3))
    assert np.array_equal(code.parity_check_c2[:, 3:6], np.identity(3))
    assert code.stabilisers() == ["X0*X3*X4*X5", "X1*X3*X5*X6", "X2*X4*X5*X6",
                                  "Z0*Z2*Z3*Z6", "Z0*Z1*Z4*Z6", "Z0*Z1*Z2*Z5"]
    assert code.z_operators() == ["Z1*Z2*Z6"]
    assert code.x_operators() == ["X3*X4*X6"]
    assert code.y_operators() == ["Z1*Z2*X3*X4*Y6"]
    for gate in ('I', 'CNOT', 'H', 'CZ', 'S'):
        assert code.is_transversal(gate)
    t, table = css_code.syndrome_table(code.parity_check_c1)
    assert t == 1 and len(table) == 8
    for s, e in table.items():
        assert s == bin_matrix.vec_to_int(np.mod(np.matmul(code.parity_check_c1, e), 2))


@pytest.mark.parametrize("tag", ["steane", "rm15"])
def test_css_code_golden(golden, tag):
    code = CSSCode(golden[tag + "_in1"], golden[tag + "_in2"])
    assert np.array_equal(code.parity_check_c1, golden[tag + "_h1"])
    assert np.array_equal(code.parity_check_c2, golden[tag + "_h2"])
    assert [code.n, code.k, code.t, code.r_1, code.r_2] == list(golden[tag + "_nktr"])
    assert sorted(code._transversal_gates) == [str(g) for g in golden[tag + "_gates"]]
    assert np.array_equal(code.z_operator_matrix(), golden[tag + "_zop"])
    assert np.array_equal(code.x_operator_matrix(), golden[tag + "_xop"])
    for which, tab in (("c1", code._c1_syndromes), ("c2", code._c2_syndromes)):
        keys, errs = golden["%s_%s_keys" % (tag, which)], golden["%s_%s_errs" % (tag, which)]
        assert [int(k) for k in tab.keys()] == [int(k) for k in keys]
        for k, e in zip(keys, errs):
            assert np.array_equal(tab[int(k)], e)


def test_css_code_errors(steane_h):
    with pytest.raises(ValueError, match="same code word length"):
        CSSCode(steane_h, steane_h[:, :6])
    with pytest.raises(ValueError, match="C_1 parity check matrix must be binary"):
        CSSCode(steane_h * 2, steane_h)
    with pytest.raises(ValueError, match="C_2 parity check matrix must be binary"):
        CSSCode(steane_h, steane_h * 3)
    with pytest.raises(ValueError, match="dual code must be a subspace"):
        CSSCode(steane_h, np.array([[1, 0, 0, 0, 0, 0, 0]]))
    with pytest.raises(InvalidCodeError):
        CSSCode(np.array([[1, 1, 1, 1]]), np.array([[1, 1, 1, 1]]))


def test_css_code_mid_size_vs_oracle():
    # random dual pair, n = 96: H2 = first rows of nullspace(H1); capped tables (reference cannot finish)
    rng = np.random.default_rng(96)
    h1 = rng.integers(0, 2, (48, 96))
    while bin_matrix.rank(h1) < 48:
        h1 = rng.integers(0, 2, (48, 96))
    h2 = bin_matrix.nullspace(h1)[:47]
    code = CSSCode(h1, h2, max_table_weight=1)
    want = cpu_ref.CSSCode(h1, h2, max_table_weight=1)
    assert np.array_equal(code.parity_check_c1, want.parity_check_c1)
    assert np.array_equal(code.parity_check_c2, want.parity_check_c2)
    assert (code.t, code.k) == (want.t, want.k)
    assert np.array_equal(code.z_operator_matrix(), want.z_operator_matrix())
    assert np.array_equal(code.x_operator_matrix(), want.x_operator_matrix())
    assert code.stabilisers() == want.stabiliser_labels()
    assert list(code._c2_syndromes.keys()) == [int(k) for k in want._c2_syndromes.keys()]


# ---- Monte-Carlo -------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [7, 64, 70, 200, 512, 513, 1000, 4095, 4096, 5000])
def test_sampler_vs_oracle(n, ctx):
    # n >= 512 with rows of at most 64 words takes the block-wise kernel (queued error words), the rest the word-per-lane one
    count, lde = 300, max(1, _native.words_for(n))
    for (px, py, pz) in ((0.01, 0.01, 0.01), (0.2, 0.1, 0.3), (0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.5, 0.5)):
        ex_buf, ez_buf = ctx.alloc(count * lde * 8), ctx.alloc(count * lde * 8)
        ctx.sample_errors_dev(n, 99, 1000, count, px, py, pz, ex_buf, ez_buf, lde)
        ex, ez = ex_buf.download((count, lde), "<u8"), ez_buf.download((count, lde), "<u8")
        want_x, want_z = c_oracle.sample_errors(n, 99, 1000, count, px, py, pz)
        assert np.array_equal(ex, want_x) and np.array_equal(ez, want_z), (n, px, py, pz)
        ex_buf.free()
        ez_buf.free()


@pytest.mark.parametrize("case", [(1000, 20, 1), (3000, 64, 13), (4096, 64, 64), (640, 10, 7)])
def test_sampler_block_kernel_padded_rows_and_ragged_counts(case, ctx):
    n, lde, count = case
    words = _native.words_for(n)
    ex_buf, ez_buf = ctx.alloc(count * lde * 8), ctx.alloc(count * lde * 8)
    ex_buf.upload(np.full((count, lde), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))     # pad words must come back zero
    ez_buf.upload(np.full((count, lde), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    ctx.sample_errors_dev(n, 5, 77, count, 0.02, 0.01, 0.03, ex_buf, ez_buf, lde)
    ex, ez = ex_buf.download((count, lde), "<u8"), ez_buf.download((count, lde), "<u8")
    want_x, want_z = c_oracle.sample_errors(n, 5, 77, count, 0.02, 0.01, 0.03)
    assert np.array_equal(ex[:, :words], want_x) and np.array_equal(ez[:, :words], want_z)
    assert not ex[:, words:].any() and not ez[:, words:].any()
    ex_buf.free(), ez_buf.free()


def test_monte_carlo_steane_config2(steane_h):
    # BASELINE.json configs[1]: Steane, 10^6 random Pauli errors, bit-exact vs CPU
    code = CSSCode(steane_h, steane_h)
    h1, h2 = c_oracle.pack_rows(code.parity_check_c1), c_oracle.pack_rows(code.parity_check_c2)
    for (px, py, pz) in ((0.01 / 3, 0.01 / 3, 0.01 / 3), (0.25, 0.25, 0.25)):
        got = code.monte_carlo(10**6, px, py, pz, seed=0xC55C0DE)
        hz, hx = c_oracle.mc(h1, 3, h2, 3, 7, 0xC55C0DE, 0, 10**6, px, py, pz, 0)
        assert got['mode'] == 'full'
        assert np.array_equal(got['hist_z'], hz) and np.array_equal(got['hist_x'], hx)


def test_monte_carlo_rm15(rm15):
    # BASELINE.json configs[2] at its full size: 10^7 samples, every bin against the oracle
    code = CSSCode(*rm15)
    h1, h2 = c_oracle.pack_rows(code.parity_check_c1), c_oracle.pack_rows(code.parity_check_c2)
    got = code.monte_carlo(10**7, 0.03, 0.01, 0.02, seed=15, first_sample=12345)
    hz, hx = c_oracle.mc(h1, 4, h2, 10, 15, 15, 12345, 10**7, 0.03, 0.01, 0.02, 0)
    assert got['hist_z'].size == 16 and got['hist_x'].size == 1024
    assert np.array_equal(got['hist_z'], hz) and np.array_equal(got['hist_x'], hx)
    # weight mode and shard-independence: two halves add up to the whole
    whole = code.monte_carlo(100000, 0.03, 0.01, 0.02, seed=15, mode='weight')
    a = code.monte_carlo(40000, 0.03, 0.01, 0.02, seed=15, mode='weight')
    b = code.monte_carlo(60000, 0.03, 0.01, 0.02, seed=15, first_sample=40000, mode='weight')
    assert np.array_equal(whole['hist_z'], a['hist_z'] + b['hist_z'])
    assert np.array_equal(whole['hist_x'], a['hist_x'] + b['hist_x'])


def test_monte_carlo_n4096_weight_histograms(ctx):
    # BASELINE.json configs[4] at reduced sample count: dense random 2048/2047 x 4096 checks
    rng = np.random.default_rng(4096)
    h1 = _native.pack_rows(rng.integers(0, 2, (2048, 4096)))
    h2 = _native.pack_rows(rng.integers(0, 2, (2047, 4096)))
    c1, c2 = ctx.check_create(h1, 2048, 4096), ctx.check_create(h2, 2047, 4096)
    count = 6000
    hz, hx = ctx.mc_run(c1, c2, 0xC55C0DE, 10**6, count, 0.01 / 3, 0.01 / 3, 0.01 / 3, _native.HIST_WEIGHT)
    wz, wx = c_oracle.mc(h1, 2048, h2, 2047, 4096, 0xC55C0DE, 10**6, count, 0.01 / 3, 0.01 / 3, 0.01 / 3, 1)
    assert np.array_equal(hz, wz) and np.array_equal(hx, wx)
    assert int(hz.sum()) == count and int(hx.sum()) == count


def test_syndrome_table_on_the_config4_checks_and_what_the_reference_returns(config4_golden):
    # BASELINE.json configs[3]: "syndrome_table capped at weight <= 1" at n = 4096.  The drop-in's keys are exact integers of 2048
    # (2047) bits; expected here from bin_matrix.vec_to_int restated on Python ints (oracle/cpu_ref.py on object arrays) of every
    # column, in the reference's insertion order.
    #   parity_check_c1 = [I | A]: 4097 distinct keys, the capped search answers t = 1 with the weight-0 and weight-1 classes;
    #   parity_check_c2 = first 2047 rows of [A^T | I], standard form [A' | I | c]: its last column c is ZERO (the row that held
    #   the last identity column's 1 is the one left out), so e_4095 has the zero error's syndrome: t = 0 and the table {0: 0},
    #   exactly (the code has a weight-1 logical operator).
    # Beside it what the REFERENCE itself returns (tests/golden/make_golden_config4.py ran its syndrome_table): t = 0 and {0: 0} for
    # BOTH checks -- for c1 only because its int64 keys wrap (bin_matrix.py:40-43): the key of e_0, 2^2047, keeps its low 64 bits = 0
    # (DESIGN.md section 5).
    import bench
    code, _, _ = bench.build_code()
    g = config4_golden
    for name, h in (("c1", code.parity_check_c1), ("c2", code.parity_check_c2)):
        assert int(g["ref_table_%s_t" % name]) == 0 and [int(k) for k in g["ref_table_%s_keys" % name]] == [0]
        t, table = css_code.syndrome_table(h, max_weight=1)
        hobj = np.array(h, dtype=object)
        col_keys = [int(cpu_ref.vec_to_int(hobj[:, j])) for j in range(h.shape[1])]
        if name == "c1":
            assert len(set([0] + col_keys)) == 4097                   # exact keys do not collide at weight 1 ...
            assert t == 1 and list(table.keys()) == [0] + col_keys    # ... so the capped search answers t = 1
            errs = np.array(list(table.values()))
            assert not errs[0].any() and np.array_equal(errs[1:], np.identity(h.shape[1], dtype=int))
            # the reference's keys are the low 64 bits of the exact ones: the first weight-1 key is 0 there, the zero error's
            assert col_keys[0] == 1 << 2047 and col_keys[0] & 0xFFFFFFFFFFFFFFFF == 0
        else:
            assert col_keys[4095] == 0 and not h[:, 4095].any()
            assert t == 0 and list(table.keys()) == [0] and not table[0].any()


def test_monte_carlo_on_the_config4_code_itself(ctx, route):
    # BASELINE.json configs[4] as bench.py builds it -- H2 the dual of H1, both in the reference's standard form (digests of
    # config4_golden.npz asserted by build_code) -- not a stand-in with independent random checks: 2^17 + 4097 samples through
    # gf2_mc_run's default route (record sampler -> gather -> combine -> misfits), through the packed-row route (sampler ->
    # compact -> gather -> combine -> redo) and through the resident-error entry point the benchmark times, against the oracle
    import bench
    code, h1, h2 = bench.build_code()
    c1, c2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
    count, first, p = (1 << 17) + 4097, 3 * 10**9, bench.P_TOTAL / 3
    want = c_oracle.mc(h1, bench.R1, h2, bench.R2, bench.N_QUBITS, bench.SEED, first, count, p, p, p, 1)
    assert int(want[0].sum()) == count and int(want[1].sum()) == count
    got = ctx.mc_run(c1, c2, bench.SEED, first, count, p, p, p, _native.HIST_WEIGHT)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), "record-sampler route"
    route.force("GF2_MC_ROWS")
    rows = ctx.mc_run(c1, c2, bench.SEED, first, count, p, p, p, _native.HIST_WEIGHT)
    route.release("GF2_MC_ROWS")
    assert np.array_equal(rows[0], want[0]) and np.array_equal(rows[1], want[1]), "packed-row route"
    lde = _native.words_for(bench.N_QUBITS)
    ex, ez = ctx.alloc(count * lde * 8), ctx.alloc(count * lde * 8)
    ctx.sample_errors_dev(bench.N_QUBITS, bench.SEED, first, count, p, p, p, ex, ez, lde)
    hz, hx = ctx.alloc((bench.R1 + 1) * 8).zero(), ctx.alloc((bench.R2 + 1) * 8).zero()
    ctx.syndrome_sparse_dev(c1, ez, count, lde, None, 0, hz, bench.R1 + 1)
    ctx.syndrome_sparse_dev(c2, ex, count, lde, None, 0, hx, bench.R2 + 1)
    assert np.array_equal(hz.download((bench.R1 + 1,), np.uint64), want[0]), "resident errors, slab pipeline, H1.e_z"
    assert np.array_equal(hx.download((bench.R2 + 1,), np.uint64), want[1]), "resident errors, slab pipeline, H2.e_x"
    # ... and the syndromes of every one of those samples, stored by the same pipeline (SURVEY.md 8d's read-errors-write-syndromes
    # variant, what bench.py's secondary.read_write_1536B times), word for word against the oracle's product (css_code.py:728)
    e_z, e_x = ez.download((count, lde), "<u8"), ex.download((count, lde), "<u8")
    for chk, hh, rr, e_host, e_dev in ((c1, h1, bench.R1, e_z, ez), (c2, h2, bench.R2, e_x, ex)):
        lds = _native.words_for(rr)
        s_dev = ctx.alloc(count * lds * 8).zero()
        ctx.syndrome_sparse_dev(chk, e_dev, count, lde, s_dev, lds)
        assert np.array_equal(s_dev.download((count, lds), "<u8"), c_oracle.syndrome_batch(hh, rr, bench.N_QUBITS, e_host, count)), rr
        s_dev.free()
    for b in (ex, ez, hz, hx):
        b.free()


def test_monte_carlo_config4_at_full_size_shards_add_up(ctx):
    # BASELINE.json configs[4] at its full size: 10^8 samples of the config-4 code, once as one run and once as the eight
    # contiguous shards the 8-GPU job cuts them into (montecarlo.shard_range), summed as the all-reduce sums them.  Sample i is a
    # function of (seed, i) alone, so the two must agree bin for bin; every sample lands in exactly one bin of each histogram;
    # and the first 2^16 samples of shard 5 are the oracle's (the size-independent property + a prefix the oracle can afford).
    import bench
    from quantum_css_codes_amd import montecarlo
    code, h1, h2 = bench.build_code()
    c1, c2 = ctx.check_create(h1, bench.R1, bench.N_QUBITS), ctx.check_create(h2, bench.R2, bench.N_QUBITS)
    total, first, p = 10**8, 7 * 10**11 + 3, bench.P_TOTAL / 3
    whole = ctx.mc_run(c1, c2, bench.SEED, first, total, p, p, p, _native.HIST_WEIGHT)
    assert int(whole[0].sum()) == total and int(whole[1].sum()) == total
    sum_z, sum_x = np.zeros(bench.R1 + 1, np.uint64), np.zeros(bench.R2 + 1, np.uint64)
    for rank in range(8):
        shard_first, shard_count = montecarlo.shard_range(first, total, rank, 8)
        hz, hx = ctx.mc_run(c1, c2, bench.SEED, shard_first, shard_count, p, p, p, _native.HIST_WEIGHT)
        assert int(hz.sum()) == shard_count and int(hx.sum()) == shard_count
        sum_z += hz
        sum_x += hx
    assert np.array_equal(sum_z, whole[0]) and np.array_equal(sum_x, whole[1])
    shard_first, _ = montecarlo.shard_range(first, total, 5, 8)
    got = ctx.mc_run(c1, c2, bench.SEED, shard_first, 1 << 16, p, p, p, _native.HIST_WEIGHT)
    want = c_oracle.mc(h1, bench.R1, h2, bench.R2, bench.N_QUBITS, bench.SEED, shard_first, 1 << 16, p, p, p, 1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


# ---- tiled device layout -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(70, 1), (130, 64), (200, 65), (4096, 300), (4100, 130)])
def test_retile_and_tiled_sampler(shape, ctx):
    n, batch = shape
    rng = np.random.default_rng(n + batch)
    lde = _native.words_for(n)
    e = _native.pack_rows(rng.integers(0, 2, (batch, n)))
    src = ctx.alloc(e.nbytes).upload(e)
    words = _native.tiled_words(n, batch)
    dst = ctx.alloc(words * 8).zero()
    ctx.retile_dev(src, batch, lde, n, dst)
    got = dst.download((words,), "<u8")
    assert np.array_equal(got, _native.tile_rows(e, n))
    assert np.array_equal(_native.untile_rows(got, n, batch)[:, :lde], e)
    # the sampler writes the same samples in either layout
    ex_t, ez_t = ctx.alloc(words * 8), ctx.alloc(words * 8)
    ctx.sample_errors_dev(n, 5, 77, batch, 0.05, 0.02, 0.03, ex_t, ez_t, 0, _native.LAYOUT_TILED)
    want_x, want_z = c_oracle.sample_errors(n, 5, 77, batch, 0.05, 0.02, 0.03)
    assert np.array_equal(ex_t.download((words,), "<u8"), _native.tile_rows(want_x, n))
    assert np.array_equal(ez_t.download((words,), "<u8"), _native.tile_rows(want_z, n))


@pytest.mark.parametrize("case", [(100, 300, 0, 500), (100, 300, 100, 130), (64, 200, 17, 64), (130, 4096, 64, 200),
                                  (2047, 4096, 2048, 300), (1000, 2600, None, 100), (70, 9000, 128, 50)])
def test_syndrome_identity_block_and_sparse_slabs(case, ctx):
    # standard forms [.. I ..] (css_code.py:51-61) take the identity columns from the error word; slabs with
    # all-zero column pairs skip them
    r, n, ioff, batch = case
    rng = np.random.default_rng(r + n)
    hm = rng.integers(0, 2, (r, n))
    if ioff is not None:
        hm[:, ioff:ioff + r] = np.identity(r, dtype=int)
    else:
        hm[:, 512:1900] = 0                                  # zero column pairs in every slab
        hm[64:128, :] = 0                                    # an all-zero slab
    h = _native.pack_rows(hm)
    e = _native.pack_rows(rng.integers(0, 2, (batch, n)))
    got = ctx.syndrome_batch(h, r, n, e, batch)
    assert np.array_equal(got, c_oracle.syndrome_batch(h, r, n, e, batch))


def test_tiled_device_path_slab_major_output(ctx):
    # device-native layouts end to end: tiled errors in, slab-major syndromes out, histogram of the slab-major buffer
    r, n, batch = 200, 1000, 777
    rng = np.random.default_rng(8)
    h = _native.pack_rows(rng.integers(0, 2, (r, n)))
    e = _native.pack_rows(rng.integers(0, 2, (batch, n)))
    chk = ctx.check_create(h, r, n)
    e_buf = ctx.alloc(_native.tiled_words(n, batch) * 8).upload(_native.tile_rows(e, n))
    stride = 832                                           # >= batch
    s_buf = ctx.alloc(chk.slabs * stride * 8).zero()
    ctx.syndrome_dev(chk, e_buf, batch, 0, s_buf, stride, _native.LAYOUT_TILED)
    got = s_buf.download((chk.slabs, stride), "<u8")[:, :batch].T
    want = c_oracle.syndrome_batch(h, r, n, e, batch)
    assert np.array_equal(got, want)
    hist = ctx.alloc((r + 1) * 8).zero()
    ctx.histogram_dev(s_buf, batch, stride, r, _native.HIST_WEIGHT, hist, r + 1, _native.LAYOUT_TILED)
    assert np.array_equal(hist.download((r + 1,), np.uint64), c_oracle.histogram(want, batch, r, 1, r + 1))


# ---- sparse-error kernel -------------------------------------------------------------------------------------------

@pytest.mark.parametrize("case", [(100, 300, None, 200, 0.02), (100, 300, 100, 130, 0.05), (64, 200, 17, 64, 0.5),
                                  (2048, 4096, 0, 300, 0.007), (2047, 4096, 2048, 300, 0.007), (2047, 4096, 2048, 40, 0.5),
                                  (3000, 5000, None, 70, 0.01), (3000, 9000, 6000, 70, 0.01), (70, 9000, 128, 50, 0.2),
                                  (5000, 6000, 1000, 20, 0.02), (65, 65, 0, 10, 0.3),
                                  # the lane-per-sample kernel (n <= 512, r <= 256): 1, 2, 3 and 4 syndrome words; 2, 4, 8 error words
                                  (130, 200, None, 100, 0.1), (192, 512, 320, 333, 0.02), (256, 512, 0, 1500, 0.01),
                                  (10, 100, None, 70, 0.3), (255, 511, 256, 2000, 0.01), (129, 257, 1, 600, 0.05)])
def test_syndrome_sparse_kernel(case, ctx):
    r, n, ioff, batch, density = case
    rng = np.random.default_rng(r * 3 + n)
    hm = rng.integers(0, 2, (r, n))
    if ioff is not None:
        hm[:, ioff:ioff + r] = np.identity(r, dtype=int)
    em = (rng.random((batch, n)) < density).astype(np.uint8)
    em[0] = 0                                               # an error-free sample
    if batch > 3:
        em[3] = 1                                           # and a full-weight one (exceeds the list capacity)
    h, e = _native.pack_rows(hm), _native.pack_rows(em)
    chk = ctx.check_create(h, r, n)
    lde, lds = e.shape[1], _native.words_for(r)
    e_buf = ctx.alloc(e.nbytes).upload(e)
    s_buf = ctx.alloc(batch * lds * 8).zero()
    hist = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, s_buf, lds, hist, r + 1)
    want = c_oracle.syndrome_batch(h, r, n, e, batch)
    assert np.array_equal(s_buf.download((batch, lds), "<u8"), want)
    assert np.array_equal(hist.download((r + 1,), np.uint64), c_oracle.histogram(want, batch, r, 1, r + 1))
    # histogram-only and syndromes-only variants
    hist2 = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist2, r + 1)
    assert np.array_equal(hist2.download((r + 1,), np.uint64), c_oracle.histogram(want, batch, r, 1, r + 1))
    s2 = ctx.alloc(batch * lds * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, s2, lds)
    assert np.array_equal(s2.download((batch, lds), "<u8"), want)
    # the dense table kernel gives the same answer
    assert np.array_equal(ctx.syndrome_batch(h, r, n, e, batch), want)


def test_monte_carlo_n4096_dense_and_sparse_pipelines_agree(ctx, route):
    rng = np.random.default_rng(11)
    hm1, hm2 = rng.integers(0, 2, (2048, 4096)), rng.integers(0, 2, (2047, 4096))
    hm1[:, :2048] = np.identity(2048, dtype=int)
    hm2[:, 2048:4095] = np.identity(2047, dtype=int)
    h1, h2 = _native.pack_rows(hm1), _native.pack_rows(hm2)
    c1, c2 = ctx.check_create(h1, 2048, 4096), ctx.check_create(h2, 2047, 4096)
    args = (0xABC, 5 * 10**6, 20000, 0.004, 0.003, 0.002, _native.HIST_WEIGHT)
    fused = ctx.mc_run(c1, c2, *args)                         # sampler fused into the sparse kernel
    route.force("GF2_MC_UNFUSED")
    sparse = ctx.mc_run(c1, c2, *args)                        # sampler kernel -> sparse kernel
    route.release("GF2_MC_UNFUSED")
    route.force("GF2_MC_DENSE")
    dense = ctx.mc_run(c1, c2, *args)                         # sampler kernel -> table kernel -> histogram kernel
    route.release("GF2_MC_DENSE")
    want = c_oracle.mc(h1, 2048, h2, 2047, 4096, 0xABC, 5 * 10**6, 20000, 0.004, 0.003, 0.002, 1)
    for got in (fused, sparse, dense):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # large counts: the record sampler of chunk k + 1 (records and identity words, no packed rows) overlaps the two
    # components' gather / combine / misfit kernels of chunk k on three streams (a chunk and a bit, and with chunks of 2^19
    # four and a bit, so the two buffer sets are reused); same histograms as the fused kernel, as the packed-row sampler
    # with the compact kernels on three streams (GF2_MC_ROWS) and, on its own pace, the serial sampler -> pipeline -> pipeline
    # path; twice, to see that nothing is left behind on a stream
    big = (0xABC, 123, (1 << 21) + 77777, 0.004, 0.003, 0.002, _native.HIST_WEIGHT)
    piped = ctx.mc_run(c1, c2, *big)
    piped_again = ctx.mc_run(c1, c2, *big)
    ctx.set_option(_native.OPT_MC_CHUNK_LOG2, 19)
    piped_small_chunks = ctx.mc_run(c1, c2, *big)
    ctx.set_option(_native.OPT_MC_CHUNK_LOG2, None)
    assert np.array_equal(piped_small_chunks[0], piped[0]) and np.array_equal(piped_small_chunks[1], piped[1])
    route.force("GF2_MC_ROWS")
    rows_big = ctx.mc_run(c1, c2, *big)
    route.release("GF2_MC_ROWS")
    assert np.array_equal(rows_big[0], piped[0]) and np.array_equal(rows_big[1], piped[1])
    route.force("GF2_MC_FUSED")
    fused_big = ctx.mc_run(c1, c2, *big)
    route.release("GF2_MC_FUSED")
    route.force("GF2_MC_UNFUSED")
    serial_big = ctx.mc_run(c1, c2, *big)
    route.release("GF2_MC_UNFUSED")
    assert int(piped[0].sum()) == big[2] and int(piped[1].sum()) == big[2]
    for got in (piped_again, fused_big, serial_big):
        assert np.array_equal(got[0], piped[0]) and np.array_equal(got[1], piped[1])
    # ... and the default route's 2.2 million samples against the oracle directly, not only against the other routes
    want_big = c_oracle.mc(h1, 2048, h2, 2047, 4096, *big[:-1], 1)
    assert np.array_equal(piped[0], want_big[0]) and np.array_equal(piped[1], want_big[1])
    for cap in (0, 2, 6):                                     # (the leftovers of the record sampler: see the next test)
        ctx.set_option(_native.OPT_MC_TAIL_CAP, cap)
        try:
            capped = ctx.mc_run(c1, c2, *big)
        finally:
            ctx.set_option(_native.OPT_MC_TAIL_CAP, None)
        assert np.array_equal(capped[0], want_big[0]) and np.array_equal(capped[1], want_big[1]), cap
    # a rate at which most tiles have samples that do not fit their records (17 columns per sample on average): the misfit
    # kernel draws those again; and a ragged last tile
    hot = (0x5EED, 99, 70001, 0.005, 0.0035, 0.002, _native.HIST_WEIGHT)
    hot_records = ctx.mc_run(c1, c2, *hot)
    route.force("GF2_MC_FUSED")
    hot_fused = ctx.mc_run(c1, c2, *hot)
    route.release("GF2_MC_FUSED")
    assert int(hot_records[0].sum()) == hot[2] and int(hot_records[1].sum()) == hot[2]
    assert np.array_equal(hot_records[0], hot_fused[0]) and np.array_equal(hot_records[1], hot_fused[1])
    want_hot = c_oracle.mc(h1, 2048, h2, 2047, 4096, *hot[:-1], 1)
    assert np.array_equal(hot_records[0], want_hot[0]) and np.array_equal(hot_records[1], want_hot[1])


@pytest.mark.parametrize("case", [(1023, 511, 511, 512, 0.02), (2047, 1023, 1023, 1024, 0.012), (3000, 1500, 1400, 1536, 0.008),
                                  (4096, 1536, 2048, 2048, 0.007), (1100, 512, 500, 512, 0.02)])
def test_monte_carlo_record_sampler_other_shapes(case, ctx, route):
    # the record sampler away from the benchmark's shape: a last segment of fewer than 512 qubits, identity blocks of other sizes
    # and positions (H1 = [I | A], H2 = [A' | I | ...] with its block starting at a multiple of 512), a second block that ends
    # inside a segment; against the fused kernel, the packed-row route and, on a prefix, the oracle
    n, r1, r2, off2, p_total = case
    rng = np.random.default_rng(n + r1)
    hm1, hm2 = rng.integers(0, 2, (r1, n)), rng.integers(0, 2, (r2, n))
    hm1[:, :r1] = np.identity(r1, dtype=int)
    hm2[:, off2:off2 + r2] = np.identity(r2, dtype=int)
    h1, h2 = _native.pack_rows(hm1), _native.pack_rows(hm2)
    c1, c2 = ctx.check_create(h1, r1, n), ctx.check_create(h2, r2, n)
    args = (31, 500, 70001 + n, p_total / 2, p_total / 4, p_total / 4, _native.HIST_WEIGHT)
    records = ctx.mc_run(c1, c2, *args)
    route.force("GF2_MC_FUSED")
    fused = ctx.mc_run(c1, c2, *args)
    route.release("GF2_MC_FUSED")
    route.force("GF2_MC_ROWS")
    rows = ctx.mc_run(c1, c2, *args)
    route.release("GF2_MC_ROWS")
    assert int(records[0].sum()) == args[2] and int(records[1].sum()) == args[2]
    for got in (fused, rows):
        assert np.array_equal(got[0], records[0]) and np.array_equal(got[1], records[1])
    want_all = c_oracle.mc(h1, r1, h2, r2, n, *args[:-1], 1)           # the default route's whole call against the oracle
    assert np.array_equal(records[0], want_all[0]) and np.array_equal(records[1], want_all[1])
    # the sampler's lanes stop after `cap` erroneous qubits of a segment and leave the rest of such a sample to a lane of its own
    # (default: by the rate; 0 = every lane to the end, as in round 3).  At cap 2 and these rates nearly every (sample, segment)
    # pair has leftovers, so the list of 64 pairs fills and is worked off in the middle of a tile, segment after segment.
    for cap in (0, 2, 4, 8):
        ctx.set_option(_native.OPT_MC_TAIL_CAP, cap)
        try:
            capped = ctx.mc_run(c1, c2, *args)
        finally:
            ctx.set_option(_native.OPT_MC_TAIL_CAP, None)
        assert np.array_equal(capped[0], want_all[0]) and np.array_equal(capped[1], want_all[1]), cap
    want = c_oracle.mc(h1, r1, h2, r2, n, 31, 500, 3000, p_total / 2, p_total / 4, p_total / 4, 1)
    small = ctx.mc_run(c1, c2, 31, 500, 3000, p_total / 2, p_total / 4, p_total / 4, _native.HIST_WEIGHT)
    assert np.array_equal(small[0], want[0]) and np.array_equal(small[1], want[1])


@pytest.mark.parametrize("case", [(127, 63, 64), (255, 127, 127), (511, 255, 250), (300, 100, 150)])
def test_monte_carlo_mid_size_checks_take_the_lane_kernel(case, ctx, route):
    # n <= 512, r <= 256: gf2_mc_run draws dense rows and runs the lane-per-sample kernel per component; same histograms
    # from the oracle, from the fused kernel and from the slab pipelines
    n, r1, r2 = case
    rng = np.random.default_rng(n)
    hm1, hm2 = rng.integers(0, 2, (r1, n)), rng.integers(0, 2, (r2, n))
    hm1[:, :r1] = np.identity(r1, dtype=int)
    hm2[:, n - r2:] = np.identity(r2, dtype=int)
    h1, h2 = _native.pack_rows(hm1), _native.pack_rows(hm2)
    c1, c2 = ctx.check_create(h1, r1, n), ctx.check_create(h2, r2, n)
    args = (77, 1000, 70001, 0.004, 0.003, 0.005, _native.HIST_WEIGHT)
    got = ctx.mc_run(c1, c2, *args)
    want = c_oracle.mc(h1, r1, h2, r2, n, 77, 1000, 70001, 0.004, 0.003, 0.005, 1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    route.force("GF2_MC_FUSED")
    fused = ctx.mc_run(c1, c2, *args)
    route.release("GF2_MC_FUSED")
    assert np.array_equal(fused[0], want[0]) and np.array_equal(fused[1], want[1])


# ---- table decode + logical-error tally -------------------------------------------------------------

def test_logical_error_rates_vs_oracle(steane_h, rm15):
    from quantum_css_codes_amd.montecarlo import dense_table, packed_word
    for code, p, count in ((CSSCode(steane_h, steane_h), (0.03, 0.01, 0.02), 400000),
                           (CSSCode(*rm15), (0.05, 0.02, 0.03), 300000)):
        got = code.logical_error_rates(count, *p, seed=77, first_sample=10**9)
        want = c_oracle.mc_decode(c_oracle.pack_rows(code.parity_check_c1), code.r_1,
                                  c_oracle.pack_rows(code.parity_check_c2), code.r_2, code.n,
                                  dense_table(code._c1_syndromes, code.r_1, code.n),
                                  dense_table(code._c2_syndromes, code.r_2, code.n),
                                  packed_word(code.x_operator_matrix()[0]), packed_word(code.z_operator_matrix()[0]),
                                  77, 10**9, count, *p)
        assert [got[f] for f in ('logical_x', 'logical_z', 'logical_any', 'uncorrectable_x', 'uncorrectable_z')] == \
            [int(v) for v in want]
        assert got['samples'] == count and 0 < got['logical_any'] < count
    # against the statement built on the reference's dict tables, small sample
    code = CSSCode(steane_h, steane_h)
    small = code.logical_error_rates(300, 0.06, 0.03, 0.05, seed=21, first_sample=1000)
    want = cpu_ref.decode_and_tally(cpu_ref.CSSCode(steane_h, steane_h), 21, 1000, 300, 0.06, 0.03, 0.05)
    assert [small[f] for f in ('logical_x', 'logical_z', 'logical_any', 'uncorrectable_x', 'uncorrectable_z')] == want
    # Steane corrects every single-qubit error: at weight <= 1 nothing flips.  p tiny -> no flips in 10^5 samples
    quiet = code.logical_error_rates(100000, 1e-9, 1e-9, 1e-9, seed=3)
    assert quiet['logical_any'] == 0 and quiet['uncorrectable_x'] == 0


def test_small_code_fused_and_pipeline_agree(rm15, steane_h, route):
    # the fused small-code kernel and the sampler -> syndrome -> histogram pipeline give identical histograms
    for code in (CSSCode(steane_h, steane_h), CSSCode(*rm15)):
        for mode in ('full', 'weight'):
            fused = code.monte_carlo(200000, 0.04, 0.01, 0.02, seed=9, first_sample=77, mode=mode)
            route.force("GF2_MC_PIPELINE")
            piped = code.monte_carlo(200000, 0.04, 0.01, 0.02, seed=9, first_sample=77, mode=mode)
            route.release("GF2_MC_PIPELINE")
            assert np.array_equal(fused['hist_z'], piped['hist_z']) and np.array_equal(fused['hist_x'], piped['hist_x'])
            assert int(fused['hist_z'].sum()) == 200000


@TALL_ROUTES
def test_rref_more_than_8192_rows_streaming_panel_kernel(ctx, sweep_k):
    # m > 8192 takes the streamed panel kernels; dependent rows and pivot-free column stripes force extra rounds
    m, n = 8300, 9100
    rng = np.random.default_rng(83)
    a = rng.integers(0, 2, (m, n), dtype=np.uint8)
    a[5000] = a[1] ^ a[2]
    a[8299] = a[8298]
    a[:, 100:164] = 0
    a[:, 3000] = 0
    a[:, 9000:9090] = a[:, 0:90]
    packed = _native.pack_rows(a)
    want, want_piv, want_rank = c_oracle.rref(packed, m, n)
    pivots, rank = ctx.rref(packed, m, n)
    assert rank == want_rank and list(pivots) == list(want_piv)
    assert np.array_equal(packed, want)


@TALL_ROUTES
@pytest.mark.parametrize("case", ["dependent rows, odd number of panels", "full rank, done half way"])
def test_rref_streamed_panels_with_look_ahead(case, ctx, route, sweep_k):
    # more than 8192 rows and several column chunks: the next sweep's (pair's) panels run on a side stream under the trailing
    # pass of this one (launch_rref_sweeps_streamed / launch_rref_blocked); same matrix, pivots and rank as the oracle and as
    # the run without look-ahead
    rng = np.random.default_rng(8400)
    if case.startswith("dependent"):
        m, n = 8300, 16400                       # 257 panels: the last pair has one; the rank never reaches m, every panel runs
        a = rng.integers(0, 2, (m, n), dtype=np.uint8)
        a[5000] = a[1] ^ a[2]
        a[8299] = a[8298]
        a[:, 100:164] = 0
        a[:, 2048:2112] = 0                      # a whole panel without a pivot, at a chunk boundary
        a[:, 3000] = 0
        a[:, 9000:9090] = a[:, 0:90]
    else:
        m, n = 8200, 16500                       # random: full rank after 129 panels, the rest are never launched
        a = rng.integers(0, 2, (m, n), dtype=np.uint8)
    packed = _native.pack_rows(a)
    want, want_piv, want_rank = c_oracle.rref(packed.copy(), m, n)
    ahead = packed.copy()
    route.force("GF2_RREF_LOOKAHEAD")            # (by default only from 128 MiB of matrix on, where it pays)
    pivots, rank = ctx.rref(ahead, m, n)
    route.release("GF2_RREF_LOOKAHEAD")
    assert rank == want_rank and list(pivots) == list(want_piv)
    assert np.array_equal(ahead, want)
    route.force("GF2_RREF_NO_LOOKAHEAD")
    plain = packed.copy()
    pivots2, rank2 = ctx.rref(plain, m, n)
    route.release("GF2_RREF_NO_LOOKAHEAD")
    assert rank2 == want_rank and list(pivots2) == list(want_piv) and np.array_equal(plain, want)


@TALL_ROUTES
def test_rref_look_ahead_on_a_batch_of_two_tall_matrices(ctx, route, sweep_k):
    # the look-ahead's state copies, coefficient sets and snapshots are per matrix: two different 8300-row matrices in one call
    m, n, batch = 8300, 8200, 2
    rng = np.random.default_rng(2)
    mats = rng.integers(0, 2, (batch, m, n), dtype=np.uint8)
    mats[1, :, 700:764] = 0                      # the second one has a panel without pivots where the first has 64
    mats[1, 17] = mats[1, 3]
    packed = np.stack([_native.pack_rows(mats[b]) for b in range(batch)])
    want = [c_oracle.rref(packed[b].copy(), m, n) for b in range(batch)]
    route.force("GF2_RREF_LOOKAHEAD")
    got = packed.copy()
    pivots, ranks = ctx.rref_batch(got, batch, m, n)
    route.release("GF2_RREF_LOOKAHEAD")
    for b in range(batch):
        assert int(ranks[b]) == want[b][2]
        assert list(pivots[b][:ranks[b]]) == list(want[b][1])
        assert np.array_equal(got[b], want[b][0])


@TALL_ROUTES
def test_rref_256_mib_matrix_with_and_without_look_ahead(ctx, route, sweep_k):
    # bench.py's 32768 x 65536 matrix: too large for the CPU oracle, so the size-independent properties -- the run with look-ahead
    # (the default at this size) and the run without it return the same bytes, the rank is full, the pivot columns ascend, the
    # pivot columns of the result form an identity, and rows of the input lie in the row space of the result (spot-checked
    # through the syndrome kernel: R's nullspace annihilates the input rows)
    m, n = 32768, 65536
    ld = n // 64
    rng = np.random.default_rng(4096)
    a = (rng.integers(0, 2**63, (m, ld), dtype=np.int64).view(np.uint64) << np.uint64(1)) | \
        rng.integers(0, 2, (m, ld), dtype=np.int64).view(np.uint64)
    outs = []
    for flag in (None, "GF2_RREF_NO_LOOKAHEAD"):
        if flag:
            route.force(flag)
        buf = ctx.alloc(a.nbytes).upload(a)
        piv, rk = ctx.alloc(m * 8).zero(), ctx.alloc(8)
        _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, 1, m, n, ld, piv.ptr, rk.ptr))
        outs.append((buf.download((m, ld), "<u8"), piv.download((m,), np.int64), int(rk.download((1,), np.int64)[0])))
        for b in (buf, piv, rk):
            b.free()
        if flag:
            route.release(flag)
    (red, piv, rank), (red2, piv2, rank2) = outs
    assert rank == rank2 == m and np.array_equal(piv, piv2) and np.array_equal(red, red2)
    assert np.all(np.diff(piv) > 0)
    # column piv[i] of the result is e_i (checked on a spread of pivots: one word column each)
    for i in range(0, m, 997):
        col = (red[:, piv[i] >> 6] >> np.uint64(piv[i] & 63)) & np.uint64(1)
        assert int(col.sum()) == 1 and int(col[i]) == 1
    # a row of the input is the XOR of the result's rows selected by its bits in the pivot columns
    for i in (0, 12345, m - 1):
        bits = (a[i][piv >> 6] >> (piv & 63).astype(np.uint64)) & np.uint64(1)
        combo = np.bitwise_xor.reduce(red[bits.astype(bool)], axis=0)
        assert np.array_equal(combo, a[i])


def test_host_syndrome_batch_routes_sparse_and_dense(ctx):
    # gf2_syndrome_batch picks the column kernel for sparse host errors and the table kernel otherwise: same answers
    r, n, batch = 300, 2000, 5000
    rng = np.random.default_rng(300)
    h = _native.pack_rows(rng.integers(0, 2, (r, n)))
    for density in (0.002, 0.05, 0.5):
        em = (rng.random((batch, n)) < density).astype(np.uint8)
        em[::7] = 0
        e = _native.pack_rows(em)
        assert np.array_equal(ctx.syndrome_batch(h, r, n, e, batch), c_oracle.syndrome_batch(h, r, n, e, batch))


def test_syndrome_table_mid_size_codes_vs_oracle():
    # tables of mid-size codes (the reference's Python loop takes minutes here)
    rng = np.random.default_rng(23)
    # a [23, 12] random code: r = 11
    h = rng.integers(0, 2, (11, 23))
    t, table = css_code.syndrome_table(h)
    want_t, want = cpu_ref.syndrome_table(h)
    assert t == want_t and list(table.keys()) == [int(k) for k in want.keys()]
    for k in table:
        assert np.array_equal(table[k], want[k])
    # r > 63: keys are exact Python ints; capped search
    h = rng.integers(0, 2, (70, 80))
    t, table = css_code.syndrome_table(h, max_weight=1)
    want_t, want = cpu_ref.syndrome_table(np.array(h, dtype=object), max_weight=1)
    assert t == want_t and list(table.keys()) == [int(k) for k in want.keys()] and len(table) == 81
    # n = 60, r = 30, weight <= 4: 523k errors (the hash-table search since round 4; rounds 1-3 enumerated in Python and were
    # checked by count): t, every key in the reference's insertion order and every error against the C oracle's enumeration
    h = rng.integers(0, 2, (30, 60))
    t, table = css_code.syndrome_table(h, max_weight=4)
    want_t, want_keys, want_errs = c_oracle.syndrome_table(c_oracle.pack_rows(h), 30, 60, 4)
    assert t == want_t and list(table.keys()) == want_keys
    assert np.array_equal(np.array(list(table.values())), c_oracle.unpack_rows(want_errs, 60))
    from math import comb
    assert len(table) == sum(comb(60, w) for w in range(t + 1))
    # more than 127 checks: the host enumerates, the device computes the syndromes, keys are exact Python ints
    h = rng.integers(0, 2, (130, 140))
    t, table = css_code.syndrome_table(h, max_weight=1)
    want_t, want = cpu_ref.syndrome_table(np.array(h, dtype=object), max_weight=1)
    assert t == want_t and list(table.keys()) == [int(k) for k in want.keys()] and len(table) == 141


@pytest.mark.parametrize("case", [(65, 10, None), (96, 12, None), (127, 14, None), (127, 20, 2), (128, 24, 2), (100, 0, 1), (70, 7, None)])
def test_syndrome_table_two_word_codes_on_the_device(case):
    # SURVEY.md 8f item 2 names n = 23..127: 64 < n <= 128 is searched on the device with two-word errors (gf2_syndrome_table_wide);
    # same threshold, same keys in the same insertion order, same error vectors as the oracle's restatement of css_code.py:715-735
    n, r, cap = case
    rng = np.random.default_rng(n * 31 + r)
    h = rng.integers(0, 2, (r, n))
    if r >= 14 and cap is None:
        # a BCH-like check: columns are distinct and pairwise sums distinct for most draws; whatever t comes out must agree
        h[:, :r] = np.identity(r, dtype=int)
    t, table = css_code.syndrome_table(h, max_weight=cap)
    want_t, want = cpu_ref.syndrome_table(h, max_weight=cap)
    assert t == want_t
    assert list(table.keys()) == [int(k) for k in want.keys()]
    for k in list(table.keys())[::max(1, len(table) // 500)]:
        assert np.array_equal(table[k], want[k])
    for key, err in list(table.items())[::97]:
        assert key == bin_matrix.vec_to_int(np.mod(h @ err, 2))


def bch_check_matrix(m_bits, poly, powers):
    """Rows of [alpha^(p j)] for p in `powers` over GF(2^m_bits), j = 0 .. 2^m_bits - 2, every field element as m_bits binary rows."""
    n = (1 << m_bits) - 1
    alpha = [1]
    for _ in range(n - 1):
        v = alpha[-1] << 1
        alpha.append(v ^ poly if v >> m_bits else v)
    rows = []
    for p in powers:
        elems = [alpha[(p * j) % n] for j in range(n)]
        rows += [[(e >> b) & 1 for e in elems] for b in range(m_bits)]
    return np.array(rows)


@pytest.mark.parametrize("case", ["bch255", "bch511cut", (200, 24, None), (300, 20, 2), (1000, 24, 1), (129, 10, None),
                                  (4096, 24, 1), (130, 0, 1), (8192, 13, None)])
def test_syndrome_table_beyond_128_bits_on_the_device(case):
    # n > 128: errors enumerated as position lists on the device (gf2_syndrome_table_cols), keys = XOR of column
    # keys; same threshold, same keys in the same insertion order, same error vectors as the oracle's restatement of css_code.py:715-735
    cap = None
    if case == "bch255":                         # double-error-correcting BCH code: the classes 0, 1, 2 are distinct, t = 2
        h = bch_check_matrix(8, 0x11D, (1, 3))
    elif case == "bch511cut":                    # the same at 9 bits with columns removed (n = 400)
        h = bch_check_matrix(9, 0x211, (1, 3))[:, 50:450]
    else:
        n, r, cap = case
        h = np.random.default_rng(n * 7 + r).integers(0, 2, (r, n))
    t, table = css_code.syndrome_table(h, max_weight=cap)
    want_t, want = cpu_ref.syndrome_table(h, max_weight=cap)
    assert t == want_t
    if isinstance(case, str):
        assert t == 2
    assert list(table.keys()) == [int(k) for k in want.keys()]
    for k in list(table.keys())[::max(1, len(table) // 500)]:
        assert np.array_equal(table[k], want[k])
    for key, err in list(table.items())[::97]:
        assert key == bin_matrix.vec_to_int(np.mod(h @ err, 2))


def test_abi_argument_errors(ctx):
    # error codes and messages of the C ABI (include/gf2hip.h): nothing is computed, nothing crashes
    import ctypes
    lib = _native.lib()
    h = _native.pack_rows(np.eye(3, 200, dtype=np.uint8))
    chk = ctx.check_create(h, 3, 200)
    small = ctx.check_create(_native.pack_rows(np.eye(3, 7, dtype=np.uint8)), 3, 7)
    buf = ctx.alloc(1 << 16).zero()

    def rc_of(fn, *args):
        code = fn(*args)
        return code, lib.gf2_last_error().decode()

    code, msg = rc_of(lib.gf2_syndrome_dev, ctx.handle, chk.handle, buf.ptr, 10, 1, 0, buf.ptr, 1)
    assert code == _native.GF2_E_ARG and "lde too small" in msg
    code, msg = rc_of(lib.gf2_syndrome_dev, ctx.handle, chk.handle, buf.ptr, 10, 4, 7, buf.ptr, 1)
    assert code == _native.GF2_E_ARG and "unknown layout" in msg
    code, msg = rc_of(lib.gf2_syndrome_dev, ctx.handle, chk.handle, buf.ptr, 10, 4, _native.LAYOUT_BIT_SLICED, buf.ptr, 1)
    assert code == _native.GF2_E_ARG and "bit-sliced" in msg
    code, msg = rc_of(lib.gf2_syndrome_dev, ctx.handle, small.handle, buf.ptr, 10, 1, _native.LAYOUT_TILED, buf.ptr, 16)
    assert code == _native.GF2_E_ARG and "tiled" in msg
    code, msg = rc_of(lib.gf2_syndrome_sparse_dev, ctx.handle, small.handle, buf.ptr, 10, 1, buf.ptr, 1, None, 0)
    assert code == _native.GF2_E_ARG and "no transposed columns" in msg
    code, msg = rc_of(lib.gf2_syndrome_sparse_dev, ctx.handle, chk.handle, buf.ptr, 10, 4, None, 0, buf.ptr, 9)
    assert code == _native.GF2_E_ARG and "r+1 bins" in msg
    code, msg = rc_of(lib.gf2_histogram_dev, ctx.handle, buf.ptr, 10, 1, 0, 30, _native.HIST_FULL, buf.ptr, 8)
    assert code == _native.GF2_E_ARG and "r <= 24" in msg
    code, msg = rc_of(lib.gf2_histogram_dev, ctx.handle, buf.ptr, 10, 1, 0, 3, _native.HIST_WEIGHT, buf.ptr, 8)
    assert code == _native.GF2_E_ARG and "r+1 bins" in msg
    code, msg = rc_of(lib.gf2_sample_errors_dev, ctx.handle, 200, 1, 0, 10, 0.6, 0.3, 0.3, buf.ptr, buf.ptr, 4, 0)
    assert code == _native.GF2_E_ARG and "sum to at most 1" in msg
    code, msg = rc_of(lib.gf2_sample_errors_dev, ctx.handle, 200, 1, 0, 10, -0.1, 0.0, 0.0, buf.ptr, buf.ptr, 4, 0)
    assert code == _native.GF2_E_ARG
    rank = ctypes.c_int64()
    code, msg = rc_of(lib.gf2_rref, ctx.handle, None, 3, 200, 1, None, ctypes.byref(rank))
    assert code == _native.GF2_E_ARG and "bad shape" in msg
    nsw = ctypes.c_int64()
    code, msg = rc_of(lib.gf2_normalize, ctx.handle, h.ctypes.data, 3, 200, 4, 199, None, ctypes.byref(nsw))
    assert code == _native.GF2_E_COLUMNS and msg == "not enough columns"
    with pytest.raises(ValueError, match="not enough columns"):
        css_code.normalize_parity_check(np.eye(3, 5, dtype=int), 3)
    with pytest.raises(_native.GF2Error):
        ctx.mc_run(chk, small, 1, 0, 10, 0.1, 0.0, 0.0, _native.HIST_WEIGHT)     # different n
    # zero-sized work is a no-op everywhere
    assert ctx.syndrome_batch(h, 3, 200, np.zeros((0, 4), dtype="<u8"), 0).shape == (0, 1)
    assert bin_matrix.reduced_row_echelon_form(np.zeros((0, 9), dtype=int)).shape == (0, 9)
    assert css_code.syndrome_batch(np.zeros((0, 9), dtype=int), np.zeros((5, 9), dtype=int)).shape == (5, 0)
    got = ctx.mc_run(chk, chk, 1, 0, 0, 0.1, 0.0, 0.0, _native.HIST_WEIGHT)
    assert int(got[0].sum()) == 0


@pytest.mark.parametrize("case", [(100, 300, 0, 100), (300, 1000, 300, 600), (2048, 4096, 0, 2048), (130, 4000, None, None)])
def test_fused_sparse_monte_carlo_shapes(case, ctx):
    # fused sampler + sparse kernel on shapes with and without identity blocks, n not a multiple of 64
    r1, n, off1, off2 = case
    rng = np.random.default_rng(r1 + n)
    r2 = r1 - 1
    hm1, hm2 = rng.integers(0, 2, (r1, n)), rng.integers(0, 2, (r2, n))
    if off1 is not None:
        hm1[:, off1:off1 + r1] = np.identity(r1, dtype=int)
        hm2[:, off2:off2 + r2] = np.identity(r2, dtype=int)
    h1, h2 = _native.pack_rows(hm1), _native.pack_rows(hm2)
    c1, c2 = ctx.check_create(h1, r1, n), ctx.check_create(h2, r2, n)
    for p in ((0.004, 0.003, 0.002), (0.02, 0.0, 0.01), (0.0, 0.0, 0.0)):
        got = ctx.mc_run(c1, c2, 77, 123456, 3000, *p, _native.HIST_WEIGHT)
        want = c_oracle.mc(h1, r1, h2, r2, n, 77, 123456, 3000, *p, 1)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (case, p)


@pytest.mark.parametrize("case", [(300, 700, 0), (200, 900, 130)])
def test_normalize_with_a_column_swap_at_every_step(case, ctx):
    # the identity-target columns are all zero, so every step swaps a column in: the blocked panels stall each time and
    # the kernel hands over to the sequential path; swaps and result must still be the reference's
    r, n, off = case
    rng = np.random.default_rng(r)
    h = rng.integers(0, 2, (r, n))
    h[:, off:off + r] = 0
    h[:, off + r:off + 2 * r] = np.identity(r, dtype=int)[rng.permutation(r)]
    packed = _native.pack_rows(h)
    rc, want, want_swaps = c_oracle.normalize(packed, r, n, off)
    assert rc == 0 and len(want_swaps) == r
    swaps = ctx.normalize(packed, r, n, off)
    assert swaps == want_swaps
    assert np.array_equal(packed, want)


@pytest.mark.parametrize("case", [(2047, 4096, 2049, 1500, 0.007), (2048, 4096, 0, 5000, 0.01), (2048, 4096, 2048, 700, 0.012),
                                  (1000, 3000, 37, 1100, 0.004), (300, 2500, None, 600, 0.005), (2048, 4000, 1952, 513, 0.007),
                                  (130, 777, 500, 2049, 0.03), (2048, 4096, 0, 64, 0.0), (1300, 3100, 1700, 900, 0.006),
                                  (200, 2000, None, 1000, 0.004), (1536, 3900, 2047, 300, 0.005), (512, 600, 88, 200, 0.02),
                                  (200, 200, 0, 300, 0.05), (2048, 4096, 1024, 400, 0.006),
                                  # the hand-scheduled gather kernel (identity words = 16-byte pieces) away from the benchmark's
                                  # shapes: fewer rows than a slab holds, a bit offset inside the dword, a slab cut by r
                                  (1000, 3000, 1024, 1300, 0.005), (700, 2600, 1285, 2200, 0.006), (300, 1536, 1152, 900, 0.01),
                                  (1500, 4096, 2560, 777, 0.006), (513, 2048, 1408, 1000, 0.008), (2040, 4096, 2051, 650, 0.007),
                                  # records of the lower half of a sorted tile are 32 bytes (15 columns): at these rates most tiles
                                  # have more than 32 samples beyond that, and the misfits are finished by the compact kernel
                                  (2048, 4096, 0, 3000, 0.0085), (2047, 4096, 2048, 2500, 0.008), (1024, 2304, 1024, 1500, 0.014)])
def test_syndrome_slab_pipeline(case, ctx, route):
    # histogram-only calls take the LDS row-slab pipeline (compact -> gather -> combine) when the check qualifies; it must
    # agree with the oracle and with the column-gather kernel on sparse samples, on samples beyond the record capacity
    # and on ragged batches
    r, n, ioff, batch, density = case
    rng = np.random.default_rng(r * 5 + n + batch)
    hm = rng.integers(0, 2, (r, n))
    if ioff is not None:
        hm[:, ioff:ioff + r] = np.identity(r, dtype=int)
    em = (rng.random((batch, n)) < density).astype(np.uint8)
    heavy = rng.choice(batch, size=min(batch, 9), replace=False)
    em[heavy] = (rng.random((len(heavy), n)) < 0.03).astype(np.uint8)     # beyond 32 listed columns
    em[heavy[0]] = 1
    h, e = _native.pack_rows(hm), _native.pack_rows(em)
    chk = ctx.check_create(h, r, n)
    lde = e.shape[1]
    e_buf = ctx.alloc(e.nbytes).upload(e)
    want = c_oracle.histogram(c_oracle.syndrome_batch(h, r, n, e, batch), batch, r, 1, r + 1)
    route.force("GF2_SPARSE_SLABS")              # small batches default to the column-gather kernel
    hist = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist, r + 1)
    assert np.array_equal(hist.download((r + 1,), np.uint64), want)
    # the syndromes themselves out of the same pipeline (round 4: every gather workgroup stores its slab's 64-byte piece; samples
    # beyond a record are stored by the compact kernel), with and without the histogram, at the tightest pitch and a wider one
    want_s = c_oracle.syndrome_batch(h, r, n, e, batch)
    for lds in (want_s.shape[1], want_s.shape[1] + 3):
        for with_hist in (True, False):
            s_buf = ctx.alloc(batch * lds * 8).upload(np.full((batch, lds), 0xA5A5A5A5A5A5A5A5, dtype=np.uint64))
            hist_s = ctx.alloc((r + 1) * 8).zero()
            ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, s_buf, lds, hist_s if with_hist else None, r + 1 if with_hist else 0)
            got_s = s_buf.download((batch, lds), np.uint64)
            assert np.array_equal(got_s[:, :want_s.shape[1]], want_s), (lds, with_hist)
            pad = got_s[:, want_s.shape[1]:]                # (pad words of a row: left alone by the slab pipeline, zeroed by the column-gather kernel)
            assert ((pad == 0xA5A5A5A5A5A5A5A5) | (pad == 0)).all(), "something was written past the syndromes"
            if with_hist:
                assert np.array_equal(hist_s.download((r + 1,), np.uint64), want)
            s_buf.free(), hist_s.free()
    # ... and with a gather step's 16 records taken from four sorted tiles instead of a quartile of one (GF2_OPT_GATHER_CROSS)
    ctx.set_option(_native.OPT_GATHER_CROSS, 1)
    try:
        ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist, r + 1)
    finally:
        ctx.set_option(_native.OPT_GATHER_CROSS, None)
    assert np.array_equal(hist.download((r + 1,), np.uint64), want * np.uint64(2))
    route.release("GF2_SPARSE_SLABS")
    route.force("GF2_SPARSE_GATHER")
    hist2 = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist2, r + 1)
    assert np.array_equal(hist2.download((r + 1,), np.uint64), want)


@pytest.mark.parametrize("case", [(2047, 4096, 2048, 3000, 0.007, None), (2046, 4096, 2048, 1500, 0.01, None),
                                  (2047, 4096, 2049, 1500, 0.007, None), (2047, 4096, 2048, 700, 0.004, 4095),
                                  (1023, 2048, 1024, 2100, 0.01, None), (2045, 4096, 2048, 500, 0.01, None),
                                  # ... and the benchmark's own H2: that column is ZERO (first n - r_1 - 1 rows of [A^T | I]): nothing
                                  # to list and nothing to redo; with a zero and a non-zero column side by side only the latter counts
                                  (2047, 4096, 2048, 3000, 0.007, "zero"), (2046, 4096, 2048, 1500, 0.01, "zero")])
def test_slab_pipeline_columns_left_to_the_redo_pass(case, ctx, route):
    # a check with one or two non-identity columns next to the identity block (H2 of a CSS code with k = 1, 2): compact leaves
    # their word out, the gather kernel flags the samples that have them and the redo kernel computes those from the row;
    # three columns (last case) stay in the records.  Same histogram with the redo pass switched off and from the oracle.
    r, n, ioff, batch, density, all_set = case
    rng = np.random.default_rng(r + n + batch)
    hm = rng.integers(0, 2, (r, n))
    hm[:, ioff:ioff + r] = np.identity(r, dtype=int)
    em = (rng.random((batch, n)) < density).astype(np.uint8)
    if all_set == "zero":
        hm[:, n - 1] = 0
        em[:, n - 1] = rng.integers(0, 2, batch)               # half the samples have the column that changes nothing
        all_set = None
    if all_set is not None:
        em[:, all_set] = 1                                     # every sample goes through the redo pass
    em[5] = (rng.random(n) < 0.03).astype(np.uint8)            # beyond a record's capacity: finished by compact
    h, e = _native.pack_rows(hm), _native.pack_rows(em)
    chk = ctx.check_create(h, r, n)
    lde = e.shape[1]
    e_buf = ctx.alloc(e.nbytes).upload(e)
    want = c_oracle.histogram(c_oracle.syndrome_batch(h, r, n, e, batch), batch, r, 1, r + 1)
    route.force("GF2_SPARSE_SLABS")
    hist = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist, r + 1)
    assert np.array_equal(hist.download((r + 1,), np.uint64), want)
    # stored syndromes: a sample with a left-out column is stored twice, without the column by the gather kernel and then whole
    # by the redo kernel
    want_s = c_oracle.syndrome_batch(h, r, n, e, batch)
    lds = want_s.shape[1]
    s_buf = ctx.alloc(batch * lds * 8).zero()
    hist_s = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, s_buf, lds, hist_s, r + 1)
    assert np.array_equal(s_buf.download((batch, lds), np.uint64), want_s)
    assert np.array_equal(hist_s.download((r + 1,), np.uint64), want)
    route.force("GF2_NO_REDO")
    hist2 = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist2, r + 1)
    assert np.array_equal(hist2.download((r + 1,), np.uint64), want)
    s_buf.zero()
    ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, s_buf, lds, None, 0)
    assert np.array_equal(s_buf.download((batch, lds), np.uint64), want_s)


def test_syndrome_slab_pipeline_default_route_large_batch(ctx, route):
    # above the batch threshold the histogram-only call takes the slab pipeline by itself (64 sample shares per slab,
    # XCD-grouped); same histogram as the column-gather kernel and as the dense table kernel's syndromes
    r, n, ioff, batch = 2047, 4096, 2049, 70001
    rng = np.random.default_rng(11)
    hm = rng.integers(0, 2, (r, n), dtype=np.uint8)
    hm[:, ioff:ioff + r] = np.identity(r, dtype=np.uint8)
    h = _native.pack_rows(hm)
    chk = ctx.check_create(h, r, n)
    ex, ez = ctx.alloc(batch * 512), ctx.alloc(batch * 512)
    ctx.sample_errors_dev(n, 3, 5, batch, 0.004, 0.003, 0.003, ex, ez, 64)
    hists = []
    for force_gather in (False, True):
        if force_gather:
            route.force("GF2_SPARSE_GATHER")
        hist = ctx.alloc((r + 1) * 8).zero()
        ctx.syndrome_sparse_dev(chk, ex, batch, 64, None, 0, hist, r + 1)
        hists.append(hist.download((r + 1,), np.uint64))
    assert np.array_equal(hists[0], hists[1]) and int(hists[0].sum()) == batch
    e = ex.download((batch, 64), "<u8")
    want = c_oracle.histogram(c_oracle.syndrome_batch(h, r, n, e[:3000].copy(), 3000), 3000, r, 1, r + 1)
    first = ctx.alloc((r + 1) * 8).zero()
    route.release("GF2_SPARSE_GATHER")
    route.force("GF2_SPARSE_SLABS")
    ctx.syndrome_sparse_dev(chk, ex, 3000, 64, None, 0, first, r + 1)
    assert np.array_equal(first.download((r + 1,), np.uint64), want)


@pytest.mark.parametrize("shape", [(2047, 4096, 2048), (2048, 4096, 0), (1000, 3000, 1024)])
def test_slab_pipeline_many_passes_and_the_folded_combine_step(shape, ctx, route):
    # a call of many passes (2^12 samples each here; 2^22 by default): the combine step of a pass rides in the next pass' gather
    # kernel (round 4; GF2_F_COMBINE_SEPARATE: a combine kernel after every pass, GF2_F_COMBINE_FOLDED: in the next pass' compact
    # kernel), and the redo list (H2-shaped check: a column next to the identity block) is the call's, worked off once at the end.  Same
    # histogram either way, as the column-gather kernel and, on a prefix, as the oracle; ragged last pass and ragged last tile.
    r, n, ioff = shape
    batch = 9 * 4096 + 1234
    rng = np.random.default_rng(r + n)
    hm = rng.integers(0, 2, (r, n), dtype=np.uint8)
    hm[:, ioff:ioff + r] = np.identity(r, dtype=np.uint8)
    h = _native.pack_rows(hm)
    chk = ctx.check_create(h, r, n)
    lde = _native.words_for(n)
    ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
    ctx.sample_errors_dev(n, 17, 3, batch, 0.004, 0.003, 0.003, ex, ez, lde)
    got = {}
    ctx.set_option(_native.OPT_SLAB_PASS_LOG2, 12)
    try:
        for name in ("default", "GF2_COMBINE_FOLDED", "GF2_COMBINE_SEPARATE", "GF2_NO_REDO"):
            if name != "default":
                route.force(name)
            route.force("GF2_SPARSE_SLABS")
            hist = ctx.alloc((r + 1) * 8).zero()
            ctx.syndrome_sparse_dev(chk, ex, batch, lde, None, 0, hist, r + 1)
            ctx.syndrome_sparse_dev(chk, ex, batch, lde, None, 0, hist, r + 1)          # twice: the redo list starts empty each call
            got[name] = hist.download((r + 1,), np.uint64)
            route.release("GF2_SPARSE_SLABS")
            if name != "default":
                route.release(name)
    finally:
        ctx.set_option(_native.OPT_SLAB_PASS_LOG2, None)
    route.force("GF2_SPARSE_GATHER")
    hist = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, ex, batch, lde, None, 0, hist, r + 1)
    gather = hist.download((r + 1,), np.uint64)
    for name, hist_n in got.items():
        assert np.array_equal(hist_n, gather * np.uint64(2)), name
    assert int(gather.sum()) == batch
    e = ex.download((batch, lde), "<u8")
    want = c_oracle.histogram(c_oracle.syndrome_batch(h, r, n, e[:2000].copy(), 2000), 2000, r, 1, r + 1)
    route.release("GF2_SPARSE_GATHER")
    route.force("GF2_SPARSE_SLABS")
    first = ctx.alloc((r + 1) * 8).zero()
    ctx.syndrome_sparse_dev(chk, ex, 2000, lde, None, 0, first, r + 1)
    assert np.array_equal(first.download((r + 1,), np.uint64), want)


@pytest.mark.parametrize("case", ["steane", "rm15_c2", "identity", "r0", "repetition", "random_31_16", "random_64_24",
                                  "duplicate_columns", "zero_column"])
def test_syndrome_table_device_search(case, ctx, steane_h, rm15):
    # gf2_syndrome_table (n <= 64, r <= 24) against the oracle's restatement of css_code.py:715-735: same threshold,
    # same keys in the same insertion order, same error vectors
    rng = np.random.default_rng(5)
    cap = None
    if case == "steane":
        h = steane_h
    elif case == "rm15_c2":
        h = rm15[1]
    elif case == "identity":
        h = np.identity(9, dtype=int)                        # every error has its own syndrome: t = n
    elif case == "r0":
        h = np.zeros((0, 5), dtype=int)
    elif case == "repetition":
        h = np.array([[1, 1, 0, 0, 0, 0, 0], [0, 1, 1, 0, 0, 0, 0], [0, 0, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 0, 0],
                      [0, 0, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 1, 1]])   # [7,1,7]: t = 3
    elif case == "random_31_16":
        h = rng.integers(0, 2, (15, 31))
    elif case == "random_64_24":
        h = rng.integers(0, 2, (24, 64))
        cap = 2
    elif case == "duplicate_columns":
        h = rng.integers(0, 2, (6, 10))
        h[:, 7] = h[:, 2]                                    # two weight-1 errors collide: t = 0
    else:
        h = rng.integers(0, 2, (6, 10))
        h[:, 4] = 0                                          # a weight-1 error with the zero syndrome: t = 0
    t, table = css_code.syndrome_table(h, max_weight=cap)
    want_t, want = cpu_ref.syndrome_table(np.array(h, dtype=object) if h.shape[0] > 62 else h, max_weight=cap)
    assert t == want_t
    assert list(table.keys()) == [int(k) for k in want.keys()]
    for k in table:
        assert np.array_equal(table[k], np.asarray(want[k], dtype=int))
    # the dense form through the C ABI: filled entries = sum of the accepted classes
    from math import comb
    r, n = h.shape
    t2, dense = ctx.syndrome_table(_native.pack_rows(h), r, n, cap)
    assert t2 == t and int((dense != ctx.TABLE_EMPTY).sum()) == sum(comb(n, w) for w in range(t + 1))


# ---- encoder gate lists and stabiliser conjugation --------------------------------------------------

def test_transform_stabilisers_golden_sequences():
    # the reference's own conjugate_h / conjugate_cnot outputs on seeded matrices and gate lists
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "conjugation_golden.npz"))
    for i in range(8):
        mat, gates, stop = np.array(g["conj_in_%d" % i]), g["conj_gates_%d" % i], int(g["conj_stop_%d" % i])
        if stop < 0:
            css_code.transform_stabilisers(mat, gates)
            assert np.array_equal(mat, g["conj_out_%d" % i])
        else:
            # the reference refuses gate `stop`: everything before it applies, then its partial row swaps
            want = np.array(g["conj_in_%d" % i])
            with pytest.raises(NotImplementedError):
                cpu_ref.transform_stabilisers(want, gates)
            with pytest.raises(NotImplementedError, match="only handles CSS codes"):
                css_code.transform_stabilisers(mat, gates)
            assert np.array_equal(mat, want)
            prefix = np.array(g["conj_in_%d" % i])
            css_code.transform_stabilisers(prefix, gates[:stop])
            assert np.array_equal(prefix, g["conj_out_%d" % i])
    for i in range(4):
        mat = np.array(g["enc_in_%d" % i])
        css_code.transform_stabilisers(mat, g["enc_gates_%d" % i])
        assert np.array_equal(mat, g["enc_out_%d" % i])
    rows = np.array(g["cnot_truth_in"])
    css_code.conjugate_cnot_with_check_mat(rows, 0, 1)
    assert np.array_equal(rows, g["cnot_truth_out"])
    rows = np.array(g["h_truth_in"])
    css_code.conjugate_h_with_check_mat(rows, 0)
    assert np.array_equal(rows, g["h_truth_out"])


@pytest.mark.parametrize("n", [70, 128, 200, 257])
def test_transform_stabilisers_runs_of_cnots_with_one_control(n):
    # gf2_conjugate_gates folds runs of CNOTs that share a control into one operation per 64-column word of targets; against
    # the restatement of css_code.py:737-781 on gate lists made of such runs: long and short, targets in the control's own word
    # (not folded), a target named twice (cancels), single gates of another control in between, H gates that end a run, CNOT(c, c),
    # a Z half that starts inside a word (n not a multiple of 64), and a refused H in the middle of it all
    rng = np.random.default_rng(n)
    k = 40
    for trial in range(4):
        mat = np.zeros((k, 2 * n), dtype=int)
        mat[:, :n] = rng.integers(0, 2, (k, n))
        if trial != 3:
            mat[: k // 2, :n] = 0
            mat[: k // 2, n:] = rng.integers(0, 2, (k // 2, n))            # CSS rows: all X or all Z
        else:
            mat[:, n:] = rng.integers(0, 2, (k, n))                         # rows with both: some H will be refused
        gates = []
        for run in range(12):
            c = int(rng.integers(0, n))
            targets = rng.permutation(n)[: int(rng.integers(1, n))]
            for t in targets:
                gates.append((1, c, int(t)))
            if run % 3 == 0:
                gates.append((1, c, int(targets[0])))                       # named twice
            if run % 4 == 1:
                gates.append((1, c, c))
            if run % 2 == 0:
                gates.append((0, int(rng.integers(0, n)), 0))
            gates.append((1, int(rng.integers(0, n)), int(rng.integers(0, n))))
        gates = np.array(gates, dtype=np.int32)
        want, got = mat.copy(), mat.copy()
        try:
            cpu_ref.transform_stabilisers(want, gates)
            refused = False
        except NotImplementedError:
            refused = True
        if refused:
            with pytest.raises(NotImplementedError, match="only handles CSS codes"):
                css_code.transform_stabilisers(got, gates)
        else:
            css_code.transform_stabilisers(got, gates)
        assert np.array_equal(got & 1, want & 1), (n, trial)
    assert refused                                                          # the last trial must have exercised the refusal


def test_steane_encoders_known_answers(steane_h):
    # test/test_css_code.py:61-106
    code = css_code.CSSCode(steane_h, steane_h)
    n = 7
    prog = code.noisy_encode_zero(range(n))
    assert np.array_equal(prog, cpu_ref.encode_zero_gates(cpu_ref.CSSCode(steane_h, steane_h)))
    mat = np.concatenate((np.zeros((n, n), dtype='int'), np.identity(n, dtype='int')), axis=1)
    for i in range(3):
        if code.parity_check_c2[i, 6] == 1:
            mat[3 + i, :] += mat[6, :]
    mat = np.mod(mat, 2)
    css_code.transform_stabilisers(mat, prog)
    expected = np.zeros((n, 2 * n), dtype='int')
    expected[0:3, 0:7] = code.parity_check_c1
    expected[3:6, 7:14] = code.parity_check_c2
    expected[6, 7:10] = np.transpose(code.parity_check_c1[:, 6:7])
    expected[6, 13:14] = np.identity(1, dtype='int')
    assert np.array_equal(mat, expected)

    prog = code.noisy_encode_plus(range(n))
    assert np.array_equal(prog, cpu_ref.encode_plus_gates(cpu_ref.CSSCode(steane_h, steane_h)))
    mat = np.concatenate((np.zeros((n, n), dtype='int'), np.identity(n, dtype='int')), axis=1)
    css_code.transform_stabilisers(mat, prog)
    expected = np.zeros((n, 2 * n), dtype='int')
    expected[0:3, 0:7] = code.parity_check_c1
    expected[3:6, 7:14] = code.parity_check_c2
    expected[6, 3:6] = np.transpose(code.parity_check_c2[:, 6:7])
    expected[6, 6] = 1
    assert np.array_equal(mat, expected)


def test_transform_stabilisers_argument_errors(steane_h):
    mat = np.concatenate((np.zeros((3, 7), dtype='int'), steane_h), axis=1)
    before = mat.copy()
    with pytest.raises(ValueError, match="qubit index must be within"):
        css_code.transform_stabilisers(mat, [('H', 0), ('CNOT', 1, 7)])
    want = before.copy()
    cpu_ref.transform_stabilisers(want, [(0, 0, 0)])
    assert np.array_equal(mat, want)                         # the H before the bad gate was applied
    with pytest.raises(ValueError, match="cannot conjugate gate"):
        css_code.transform_stabilisers(mat, [('X', 0)])
    css_code.transform_stabilisers(mat, np.zeros((0, 3), dtype=np.int32))
    assert np.array_equal(mat, want)
    empty = np.zeros((0, 14), dtype='int')
    css_code.transform_stabilisers(empty, [('H', 1)])


def test_encoders_of_the_n4096_code_prepare_its_stabilisers(ctx):
    # config 4 of BASELINE.json: the |+> encoder of the random dual code (about 2 million CNOTs) conjugates
    # Z_1 .. Z_n into [H1 | 0], [0 | H2] and the logical X row -- the reference's Steane assertion
    # (test/test_css_code.py:89-106) at full size, where its Python loop over gates x rows cannot finish
    import bench
    code, _, _ = bench.build_code()
    n, r_1, r_2 = code.n, code.r_1, code.r_2
    prog = code.noisy_encode_plus(range(n))
    assert prog.shape[0] > 10 ** 6
    mat = np.concatenate((np.zeros((n, n), dtype=np.uint8), np.identity(n, dtype=np.uint8)), axis=1)
    css_code.transform_stabilisers(mat, prog)
    expected = np.zeros((n, 2 * n), dtype=np.uint8)
    expected[0:r_1, 0:n] = code.parity_check_c1
    expected[r_1:r_1 + r_2, n:2 * n] = code.parity_check_c2
    expected[r_1 + r_2:, r_1:r_1 + r_2] = np.transpose(code.parity_check_c2[:, r_1 + r_2:])
    expected[r_1 + r_2:, r_1 + r_2:n] = np.identity(n - r_1 - r_2, dtype=np.uint8)
    assert np.array_equal(mat, expected)


def test_slab_pipeline_padded_rows_and_several_workspace_passes(ctx, route):
    # (a) rows padded to lde = 71 words (odd: the identity words are not 16-byte aligned) with garbage-free padding,
    # (b) a batch above the 2^21-sample workspace pass, so the pipeline runs twice and carries on at the right row;
    # both against the column-gather kernel on the same resident errors
    r, n, ioff = 1024, 3000, 1900
    rng = np.random.default_rng(2)
    hm = rng.integers(0, 2, (r, n), dtype=np.uint8)
    hm[:, ioff:ioff + r] = np.identity(r, dtype=np.uint8)
    chk = ctx.check_create(_native.pack_rows(hm), r, n)
    words, lde, batch = _native.words_for(n), 71, 40000
    em = (rng.random((batch, n)) < 0.004).astype(np.uint8)
    e = np.zeros((batch, lde), dtype="<u8")
    e[:, :words] = _native.pack_rows(em)
    e_buf = ctx.alloc(e.nbytes).upload(e)
    want = c_oracle.histogram(c_oracle.syndrome_batch(_native.pack_rows(hm), r, n, _native.pack_rows(em)[:5000].copy(), 5000),
                              5000, r, 1, r + 1)
    for count in (batch, 5000):
        hists = []
        for force_gather in (False, True):
            route.force("GF2_SPARSE_GATHER" if force_gather else "GF2_SPARSE_SLABS")
            hist = ctx.alloc((r + 1) * 8).zero()
            ctx.syndrome_sparse_dev(chk, e_buf, count, lde, None, 0, hist, r + 1)
            hists.append(hist.download((r + 1,), np.uint64))
            route.release("GF2_SPARSE_GATHER" if force_gather else "GF2_SPARSE_SLABS")
        assert np.array_equal(hists[0], hists[1]) and int(hists[0].sum()) == count
        if count == 5000:
            assert np.array_equal(hists[0], want)
    # (b)
    r, n, batch = 2048, 4096, (1 << 21) + 12345
    hm = rng.integers(0, 2, (r, n), dtype=np.uint8)
    hm[:, :r] = np.identity(r, dtype=np.uint8)
    chk = ctx.check_create(_native.pack_rows(hm), r, n)
    ex, ez = ctx.alloc(batch * 512), ctx.alloc(batch * 512)
    ctx.sample_errors_dev(n, 9, 0, batch, 0.003, 0.003, 0.003, ex, ez, 64)
    hists = []
    for force_gather in (False, True):
        if force_gather:
            route.force("GF2_SPARSE_GATHER")
        hist = ctx.alloc((r + 1) * 8).zero()
        ctx.syndrome_sparse_dev(chk, ez, batch, 64, None, 0, hist, r + 1)
        hists.append(hist.download((r + 1,), np.uint64))
    assert np.array_equal(hists[0], hists[1]) and int(hists[0].sum()) == batch
    ex.free(), ez.free()


# ---- randomised sweep over shapes around the word, slab and panel boundaries -----------------------------------------------

def _boundary_dim(rng, top):
    pool = [0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 255, 256, 257, 511, 512, 513]
    return int(rng.choice([d for d in pool if d <= top] + [int(rng.integers(0, top + 1))]))


def test_randomised_shapes_rref_nullspace_normalize_syndromes(ctx):
    # 60 seeded draws of (m, n, density, batch): RREF + pivots, nullspace, normalisation (swap list, result or the
    # reference's two exceptions), syndromes with the histogram of weights -- everything against the C oracle
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        m, n = _boundary_dim(rng, 300), _boundary_dim(rng, 600)
        density = float(rng.choice([0.02, 0.1, 0.5, 0.9]))
        a = (rng.random((m, n)) < density).astype(np.uint8)
        if m > 3 and rng.random() < 0.3:
            a[m // 2] = a[0] ^ a[m - 1]                       # a dependent row
        packed = _native.pack_rows(a)
        want, want_piv, want_rank = c_oracle.rref(packed, m, n)
        got = packed.copy()
        piv, rank = ctx.rref(got, m, n)
        assert rank == want_rank and np.array_equal(piv, want_piv) and np.array_equal(got, want), (trial, m, n)
        assert np.array_equal(ctx.nullspace(packed.copy(), m, n), c_oracle.nullspace(packed, m, n)), (trial, m, n)
        if m and n:
            offset = int(rng.integers(0, n))
            rc, want_h, want_swaps = c_oracle.normalize(packed, m, n, offset)
            work = packed.copy()
            if rc == 0:
                assert ctx.normalize(work, m, n, offset) == want_swaps and np.array_equal(work, want_h), (trial, m, n, offset)
            else:
                with pytest.raises(_native.GF2Error) as err:
                    ctx.normalize(work, m, n, offset)
                assert err.value.code == rc, (trial, m, n, offset)
            batch = int(rng.integers(1, 200))
            e = _native.pack_rows((rng.random((batch, n)) < float(rng.choice([0.01, 0.3]))).astype(np.uint8))
            syn = ctx.syndrome_batch(packed, m, n, e, batch)
            want_syn = c_oracle.syndrome_batch(packed, m, n, e, batch)
            assert np.array_equal(syn, want_syn), (trial, m, n, batch)


# ---- the N > 1 product path on a GPU: two gloo ranks share the card, each runs its shard through the real kernels -----------

def _sharded_gpu_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from quantum_css_codes_amd import montecarlo
    from quantum_css_codes_amd.css_code import CSSCode as Code
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GF2_DEVICE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
    code = Code(h, h)
    res = montecarlo.run_sharded(code, 300001, 0.02, 0.01, 0.03, seed=5, first_sample=17)
    dec = montecarlo.decode_sharded(code, 200001, 0.02, 0.01, 0.03, seed=6, first_sample=3)
    np.savez(os.path.join(out_dir, "gpu_rank%d.npz" % rank), hist_z=res['hist_z'], hist_x=res['hist_x'],
             shard=np.array(res['shard']), decode=np.array([dec[f] for f in montecarlo.DECODE_FIELDS]))
    dist.destroy_process_group()


def test_two_rank_sharded_monte_carlo_on_the_gpu(tmp_path, steane_h):
    import socket
    import torch.multiprocessing as mp
    from quantum_css_codes_amd import montecarlo
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    code = CSSCode(steane_h, steane_h)
    whole = code.monte_carlo(300001, 0.02, 0.01, 0.03, seed=5, first_sample=17)
    whole_dec = code.logical_error_rates(200001, 0.02, 0.01, 0.03, seed=6, first_sample=3)
    h1, h2 = c_oracle.pack_rows(code.parity_check_c1), c_oracle.pack_rows(code.parity_check_c2)
    want_z, want_x = c_oracle.mc(h1, 3, h2, 3, 7, 5, 17, 300001, 0.02, 0.01, 0.03, 0)
    r0, r1 = np.load(tmp_path / "gpu_rank0.npz"), np.load(tmp_path / "gpu_rank1.npz")
    assert list(r0["shard"]) == [17, 150001] and list(r1["shard"]) == [150018, 150000]
    for r in (r0, r1):                                        # every rank holds the global result
        assert np.array_equal(r["hist_z"], want_z) and np.array_equal(r["hist_x"], want_x)
        assert np.array_equal(r["hist_z"], whole['hist_z'])
        assert list(r["decode"]) == [whole_dec[f] for f in montecarlo.DECODE_FIELDS]


def test_bench_under_torchrun_two_ranks_share_the_gpu(tmp_path):
    # the driver's N > 1 launch line (python -m torch.distributed.run ... bench.py --gpus N), rehearsed with two ranks that
    # share this box's one GPU over gloo: a fresh child process, the JSON line checked.  (RCCL with N > 1 needs N GPUs.)
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GF2_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--batch-log2", "17", "--no-cpu-baseline", "--no-secondary", "--no-settle"]
    done = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # rank 0 prints the one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["unit"] == "syndromes/s"
    assert out["config"]["global_samples_per_step"] == 2 << 17
    assert out["value"] > 0 and abs(out["value"] - 2 * (2 << 17) / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["checks"]["histogram_total"] == 2 * (2 << 17)   # both shards' samples arrived in the all-reduced histogram


def test_bench_strong_scaling_line_four_ranks_share_the_gpu_same_histogram_as_one_rank(tmp_path):
    # BASELINE.json configs[4] as the driver will launch it on the 8-GPU node (--total-samples 100000000: the global stream of 10^8
    # samples cut into contiguous shards, one all-reduce), rehearsed with as many ranks as this pool lets one GPU carry (6 processes
    # on the card: this test runner, the launcher and 4 ranks; the 8-rank world is rehearsed on the CPU by
    # tests/test_sharding_gloo.py::test_eight_rank_histogram_allreduce): gloo, every rank on GPU 0.  The all-reduced histogram of
    # the 4 ranks must be the 1-rank histogram bin for bin (key convention
    # css_code.py:729, X <-> H2 / Z <-> H1 css_code.py:457-470), one JSON line from rank 0, the shards listed and contiguous.
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GF2_DEVICE="0")
    tail = ["--dist-backend", "gloo", "--steps", "1", "--warmup", "0", "--total-samples", "100000000", "--no-cpu-baseline",
            "--no-secondary", "--no-settle"]

    def run(ranks):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(ranks)] + tail
        if ranks == 1:
            cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + tail
        done = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
        assert done.returncode == 0, done.stderr[-2000:]
        lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1                                 # rank 0 prints the one line
        return json.loads(lines[0])

    one, six = run(1), run(4)
    for out, ranks in ((one, 1), (six, 4)):
        assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["config"]["global_samples_per_step"] == 10**8
        assert out["checks"]["histogram_total"] == 10**8 == out["checks"]["expected_total"]
    assert six["checks"]["histogram_sha256"] == one["checks"]["histogram_sha256"]
    assert one["shards"] is None and one["collective"] is None
    assert six["collective"]["rccl_ranks"] == 0 and six["collective"]["backend"] == "gloo"
    pos = 0
    for r, sh in enumerate(six["shards"]):
        assert sh["rank"] == r and sh["first"] == pos and sh["count"] == 25000000 and sh["roofline"]["frac"] > 0
        pos += sh["count"]
    assert pos == 10**8


def test_context_options_are_validated(ctx):
    for option, bad in ((_native.OPT_SLAB_PASS_LOG2, 11), (_native.OPT_SLAB_PASS_LOG2, 25), (_native.OPT_COMBINE_BLOCKS, 0),
                        (_native.OPT_MC_CHUNK_LOG2, 15), (_native.OPT_MC_CHUNK_LOG2, 23), (_native.OPT_COMBINE_THREADS, 96),
                        (_native.OPT_REDO_BLOCKS_PER_CU, 0), (_native.OPT_REDO_BLOCKS_PER_CU, 65), (_native.OPT_GATHER_REVERSE, 2),
                        (_native.OPT_GATHER_CROSS, 2), (99, 1)):
        with pytest.raises(_native.GF2Error):
            ctx.set_option(option, bad)
    for option, good in ((_native.OPT_SLAB_PASS_LOG2, 20), (_native.OPT_MC_CHUNK_LOG2, 20), (_native.OPT_COMBINE_THREADS, 256),
                         (_native.OPT_COMBINE_BLOCKS, 64), (_native.OPT_REDO_BLOCKS_PER_CU, 16), (_native.OPT_GATHER_REVERSE, 0),
                         (_native.OPT_GATHER_CROSS, 1)):
        ctx.set_option(option, good)
        ctx.set_option(option, None)                                        # back to the default
    with pytest.raises(_native.GF2Error):
        ctx.set_flags(1 << 30)                                              # not a defined flag
    before = ctx.get_flags()
    with ctx.flags(_native.F_MC_ROWS | _native.F_COMBINE_FOLDED):
        assert ctx.get_flags() == before | _native.F_MC_ROWS | _native.F_COMBINE_FOLDED
    assert ctx.get_flags() == before


def test_two_contexts_share_checks_and_run_concurrently(ctx):
    # bench.py issues the two components of a step on two contexts (two HIP streams, two workspaces) that share the prepared
    # checks and the resident errors; 30 overlapping steps must accumulate exactly 30 times the one-stream histograms
    r1, r2, n, batch = 1024, 1023, 2304, 40000
    rng = np.random.default_rng(5)
    h1 = rng.integers(0, 2, (r1, n), dtype=np.uint8)
    h1[:, :r1] = np.identity(r1, dtype=np.uint8)
    h2 = rng.integers(0, 2, (r2, n), dtype=np.uint8)
    h2[:, n - r2:] = np.identity(r2, dtype=np.uint8)
    c1, c2 = ctx.check_create(_native.pack_rows(h1), r1, n), ctx.check_create(_native.pack_rows(h2), r2, n)
    lde = _native.words_for(n)
    ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
    ctx.sample_errors_dev(n, 8, 0, batch, 0.004, 0.003, 0.005, ex, ez, lde)
    one_z, one_x = ctx.alloc((r1 + 1) * 8).zero(), ctx.alloc((r2 + 1) * 8).zero()
    ctx.syndrome_sparse_dev(c1, ez, batch, lde, None, 0, one_z, r1 + 1)
    ctx.syndrome_sparse_dev(c2, ex, batch, lde, None, 0, one_x, r2 + 1)
    wz, wx = one_z.download((r1 + 1,), np.uint64), one_x.download((r2 + 1,), np.uint64)
    other = _native.Context(ctx.device)
    try:
        hz, hx = ctx.alloc((r1 + 1) * 8).zero(), ctx.alloc((r2 + 1) * 8).zero()
        ctx.sync()
        for _ in range(30):
            ctx.syndrome_sparse_dev(c1, ez, batch, lde, None, 0, hz, r1 + 1)
            other.syndrome_sparse_dev(c2, ex, batch, lde, None, 0, hx, r2 + 1)
        ctx.sync(), other.sync()
        assert np.array_equal(hz.download((r1 + 1,), np.uint64), wz * np.uint64(30))
        assert np.array_equal(hx.download((r2 + 1,), np.uint64), wx * np.uint64(30))
    finally:
        other.sync()
        other.close()


@pytest.mark.parametrize("shape", [(1, 1), (3, 7), (64, 64), (64, 65), (33, 128), (64, 200), (64, 512), (60, 1000), (64, 1024),
                                   (65, 100), (128, 256), (100, 500), (128, 1024), (100, 1000), (70, 1024), (129, 250), (256, 256), (200, 512), (256, 512)])
def test_rref_batch_small_matrix_kernel_every_variant(shape, ctx, route):
    # the wavefront-per-matrix kernel (rows in registers): every (rows per lane, words per row) instantiation, batches of
    # different matrices incl. rank-deficient and zero ones; pivots, ranks and matrices against the C oracle, and the
    # blocked path on the same input
    m, n = shape
    rng = np.random.default_rng(m * 1000 + n)
    batch = 5
    mats = []
    for b in range(batch):
        a = (rng.random((m, n)) < (0.5 if b < 3 else 0.03)).astype(np.uint8)
        if b == 1 and m > 2:
            a[m - 1] = a[0] ^ a[1]
        if b == 4:
            a[:] = 0
        mats.append(_native.pack_rows(a))
    packed = np.ascontiguousarray(np.stack(mats))
    got = packed.copy()
    piv, ranks = ctx.rref_batch(got, batch, m, n)
    for b in range(batch):
        want, want_piv, want_rank = c_oracle.rref(packed[b], m, n)
        assert int(ranks[b]) == want_rank and np.array_equal(piv[b][:want_rank], want_piv), (shape, b)
        assert np.array_equal(got[b], want), (shape, b)
    # the ways the pivot rows reach the other rows (GF2_OPT_RREF_SMALL_BCAST): one at a time through LDS (0) or v_readlane (1), four
    # at a time through a table of their sums (2: where rows are whole 16-byte pieces and the registers allow; else as the default)
    for how in (0, 1, 2):
        ctx.set_option(_native.OPT_RREF_SMALL_BCAST, how)
        try:
            other = packed.copy()
            piv_o, ranks_o = ctx.rref_batch(other, batch, m, n)
        finally:
            ctx.set_option(_native.OPT_RREF_SMALL_BCAST, None)
        assert np.array_equal(other, got) and np.array_equal(ranks_o, ranks), (shape, how)
        for b in range(batch):
            assert np.array_equal(piv_o[b][:int(ranks[b])], piv[b][:int(ranks[b])]), (shape, how, b)
    route.force("GF2_RREF_NO_SMALL")
    again = packed.copy()
    piv2, ranks2 = ctx.rref_batch(again, batch, m, n)
    assert np.array_equal(again, got) and np.array_equal(ranks2, ranks)


@pytest.mark.parametrize("k", [4, 2, 0])
@pytest.mark.parametrize("shape", [(2048, 4096, 2), (300, 2500, 3), (1025, 1030, 3), (3000, 1000, 2), (520, 8200, 9), (129, 65, 33)])
def test_rref_sweep_routes(shape, k, ctx):
    # Round 5: the blocked RREF below 4097 rows with K = 4 and K = 2 panels per sweep (fused right-looking panel kernel, K-table trailing
    # pass, side buffer of the next sweep's column words, few-pivot tail) and round 4's pair kernels (K = 0), forced through the
    # internal option, with small row blocks too: same reduced forms, pivots and ranks as the oracle's (bin_matrix.py:8-34).  The
    # batches hold dense and sparse matrices, leading pivot-free panels (a sweep without pivots: the side buffer is not written, the
    # next panel kernel reads the rows), rank-deficient matrices (several rounds per panel) and every-other-column-zero matrices;
    # 520 x 8200 has a ragged last chunk, 3000 x 1000 more rows than columns, 129 x 65 a partial only panel.
    m, n, batch = shape
    rng = np.random.default_rng(m * 11 + n + batch)
    mats = []
    for b in range(batch):
        a = (rng.random((m, n)) < (0.5 if b % 3 != 1 else 0.03)).astype(np.uint8)
        if b % 4 == 2:
            a[:, :min(n, 200)] = 0
        if b % 5 == 3 and m >= 2:
            a[m // 2:] = a[: m - m // 2]
        if b % 7 == 6:
            a[:, ::2] = 0
        mats.append(a)
    want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
    flags = ctx.get_flags()
    try:
        ctx.set_flags(flags | _native.F_RREF_NO_SMALL)               # (129 x 65 would take the wavefront-per-matrix kernel)
        ctx.set_option(_native.OPT_RREF_SWEEP_K, k)
        for rows_wg in ((None, 128) if k else (None,)):
            ctx.set_option(_native.OPT_RREF_ROWS_WG, rows_wg)
            packed = np.stack([_native.pack_rows(a) for a in mats])
            pivots, ranks = ctx.rref_batch(packed, batch, m, n)
            for b in range(batch):
                assert ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]), (k, rows_wg, b)
                assert list(pivots[b, :want[b][2]]) == list(want[b][1]), (k, rows_wg, b)
    finally:
        ctx.set_flags(flags)
        ctx.set_option(_native.OPT_RREF_SWEEP_K, None)
        ctx.set_option(_native.OPT_RREF_ROWS_WG, None)


@pytest.mark.parametrize("k", [4, 2])
@pytest.mark.parametrize("shape", [(300, 600, 400), (1030, 1100, 300), (257, 4200, 260), (2048, 2100, 130)])
def test_rref_large_batches_read_pivot_rows_in_place(shape, k, ctx):
    # Round 5: when the batch is large enough for a trailing-pass workgroup to own ALL rows of its chunk, the panel kernel leaves the
    # pivot rows' numbers instead of snapshots and the pass builds its tables from the rows where they lie (before it writes any).
    # Mixed batches as in test_rref_sweep_routes, K = 4 and 2, with the snapshots kept beside it (internal option): the oracle's
    # matrices, pivots and ranks (bin_matrix.py:8-34).
    m, n, batch = shape
    rng = np.random.default_rng(m * 7 + n + batch)
    distinct = []
    for b in range(24):
        a = (rng.random((m, n)) < (0.5 if b % 3 != 1 else 0.03)).astype(np.uint8)
        if b % 4 == 2:
            a[:, :min(n, 200)] = 0
        if b % 5 == 3:
            a[m // 2:] = a[: m - m // 2]
        if b % 7 == 6:
            a[:, ::2] = 0
        distinct.append(a)
    want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in distinct]
    packed_one = [_native.pack_rows(a) for a in distinct]
    flags = ctx.get_flags()
    try:
        ctx.set_flags(flags | _native.F_RREF_NO_SMALL)
        ctx.set_option(_native.OPT_RREF_SWEEP_K, k)
        for keep_snapshots in (None, 1):
            ctx.set_option(_native.OPT_RREF_STREAM_VARIANT, keep_snapshots)
            packed = np.stack([packed_one[b % 24] for b in range(batch)])
            pivots, ranks = ctx.rref_batch(packed, batch, m, n)
            for b in range(batch):
                w = want[b % 24]
                assert ranks[b] == w[2] and np.array_equal(packed[b], w[0]), (k, keep_snapshots, b)
                assert list(pivots[b, :w[2]]) == list(w[1]), (k, keep_snapshots, b)
    finally:
        ctx.set_flags(flags)
        ctx.set_option(_native.OPT_RREF_SWEEP_K, None)
        ctx.set_option(_native.OPT_RREF_STREAM_VARIANT, None)


@pytest.mark.parametrize("shape", [(4100, 700, 2), (5000, 5100, 3), (8200, 8300, 2), (9000, 2000, 1), (4500, 4600, 7), (4097, 65, 2)])
def test_rref_streamed_sweeps_on_tall_matrices(shape, ctx, route):
    # Round 5: more than 4096 rows take four panels per sweep with the rows STREAMED (sweep_column / sweep_stream_panel / sweep_finish /
    # sweep_snapshot kernels + the K = 4 trailing pass), the next sweep's panels on a side stream under the pass.  Mixed batches as in
    # test_rref_sweep_routes -- dense, sparse (several window rounds per panel), leading pivot-free panels (a sweep without pivots:
    # sweep_column_kernel refills the side buffer), duplicated halves (rank deficient: every sweep runs), every other column zero --
    # with and without look-ahead, and with the pair kernels (K = 0) beside them: oracle's matrices, pivots, ranks (bin_matrix.py:8-34).
    m, n, batch = shape
    rng = np.random.default_rng(m * 13 + n + batch)
    mats = []
    for b in range(batch):
        a = (rng.random((m, n)) < (0.5 if b % 3 != 1 else 0.02)).astype(np.uint8)
        if b % 4 == 2:
            a[:, :min(n, 300)] = 0
        if b % 5 == 3:
            a[m // 2:] = a[: m - m // 2]
        if b % 7 == 6:
            a[:, ::2] = 0
        mats.append(a)
    want = [c_oracle.rref(c_oracle.pack_rows(a), m, n) for a in mats]
    try:
        for k, flag in ((None, None), (None, "GF2_RREF_NO_LOOKAHEAD"), (0, None)):
            ctx.set_option(_native.OPT_RREF_SWEEP_K, k)
            if flag:
                route.force(flag)
            packed = np.stack([_native.pack_rows(a) for a in mats])
            pivots, ranks = ctx.rref_batch(packed, batch, m, n)
            if flag:
                route.release(flag)
            for b in range(batch):
                assert ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]), (k, flag, b)
                assert list(pivots[b, :want[b][2]]) == list(want[b][1]), (k, flag, b)
    finally:
        ctx.set_option(_native.OPT_RREF_SWEEP_K, None)
