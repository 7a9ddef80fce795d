"""
Seeded random sweeps of the HIP path against the C oracle (oracle/gf2_oracle.c, pinned by tests/test_oracle_golden.py): shapes
nobody picked by hand -- row and column counts around the 64-bit word, the 64-column panel, the 128-row window and the 1024-lane
workgroup, padded leading dimensions, densities from empty to full, batches of unequal matrices.  Bit-exact, as everywhere.
"""
import numpy as np
import pytest

from oracle import c_oracle
from quantum_css_codes_amd import _native

pytestmark = pytest.mark.gpu

EDGES = (1, 2, 3, 31, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 500, 1023, 1024, 1025, 1500, 2047, 2049)


@pytest.fixture(scope="module")
def ctx():
    return _native.default_context()


def random_matrix(rng, m, n, density, pad_words=0):
    a = (rng.random((m, n)) < density).astype(np.uint8)
    kind = rng.integers(0, 6)
    if kind == 0 and m > 2:
        a[rng.integers(0, m)] = a[0] ^ a[m - 1]                       # a dependent row
    if kind == 1 and n > 3:
        a[:, rng.integers(0, n)] = 0                                  # a pivot-free column
    if kind == 2 and n > 70:
        a[:, : int(rng.integers(1, min(n, 200)))] = 0                 # empty leading columns
    if kind == 3 and m > 4:
        a[m // 2:] = a[: m - m // 2]                                  # rank at most m / 2
    packed = _native.pack_rows(a)
    if pad_words:                                                     # a leading dimension wider than the matrix needs
        wide = np.zeros((m, packed.shape[1] + pad_words), dtype="<u8")
        wide[:, : packed.shape[1]] = packed
        packed = wide
    return a, packed


def test_fuzz_rref_nullspace_single_matrices(ctx):
    rng = np.random.default_rng(20260)
    for case in range(60):
        m, n = int(rng.choice(EDGES)), int(rng.choice(EDGES))
        if case % 7 == 0:
            m, n = int(rng.integers(1, 700)), int(rng.integers(1, 3000))
        density = float(rng.choice([0.0, 0.01, 0.1, 0.5, 0.9, 1.0]))
        a, packed = random_matrix(rng, m, n, density, pad_words=int(rng.integers(0, 3)))
        want, want_piv, want_rank = c_oracle.rref(packed, m, n)
        null_want = c_oracle.nullspace(packed, m, n)
        got = packed.copy()
        pivots, rank = ctx.rref(got, m, n)
        label = "case %d: %d x %d, density %.2f, ld %d" % (case, m, n, density, packed.shape[1])
        assert rank == want_rank and list(pivots) == list(want_piv), label
        assert np.array_equal(got, want), label
        null_got = ctx.nullspace(packed.copy(), m, n)
        assert np.array_equal(null_got, null_want), label


def test_fuzz_rref_batches(ctx):
    rng = np.random.default_rng(20261)
    for case in range(25):
        m, n = int(rng.choice(EDGES[3:])), int(rng.choice(EDGES[3:]))
        batch = int(rng.integers(1, 12))
        mats = [random_matrix(rng, m, n, float(rng.choice([0.02, 0.3, 0.5, 1.0])))[1] for _ in range(batch)]
        packed = np.stack(mats)
        want = [c_oracle.rref(mat, m, n) for mat in mats]
        original = packed.copy()
        pivots, ranks = ctx.rref_batch(packed, batch, m, n)
        for b in range(batch):
            label = "case %d matrix %d: %d x %d" % (case, b, m, n)
            assert ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]), label
            assert list(pivots[b, : want[b][2]]) == list(want[b][1]), label
        # small matrices once more with four pivots at a time forced (GF2_OPT_RREF_SMALL_BCAST = 2; other sizes take their usual path)
        ctx.set_option(_native.OPT_RREF_SMALL_BCAST, 2)
        try:
            again = original.copy()
            pivots2, ranks2 = ctx.rref_batch(again, batch, m, n)
        finally:
            ctx.set_option(_native.OPT_RREF_SMALL_BCAST, None)
        assert np.array_equal(again, packed) and np.array_equal(ranks2, ranks), "case %d: %d x %d, four pivots at a time" % (case, m, n)


def test_fuzz_rref_tall_matrices(ctx):
    # more than 4096 rows: the streamed K = 4 sweeps with look-ahead (round 5), batches of unequal matrices, padded leading
    # dimensions, column counts around the word, the panel, the sweep of four panels and the 16-word chunk of the trailing pass
    rng = np.random.default_rng(20265)
    cols = (1, 63, 64, 65, 255, 256, 257, 300, 1023, 1024, 1025, 1100, 2100)
    for case in range(14):
        m = int(rng.choice([4097, 4100, 5000, 6143, 8192, 8193, 9001]))
        n = int(rng.choice(cols))
        batch = int(rng.integers(1, 4))
        pad = int(rng.integers(0, 3))
        mats = [random_matrix(rng, m, n, float(rng.choice([0.003, 0.3, 0.5, 1.0])), pad_words=pad)[1] for _ in range(batch)]
        packed = np.stack(mats)
        want = [c_oracle.rref(mat, m, n) for mat in mats]
        pivots, ranks = ctx.rref_batch(packed, batch, m, n)
        for b in range(batch):
            label = "case %d matrix %d: %d x %d, ld %d" % (case, b, m, n, packed.shape[2])
            assert ranks[b] == want[b][2] and np.array_equal(packed[b], want[b][0]), label
            assert list(pivots[b, : want[b][2]]) == list(want[b][1]), label


def test_fuzz_normalize(ctx):
    rng = np.random.default_rng(20262)
    done = 0
    for case in range(80):
        r = int(rng.choice(EDGES[:18]))
        n = r + int(rng.integers(0, 400))
        offset = int(rng.integers(0, n - r + 1))
        a, packed = random_matrix(rng, r, n, float(rng.choice([0.2, 0.5, 0.8])))
        rc, want, want_swaps = c_oracle.normalize(packed, r, n, offset)
        label = "case %d: %d x %d, offset %d" % (case, r, n, offset)
        got = packed.copy()
        if rc != 0:                                                    # the reference raises for this matrix: so must the drop-in
            with pytest.raises(_native.GF2Error):
                ctx.normalize(got, r, n, offset)
            continue
        swaps = ctx.normalize(got, r, n, offset)
        assert swaps == want_swaps, label
        assert np.array_equal(got, want), label
        done += 1
    assert done >= 20


def test_fuzz_syndromes_host_batches(ctx):
    rng = np.random.default_rng(20263)
    for case in range(40):
        r, n = int(rng.choice(EDGES)), int(rng.choice(EDGES))
        batch = int(rng.choice([1, 2, 63, 64, 65, 200, 1000]))
        hm = (rng.random((r, n)) < rng.choice([0.05, 0.5])).astype(np.uint8)
        if case % 3 == 0 and n >= r:                                   # a standard form: an identity block somewhere
            off = int(rng.integers(0, n - r + 1))
            hm[:, off:off + r] = np.identity(r, dtype=np.uint8)
        em = (rng.random((batch, n)) < rng.choice([0.0, 0.01, 0.3, 1.0])).astype(np.uint8)
        h, e = _native.pack_rows(hm), _native.pack_rows(em)
        want = c_oracle.syndrome_batch(h, r, n, e, batch)
        got = ctx.syndrome_batch(h, r, n, e, batch)
        assert np.array_equal(got, want), "case %d: check %d x %d, %d errors" % (case, r, n, batch)


def test_fuzz_sparse_weight_histograms_three_routes(ctx):
    # gf2_syndrome_sparse_dev's three implementations on random standard-form checks: the LDS row-slab pipeline (forced: the
    # batches are small), the wavefront-per-sample column gather, and whatever the library picks by itself; identity block at a
    # random column (word-aligned, dword-aligned or neither), rates from nothing to records that overflow, ragged batches
    rng = np.random.default_rng(20264)
    for case in range(24):
        r = int(rng.choice([65, 130, 300, 512, 513, 1000, 1536, 2040, 2047, 2048]))
        extra = int(rng.integers(1, 2400))
        n = min(r + extra, 4096 if r > 1024 else r + extra)
        n = max(n, r + 1)
        align = int(rng.choice([1, 32, 64]))
        off = int(rng.integers(0, (n - r) // align + 1)) * align
        batch = int(rng.choice([1, 63, 64, 65, 500, 1300, 3000]))
        density = float(rng.choice([0.0, 0.002, 0.006, 0.012, 0.03]))
        hm = rng.integers(0, 2, (r, n)).astype(np.uint8)
        hm[:, off:off + r] = np.identity(r, dtype=np.uint8)
        em = (rng.random((batch, n)) < density).astype(np.uint8)
        if batch > 2:
            em[int(rng.integers(0, batch))] = 1                        # every column at once
        h, e = _native.pack_rows(hm), _native.pack_rows(em)
        chk = ctx.check_create(h, r, n)
        lde = e.shape[1]
        e_buf = ctx.alloc(e.nbytes).upload(e)
        want = c_oracle.histogram(c_oracle.syndrome_batch(h, r, n, e, batch), batch, r, 1, r + 1)
        label = "case %d: check %d x %d, identity at %d, %d errors at %.3f" % (case, r, n, off, batch, density)
        for flags in (_native.F_SPARSE_SLABS, _native.F_SPARSE_GATHER, 0):
            keep = ctx.get_flags()
            ctx.set_flags(keep | flags)
            try:
                hist = ctx.alloc((r + 1) * 8).zero()
                ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, None, 0, hist, r + 1)
                got = hist.download((r + 1,), np.uint64)
                hist.free()
            finally:
                ctx.set_flags(keep)
            assert np.array_equal(got, want), label + ", flags %x" % flags
        # ... and the syndromes themselves, stored by the slab pipeline (round 4) and by the column-gather kernel, at the tightest
        # pitch and a wider one, with and without the histogram
        want_s = c_oracle.syndrome_batch(h, r, n, e, batch)
        for flags in (_native.F_SPARSE_SLABS, _native.F_SPARSE_GATHER):
            keep = ctx.get_flags()
            ctx.set_flags(keep | flags)
            try:
                lds = want_s.shape[1] + int(rng.integers(0, 3))
                s_buf = ctx.alloc(batch * lds * 8).zero()
                with_hist = bool(rng.integers(0, 2))
                hist = ctx.alloc((r + 1) * 8).zero()
                ctx.syndrome_sparse_dev(chk, e_buf, batch, lde, s_buf, lds, hist if with_hist else None, r + 1 if with_hist else 0)
                got_s = s_buf.download((batch, lds), np.uint64)[:, :want_s.shape[1]]
                got = hist.download((r + 1,), np.uint64)
                s_buf.free(), hist.free()
            finally:
                ctx.set_flags(keep)
            assert np.array_equal(got_s, want_s), label + ", stored syndromes, flags %x" % flags
            assert not with_hist or np.array_equal(got, want), label + ", histogram beside the syndromes, flags %x" % flags
        e_buf.free()


def test_fuzz_monte_carlo_random_codes_and_rates(ctx):
    # gf2_mc_run on random pairs of standard-form checks (H1 = [I | A], H2 = [A' | I | c] with k leftover columns, or no identity
    # block at all) against the oracle's sequential statement: whatever route the library picks for the size and the rates --
    # fused small-code kernel, lane-per-sample kernel, record sampler + slab pipeline, packed rows -- the histograms are the
    # oracle's, from a sample index far into the stream
    rng = np.random.default_rng(20265)
    for case in range(14):
        n = int(rng.choice([7, 15, 63, 64, 65, 127, 255, 511, 1000, 2048, 3000, 4096]))
        r1 = max(1, int(n * rng.choice([0.25, 0.4, 0.5])))
        k = int(rng.choice([0, 1, 2, 5]))
        r2 = max(1, min(n - r1 - k, n - 1)) if n - r1 - k > 0 else 1
        h1 = rng.integers(0, 2, (r1, n)).astype(np.uint8)
        h2 = rng.integers(0, 2, (r2, n)).astype(np.uint8)
        if case % 4 != 3:
            h1[:, :r1] = np.identity(r1, dtype=np.uint8)
            off2 = n - k - r2
            h2[:, off2:off2 + r2] = np.identity(r2, dtype=np.uint8)
        p = [float(v) for v in rng.choice([0.0, 0.001, 0.0033, 0.01, 0.05], size=3)]
        count = int(rng.choice([1, 777, 20000, 70001])) if n > 64 else int(rng.choice([1, 1000, 300000]))
        first = int(rng.integers(0, 2**40))
        seed = int(rng.integers(0, 2**63))
        hp1, hp2 = _native.pack_rows(h1), _native.pack_rows(h2)
        c1, c2 = ctx.check_create(hp1, r1, n), ctx.check_create(hp2, r2, n)
        want = c_oracle.mc(hp1, r1, hp2, r2, n, seed, first, count, p[0], p[1], p[2], 1)
        got = ctx.mc_run(c1, c2, seed, first, count, p[0], p[1], p[2], _native.HIST_WEIGHT)
        label = "case %d: n %d, r1 %d, r2 %d, k %d, p %s, %d samples from %d" % (case, n, r1, r2, k, p, count, first)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), label
        assert int(got[0].sum()) == count and int(got[1].sum()) == count, label


def test_fuzz_syndrome_table_random_checks():
    # css_code.syndrome_table (the device search behind it: one word, two words, position lists) on random checks of random
    # widths against the oracle's restatement of css_code.py:715-735: same t, same keys in the same insertion order, same errors
    from oracle import cpu_ref
    from quantum_css_codes_amd import css_code
    rng = np.random.default_rng(20266)
    for case in range(18):
        n = int(rng.choice([5, 10, 23, 40, 63, 64, 65, 100, 128, 129, 200, 300]))
        r = int(rng.integers(3, 13)) if n > 30 else int(rng.integers(2, min(n, 9)))
        h = rng.integers(0, 2, (r, n))
        if case % 3 == 0:
            h[:, :r] = np.identity(r, dtype=int)
        if case % 5 == 4:
            h[:, int(rng.integers(0, n))] = 0                          # a weight-1 error with the zero syndrome
        cap = 2 if n > 64 else (3 if n > 30 else None)
        t, table = css_code.syndrome_table(h, max_weight=cap)
        want_t, want = cpu_ref.syndrome_table(h, max_weight=cap)
        label = "case %d: check %d x %d, cap %s" % (case, r, n, cap)
        assert t == want_t, label
        assert list(table.keys()) == [int(k) for k in want.keys()], label
        for k in list(table.keys())[:: max(1, len(table) // 300)]:
            assert np.array_equal(table[k], np.asarray(want[k], dtype=int)), label


def test_fuzz_css_code_constructor_random_dual_pairs():
    # CSSCode(H1, H2) on random dual pairs of random sizes (H2 = some rows of nullspace(H1), so H1 . H2^T = 0 by construction):
    # both standard forms, the logical operators, t and k, the stabiliser labels and the gate set as the oracle's restatement of
    # css_code.py:32-75 makes them; a pair the reference rejects (InvalidCodeError out of normalize_parity_check) must be
    # rejected here too
    from oracle import cpu_ref
    from quantum_css_codes_amd import bin_matrix
    from quantum_css_codes_amd.css_code import CSSCode
    from quantum_css_codes_amd.errors import InvalidCodeError
    rng = np.random.default_rng(20267)
    built = 0
    for case in range(30):
        n = int(rng.choice([6, 9, 16, 31, 40, 64, 65, 100, 130, 200]))
        r1 = int(rng.integers(1, max(2, n // 2)))
        h1 = rng.integers(0, 2, (r1, n))
        if case % 6 != 5:                                              # (every sixth pair keeps whatever rank it has)
            while bin_matrix.rank(h1) < r1:
                h1 = rng.integers(0, 2, (r1, n))
        null = bin_matrix.nullspace(h1)
        if null.shape[0] < 2:
            continue
        # the reference supports one logical qubit: all but one row of the dual's basis (now and then one row fewer, which it rejects)
        r2 = null.shape[0] - (1 if case % 7 != 6 else 2)
        rows = rng.choice(null.shape[0], size=r2, replace=False)
        h2 = null[np.sort(rows)]
        label = "case %d: n %d, r1 %d, r2 %d" % (case, n, r1, r2)
        try:
            want = cpu_ref.CSSCode(h1.copy(), h2.copy(), max_table_weight=1)
        except Exception as err:                                        # noqa: BLE001 -- whatever the reference's logic raises
            with pytest.raises((InvalidCodeError, ValueError)):
                CSSCode(h1.copy(), h2.copy(), max_table_weight=1)
            assert isinstance(err, (cpu_ref.InvalidCodeError, ValueError)), (label, repr(err))
            continue
        code = CSSCode(h1.copy(), h2.copy(), max_table_weight=1)
        assert np.array_equal(code.parity_check_c1, want.parity_check_c1), label
        assert np.array_equal(code.parity_check_c2, want.parity_check_c2), label
        assert (code.n, code.k, code.t) == (want.n, want.k, want.t), label
        assert np.array_equal(code.z_operator_matrix(), want.z_operator_matrix()), label
        assert np.array_equal(code.x_operator_matrix(), want.x_operator_matrix()), label
        assert code.stabilisers() == want.stabiliser_labels(), label
        assert sorted(code._transversal_gates) == sorted(want._transversal_gates), label
        built += 1
    assert built >= 8


def test_fuzz_decode_and_tally_random_small_codes():
    # table decoding and the logical-error tally on random one-logical-qubit CSS codes small enough for
    # full tables, against the statement built on the reference's dict tables (cpu_ref.decode_and_tally) and, at a larger sample
    # count, against the C oracle's packed statement
    from oracle import cpu_ref
    from quantum_css_codes_amd import bin_matrix
    from quantum_css_codes_amd.css_code import CSSCode
    from quantum_css_codes_amd.montecarlo import dense_table, packed_word
    rng = np.random.default_rng(20268)
    fields = ('logical_x', 'logical_z', 'logical_any', 'uncorrectable_x', 'uncorrectable_z')
    done = 0
    for case in range(40):
        n = int(rng.choice([5, 7, 9, 11, 13, 15]))
        r1 = int(rng.integers(1, n - 2))
        h1 = rng.integers(0, 2, (r1, n))
        if bin_matrix.rank(h1) < r1:
            continue
        null = bin_matrix.nullspace(h1)
        rows = np.sort(rng.choice(null.shape[0], size=null.shape[0] - 1, replace=False))
        h2 = null[rows]
        try:
            want_code = cpu_ref.CSSCode(h1.copy(), h2.copy())
        except Exception:                                               # noqa: BLE001 -- pairs the reference rejects are another test's
            continue
        code = CSSCode(h1.copy(), h2.copy())
        p = [float(v) for v in rng.choice([0.01, 0.05, 0.1, 0.2], size=3)]
        seed, first = int(rng.integers(0, 2**31)), int(rng.integers(0, 2**33))
        label = "case %d: n %d, r1 %d, r2 %d, t %d, p %s" % (case, n, r1, h2.shape[0], code.t, p)
        small = code.logical_error_rates(250, *p, seed=seed, first_sample=first)
        assert [small[f] for f in fields] == cpu_ref.decode_and_tally(want_code, seed, first, 250, *p), label
        big = code.logical_error_rates(150000, *p, seed=seed, first_sample=first)
        want = c_oracle.mc_decode(c_oracle.pack_rows(code.parity_check_c1), code.r_1, c_oracle.pack_rows(code.parity_check_c2),
                                  code.r_2, code.n, dense_table(code._c1_syndromes, code.r_1, code.n),
                                  dense_table(code._c2_syndromes, code.r_2, code.n), packed_word(code.x_operator_matrix()[0]),
                                  packed_word(code.z_operator_matrix()[0]), seed, first, 150000, *p)
        assert [big[f] for f in fields] == [int(v) for v in want], label
        done += 1
    assert done >= 6


def test_fuzz_hashed_tables_and_decode_random_dual_pairs():
    # random k = 1 dual pairs of 26 .. 128 qubits with checks on either side of the 24-row and 63-row marks: both syndrome tables
    # through the device hash table (css_code.py:715-735; keys of one or two words) against the C oracle's enumeration -- t, keys
    # in the reference's insertion order, errors --, then the table decode and logical tally (css_code.py:649-685, 599-646) of
    # 10^5 sampled errors through gf2_mc_decode_hashed against the oracle's tally with the oracle's own tables
    from quantum_css_codes_amd import bin_matrix, css_code, montecarlo
    from quantum_css_codes_amd.css_code import CSSCode
    rng = np.random.default_rng(20268)
    for case in range(10):
        n = int(rng.choice([26, 40, 51, 57, 63, 64, 65, 90, 127, 128]))
        r1 = int(rng.integers(max(2, n // 8), n - 3))
        while True:
            h1 = rng.integers(0, 2, (r1, n))
            if bin_matrix.rank(h1) == r1:
                break
        null = bin_matrix.nullspace(h1)
        h2 = null[: null.shape[0] - 1]
        cap = int(rng.choice([1, 2, 3])) if n > 70 else int(rng.choice([2, 3, 4]))
        code = CSSCode(h1, h2, max_table_weight=cap)
        label = "case %d: n %d, r1 %d, r2 %d, cap %d" % (case, n, code.r_1, code.r_2, cap)
        tables = []
        for h, table in ((code.parity_check_c1, code._c1_syndromes), (code.parity_check_c2, code._c2_syndromes)):
            r = h.shape[0]
            t, keys, errs = c_oracle.syndrome_table(c_oracle.pack_rows(h), r, n, cap)
            assert list(table.keys()) == keys, label
            assert np.array_equal(np.array(list(table.values())), c_oracle.unpack_rows(errs, n)), label
            tables.append((t, keys, errs))
        assert code.t == min(tables[0][0], tables[1][0]), label
        p = [float(v) for v in rng.choice([0.0, 0.002, 0.01, 0.03, 0.1], size=3)]
        count, first = 100000, int(rng.integers(0, 1 << 40))
        got = montecarlo.decode_local(code, count, *p, seed=case, first_sample=first, hashed=True)
        want = c_oracle.mc_decode_wide(c_oracle.pack_rows(code.parity_check_c1), code.r_1, c_oracle.pack_rows(code.parity_check_c2), code.r_2,
                                       n, tables[0][1], tables[0][2], tables[1][1], tables[1][2],
                                       c_oracle.pack_rows(code.x_operator_matrix())[0], c_oracle.pack_rows(code.z_operator_matrix())[0],
                                       case, first, count, *p)
        assert [got[f] for f in montecarlo.DECODE_FIELDS] == [int(v) for v in want], label
