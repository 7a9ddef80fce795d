"""
syndrome_table (css_code.py:715-735) and the table decode (css_code.py:649-685, 599-646) beyond 24 checks: the hash tables of
gf2_table.hip against the C oracle's restatement (oracle/gf2_oracle.c: orc_syndrome_table, orc_mc_decode_wide, pinned by
tests/test_oracle_golden.py against the reference's own tables).  A k = 1 CSS code has r_1 + r_2 = n - 1, so every such code
from n = 51 on has a check with more than 24 rows: SURVEY.md 8f items 1 and 2 name n up to 127.
"""
import numpy as np
import pytest

from oracle import c_oracle
from quantum_css_codes_amd import _native, bin_matrix, css_code, montecarlo
from quantum_css_codes_amd.css_code import CSSCode

pytestmark = pytest.mark.gpu


def dual_pair(rng, n, r1):
    """H1 (r1 x n, full rank) and all but one row of a basis of its dual: a k = 1 CSS pair (H1 . H2^T = 0 by construction)."""
    while True:
        h1 = rng.integers(0, 2, (r1, n))
        if bin_matrix.rank(h1) == r1:
            break
    null = bin_matrix.nullspace(h1)
    return h1, null[: null.shape[0] - 1]


def assert_table_is_the_oracles(table, t, h, cap):
    r, n = h.shape
    want_t, want_keys, want_errs = c_oracle.syndrome_table(c_oracle.pack_rows(h), r, n, cap)
    assert t == want_t
    assert list(table.keys()) == want_keys                           # the reference's insertion order too
    got = np.array(list(table.values()), dtype=np.int64).reshape(len(table), n)
    assert np.array_equal(got, c_oracle.unpack_rows(want_errs, n))
    return want_t, len(want_keys)


@pytest.mark.parametrize("case", [(25, 40, None), (30, 60, 4), (31, 63, None), (40, 64, 4), (48, 65, 3), (63, 127, 3), (63, 128, 2),
                                  (64, 80, 3), (70, 80, 3), (100, 104, 3), (127, 128, 2), (26, 300, 2), (40, 1000, 2),
                                  (33, 8192, 1), (90, 700, 2), (28, 29, None), (35, 36, 6), (32, 40, None), (60, 70, 4)])
def test_syndrome_table_hashed_random_checks(case):
    # (a random check of high rate has a small distance and collides by itself; one of low rate would not within 2^28 errors: capped)
    # random checks of 25 .. 127 rows: t, the keys in insertion order and the errors as the oracle enumerates them; one-word keys up
    # to 63 rows, two-word keys beyond (exact where the reference's int64 keys wrap); several table sizes on the way up
    r, n, cap = case
    rng = np.random.default_rng(r * 1000 + n)
    h = rng.integers(0, 2, (r, n))
    t, table = css_code.syndrome_table(h, max_weight=cap)
    want_t, entries = assert_table_is_the_oracles(table, t, h, cap)
    assert entries >= 1 + (n if want_t >= 1 else 0)


def test_syndrome_table_hashed_on_a_code_with_known_distance():
    # a check whose columns are all distinct and non-zero but with one pair of equal columns planted: weight 1 collides, t = 0;
    # and the [63, 57] Hamming check padded with 20 random rows of its row space (r = 26 > 24): distance 3, t = 1
    rng = np.random.default_rng(5)
    cols = np.arange(1, 64)
    ham = np.array([(cols >> b) & 1 for b in range(6)])
    mix = rng.integers(0, 2, (20, 6))
    h = np.vstack([ham, mix.dot(ham) % 2])
    t, table = css_code.syndrome_table(h)
    assert t == 1 and len(table) == 64
    assert_table_is_the_oracles(table, t, h, None)
    h2 = rng.integers(0, 2, (30, 50))
    h2[:, 17] = h2[:, 3]
    t, table = css_code.syndrome_table(h2)
    assert t == 0 and list(table.keys()) == [0]


@pytest.mark.parametrize("n,r1", [(51, 25), (55, 27), (63, 31), (63, 20), (80, 39)])
def test_css_code_constructor_on_mid_size_dual_pairs(n, r1):
    # CSSCode(H1, H2) for k = 1 pairs whose checks are beyond the dense tables: both tables and t as the oracle makes them from the
    # standard forms the constructor left
    rng = np.random.default_rng(n * 100 + r1)
    h1, h2 = dual_pair(rng, n, r1)
    code = CSSCode(h1, h2)
    assert code.k == 1 and code.r_1 + code.r_2 == n - 1
    # (the constructor keeps min(t_1, t_2) only: a table's own t is the weight of its heaviest entry)
    heaviest = lambda table: max(int(np.sum(v)) for v in table.values())
    t1, _ = assert_table_is_the_oracles(code._c1_syndromes, heaviest(code._c1_syndromes), code.parity_check_c1, None)
    t2, _ = assert_table_is_the_oracles(code._c2_syndromes, heaviest(code._c2_syndromes), code.parity_check_c2, None)
    assert code.t == min(t1, t2)


def oracle_tally(code, seed, first, count, p, cap):
    t1 = c_oracle.syndrome_table(c_oracle.pack_rows(code.parity_check_c1), code.r_1, code.n, cap)
    t2 = c_oracle.syndrome_table(c_oracle.pack_rows(code.parity_check_c2), code.r_2, code.n, cap)
    return c_oracle.mc_decode_wide(c_oracle.pack_rows(code.parity_check_c1), code.r_1, c_oracle.pack_rows(code.parity_check_c2), code.r_2,
                                   code.n, t1[1], t1[2], t2[1], t2[2], c_oracle.pack_rows(code.x_operator_matrix())[0],
                                   c_oracle.pack_rows(code.z_operator_matrix())[0], seed, first, count, *p)


@pytest.mark.parametrize("n,r1,cap,p", [(47, 23, None, (0.01, 0.005, 0.01)), (55, 27, None, (0.02, 0.01, 0.01)),
                                        (63, 31, None, (0.004, 0.004, 0.004)), (63, 31, None, (0.08, 0.08, 0.09)),
                                        (100, 49, 3, (0.01, 0.0, 0.02)), (127, 63, 2, (0.003, 0.003, 0.003)),
                                        (128, 64, 2, (0.002, 0.001, 0.004)), (70, 5, 3, (0.01, 0.01, 0.01))])
def test_decode_and_tally_through_hashed_tables(n, r1, cap, p):
    # gf2_mc_decode_hashed on k = 1 dual pairs of 47 .. 128 qubits (one- and two-word errors, one- and two-word keys): the five
    # counts of 300 000 samples against the oracle's tally with the oracle's own tables
    rng = np.random.default_rng(n + r1)
    h1, h2 = dual_pair(rng, n, r1)
    code = CSSCode(h1, h2, max_table_weight=cap)
    count, first = 300000, 12345
    got = code.logical_error_rates(count, *p, seed=99, first_sample=first)
    want = oracle_tally(code, 99, first, count, p, cap)
    assert [got[f] for f in montecarlo.DECODE_FIELDS] == [int(v) for v in want]
    assert got['samples'] == count


def test_decode_small_codes_through_both_kernels():
    # Steane and Reed-Muller [[15,1,3]]: the dense-table kernel (gf2_mc_decode) and the hash-table kernel give the same counts,
    # the oracle's
    steane = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
    cols = np.arange(1, 16)
    rm_h1 = np.array([(cols >> b) & 1 for b in range(4)])
    rm_h2 = np.vstack([rm_h1] + [rm_h1[a] & rm_h1[b] for a in range(4) for b in range(a + 1, 4)])
    for code in (CSSCode(steane, steane), CSSCode(rm_h1, rm_h2)):
        p = (0.03, 0.02, 0.04)
        dense = montecarlo.decode_local(code, 10**6, *p, seed=3, first_sample=7)
        hashed = montecarlo.decode_local(code, 10**6, *p, seed=3, first_sample=7, hashed=True)
        assert dense == hashed
        want = oracle_tally(code, 3, 7, 10**6, p, None)
        assert [hashed[f] for f in montecarlo.DECODE_FIELDS] == [int(v) for v in want]


def test_hashed_table_refuses_a_search_without_end():
    # no collision in sight and no cap: the search stops with an error once a class passes 2^28 errors instead of filling the
    # device (the reference would not return either)
    h = np.hstack([np.identity(60, dtype=int), np.zeros((60, 0), dtype=int)])
    with pytest.raises(ValueError, match="max_weight") as err:
        css_code.syndrome_table(h)
    cause = err.value.__cause__                                  # the device search's own refusal, re-raised for the caller
    assert isinstance(cause, _native.GF2Error) and cause.code == _native.GF2_E_NOMEM and "max_weight" in cause.message


def test_syndrome_table_without_a_bound_on_many_checks_says_how_to_bound_it():
    # a random check with 60 rows finds no two errors of one syndrome among its first 2^28 (the first collision of r
    # checks is expected after some 2^((r+1)/2) errors); the device search then stops with GF2_E_NOMEM, and css_code.syndrome_table
    # turns that into a ValueError that names max_weight -- the reference's loop (css_code.py:722-733) would run for days.  With a
    # bound the same check gives its table.
    rng = np.random.default_rng(60)
    h = rng.integers(0, 2, (60, 70))
    with pytest.raises(ValueError, match="max_weight"):
        css_code.syndrome_table(h)
    t, table = css_code.syndrome_table(h, max_weight=2)
    assert t == 2 and len(table) == 1 + 70 + 70 * 69 // 2
