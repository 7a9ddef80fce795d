#!/usr/bin/env python3
"""
Generates tests/golden/config4_golden.npz: the REFERENCE's own CSSCode constructor (css_code.py:32-75) run on the
benchmark's code -- configs[3] / configs[4] of BASELINE.json, SURVEY.md 8d config 4:

    H1 = default_rng(4096).integers(0, 2, (2048, 4096))          rank 2048
    H2 = first 2047 rows of nullspace(H1)                          => k = 1

Run in the build container only (the reference never travels; this file of digests does):

    python tests/golden/make_golden_config4.py          # about a minute, single thread

css_code.py is imported exactly as tests/golden/make_golden.py does it (pyquil is absent offline, the module names are
pre-seeded with empty placeholders; only NumPy code runs).  Two things of the imported module are wrapped, nothing of
its arithmetic is replaced:
  * syndrome_table (css_code.py:715-735) is exponential and CSSCode.__init__ calls it unconditionally (:69-70): at
    n = 4096 it cannot finish, so for the duration of the constructor call it returns (0, {}).
    The standard forms are complete before it is called (:55-68).
  * normalize_parity_check is wrapped to record the swap lists it returns (the constructor does not keep them).
nullspace is build-defined; H2 is read off the reference's own reduced_row_echelon_form(H1) by that
definition here, and its digest is stored so that the tests can tell that the product derives the same input.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, pack_rows, sha          # noqa: E402


def main():
    bm, cc, errs = import_reference()
    g = {}
    r1, r2, n = 2048, 2047, 4096
    h1 = np.random.default_rng(4096).integers(0, 2, (r1, n)).astype(np.int64)
    red = bm.reduced_row_echelon_form(h1)
    rank = int(np.count_nonzero(red.any(axis=1)))
    assert rank == r1
    # [build-defined x1] canonical nullspace basis from the RREF: row t has a 1 at free column F[t] and R[i, F[t]] at
    # pivot column P[i]
    piv = np.array([int(np.flatnonzero(red[i])[0]) for i in range(rank)])
    free = np.setdiff1d(np.arange(n), piv)
    basis = np.zeros((n - rank, n), dtype=np.int64)
    basis[np.arange(n - rank), free] = 1
    basis[:, piv] = red[:rank][:, free].T
    h2 = basis[:r2].copy()
    g["h1_in_sha"] = np.array(sha(pack_rows(h1)))
    g["h2_in_sha"] = np.array(sha(pack_rows(h2)))

    swap_log = []
    real_normalize, real_table = cc.normalize_parity_check, cc.syndrome_table

    def recording_normalize(h, offset):
        out, swaps = real_normalize(h, offset)
        swap_log.append((offset, list(swaps)))
        return out, swaps

    cc.normalize_parity_check = recording_normalize
    cc.syndrome_table = lambda parity_check: (0, {})
    try:
        code = cc.CSSCode(h1, h2)                               # css_code.py:32-75, the reference's own statements
    finally:
        cc.normalize_parity_check, cc.syndrome_table = real_normalize, real_table

    assert [off for off, _ in swap_log] == [0, r1]
    g["nk_r1_r2"] = np.array([code.n, code.k, code.r_1, code.r_2])
    g["c1_sha"] = np.array(sha(pack_rows(code.parity_check_c1)))
    g["c2_sha"] = np.array(sha(pack_rows(code.parity_check_c2)))
    g["swaps_c1"] = np.array(swap_log[0][1], dtype=np.int64).reshape(len(swap_log[0][1]), 2)
    g["swaps_c2"] = np.array(swap_log[1][1], dtype=np.int64).reshape(len(swap_log[1][1]), 2)
    g["gates"] = np.array(sorted(code._transversal_gates))
    g["zop"] = code.z_operator_matrix()                         # css_code.py:124-136
    g["xop"] = code.x_operator_matrix()                         # css_code.py:149-161
    # the identity blocks the standard forms must show (css_code.py:51-54)
    assert np.array_equal(code.parity_check_c1[:, :r1], np.identity(r1, dtype=int))
    assert np.array_equal(code.parity_check_c2[:, r1:r1 + r2], np.identity(r2, dtype=int))
    # a few syndrome products of the standard forms (css_code.py:728), X errors against c2 and Z errors against c1
    e = np.random.default_rng(78).integers(0, 2, (16, n)).astype(np.int64)
    g["syn_c1_sha"] = np.array(sha(pack_rows(np.array([np.mod(np.matmul(code.parity_check_c1, e[i]), 2) for i in range(16)]))))
    g["syn_c2_sha"] = np.array(sha(pack_rows(np.array([np.mod(np.matmul(code.parity_check_c2, e[i]), 2) for i in range(16)]))))
    # What the reference's syndrome_table (css_code.py:715-735) itself returns on these two checks: its keys come from
    # bin_matrix.vec_to_int (bin_matrix.py:40-43), whose running value is a NumPy int64 from the first addition on, so a key of
    # 2048 (2047) bits keeps its low 64 bits only.  The first weight-1 error, e_0, has the syndrome of H's column 0 -- for
    # parity_check_c1 = [I | A] the unit vector of row 0, key 2^2047 -> 0 after the wrap, the key of the zero error: the search ends
    # in class 1 with t = 0 and the one-entry table {0: 0}.  (Exact keys find no collision at weight 1: every column of a
    # standard form is non-zero and no two are equal for a code of distance >= 3; the drop-in answers t >= 1, DESIGN.md
    # section 5.)  Stored: t and the table's keys as the reference made them.
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                          # (NumPy's overflow warnings of exactly that wrap)
        for name, h in (("c1", code.parity_check_c1), ("c2", code.parity_check_c2)):
            t, table = real_table(h)
            g["ref_table_%s_t" % name] = np.array(t)
            g["ref_table_%s_keys" % name] = np.array([int(k) for k in table.keys()], dtype=np.int64)
            g["ref_table_%s_weights" % name] = np.array([int(np.sum(v)) for v in table.values()], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "config4_golden.npz"), **g)
    print("wrote", len(g), "arrays;", "swaps", len(swap_log[0][1]), len(swap_log[1][1]), "gates", g["gates"])


if __name__ == "__main__":
    main()
