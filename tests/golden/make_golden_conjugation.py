#!/usr/bin/env python3
"""
Generates tests/golden/conjugation_golden.npz by running the REFERENCE's own NumPy-only conjugation rules
(css_code.py:757-781: conjugate_h_with_check_mat, conjugate_cnot_with_check_mat) on seeded stabiliser matrices and
gate sequences.  Run in the build container only; the fixtures (inputs, gate lists, the reference's outputs) travel.

    python tests/golden/make_golden_conjugation.py

css_code.transform_stabilisers itself (css_code.py:737-755) dispatches on pyquil Gate objects and is not called; the
import uses the same inert pyquil placeholders as make_golden.py.  Gates are rows (kind, a, b): kind 0 = H on qubit a,
kind 1 = CNOT control a target b.
"""
import os

import numpy as np

from make_golden import import_reference

OUT = os.path.dirname(os.path.abspath(__file__))


def apply(cc, mat, gates):
    """Returns the index of the first gate the reference refuses (NotImplementedError), or -1."""
    for idx, (kind, a, b) in enumerate(gates):
        try:
            if kind == 0:
                cc.conjugate_h_with_check_mat(mat, int(a))
            else:
                cc.conjugate_cnot_with_check_mat(mat, int(a), int(b))
        except NotImplementedError:
            return idx
    return -1


def main():
    _, cc, _ = import_reference()
    rng = np.random.default_rng(737)
    g = {}
    cases = [(1, 1, 5), (3, 2, 12), (7, 7, 60), (10, 15, 200), (40, 33, 300), (70, 64, 400), (130, 100, 500),
             (64, 129, 300)]
    for idx, (k, n, ngates) in enumerate(cases):
        # CSS-type start: every row is X-only or Z-only
        mat = np.zeros((k, 2 * n), dtype=np.int64)
        for i in range(k):
            half = rng.integers(0, 2)
            mat[i, half * n:(half + 1) * n] = rng.integers(0, 2, n)
        gates = np.zeros((ngates, 3), dtype=np.int32)
        gates[:, 0] = rng.random(ngates) < 0.7 if n > 1 else 0
        gates[:, 1] = rng.integers(0, n, ngates)
        gates[:, 2] = (gates[:, 1] + rng.integers(1, max(2, n), ngates)) % n if n > 1 else 0
        gates[gates[:, 0] == 0, 2] = 0
        work = np.array(mat)
        stop = apply(cc, work, gates)
        g["conj_in_%d" % idx] = mat
        g["conj_gates_%d" % idx] = gates
        g["conj_stop_%d" % idx] = np.array(stop)
        if stop < 0:
            g["conj_out_%d" % idx] = work
        else:
            # the prefix that the reference accepts, replayed on a fresh copy
            work = np.array(mat)
            assert apply(cc, work, gates[:stop]) == -1
            g["conj_out_%d" % idx] = work
    # encoder-shaped sequences that the reference accepts to the end: a layer of H on distinct qubits, then CNOTs only
    for idx, (k, n, ncnot) in enumerate([(6, 7, 40), (33, 70, 600), (100, 128, 1500), (65, 200, 900)]):
        mat = np.zeros((k, 2 * n), dtype=np.int64)
        for i in range(k):
            half = rng.integers(0, 2)
            mat[i, half * n:(half + 1) * n] = rng.integers(0, 2, n)
        hq = rng.permutation(n)[:n // 2]
        gates = np.zeros((len(hq) + ncnot, 3), dtype=np.int32)
        gates[:len(hq), 1] = hq
        gates[len(hq):, 0] = 1
        gates[len(hq):, 1] = rng.integers(0, n, ncnot)
        gates[len(hq):, 2] = (gates[len(hq):, 1] + rng.integers(1, n, ncnot)) % n
        work = np.array(mat)
        assert apply(cc, work, gates) == -1
        g["enc_in_%d" % idx], g["enc_gates_%d" % idx], g["enc_out_%d" % idx] = mat, gates, work
    # single-gate truth tables on all 16 (x_c, x_t, z_c, z_t) patterns of one row pair
    rows = np.array([[(v >> 3) & 1, (v >> 2) & 1, (v >> 1) & 1, v & 1] for v in range(16)], dtype=np.int64)
    work = np.array(rows)
    cc.conjugate_cnot_with_check_mat(work, 0, 1)
    g["cnot_truth_in"], g["cnot_truth_out"] = rows, work
    ok = rows[~((rows[:, 0] == 1) & (rows[:, 2] == 1))]
    work = np.array(ok)
    cc.conjugate_h_with_check_mat(work, 0)
    g["h_truth_in"], g["h_truth_out"] = ok, work
    np.savez_compressed(os.path.join(OUT, "conjugation_golden.npz"), **g)
    print("wrote", len(g), "arrays;", "stops:", [int(g["conj_stop_%d" % i]) for i in range(len(cases))])


if __name__ == "__main__":
    main()
