#!/usr/bin/env python3
"""
Generates tests/golden/*.npz by running the REFERENCE ITSELF (jimpo/quantum-css-codes mounted at
/root/reference) on fixed inputs.  Run in the build container only -- the reference never travels
to the GPU box, these fixtures (plain data: inputs and the reference's outputs) do.

    python tests/golden/make_golden.py

bin_matrix.py imports as is (NumPy only).  css_code.py imports pyquil at css_code.py:7-11, which is
not installed here and cannot be fetched offline; its numeric functions do not touch pyquil, so
the pyquil module names are pre-seeded with empty placeholder modules and only
the NumPy-only functions are called: CSSCode.__init__, normalize_parity_check, swap_columns,
syndrome_table, codes_equal, is_doubly_even, x/z_operator_matrix.  Pauli-term outputs
(stabilisers(), x/y/z_operators()) are NOT taken from this import; they are pinned from the
reference's own expectations in test/test_css_code.py:32-59.
"""
import hashlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ("pyquil", "pyquil.gates", "pyquil.paulis", "pyquil.quil", "pyquil.quilatom",
                 "pyquil.quilbase"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for name, attrs in (("pyquil", ("Program",)),       # quil_classical.py:5
                        ("pyquil.paulis", ("PauliTerm", "ID", "sX", "sY", "sZ")),
                        ("pyquil.quil", ("Program",)),
                        ("pyquil.quilatom", ("MemoryReference", "Qubit", "QubitPlaceholder")),
                        ("pyquil.quilbase", ("Gate",))):
        for attr in attrs:
            setattr(sys.modules[name], attr, type(attr, (), {}))
    sys.modules["pyquil"].gates = sys.modules["pyquil.gates"]
    sys.path.insert(0, REF)
    import bin_matrix
    import css_code
    import errors
    return bin_matrix, css_code, errors


def pack_rows(mat):
    """rows -> little-endian packed uint64 words (column j at word j>>6, bit j&63)."""
    mat = np.asarray(mat) & 1
    m, n = mat.shape
    ld = max(1, (n + 63) // 64)
    padded = np.zeros((m, ld * 64), dtype=np.uint8)
    padded[:, :n] = mat
    return np.packbits(padded, axis=1, bitorder="little").view("<u8").reshape(m, ld)


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def table_arrays(table):
    keys = np.array(list(table.keys()), dtype=np.int64)
    vals = np.array([table[k] for k in table.keys()], dtype=np.int64).reshape(len(keys), -1)
    return keys, vals


def steane_h():
    return np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])


def rm15():
    cols = np.arange(1, 16)
    h1 = np.array([(cols >> b) & 1 for b in range(4)])
    pairs = [h1[a] & h1[b] for a in range(4) for b in range(a + 1, 4)]
    h2 = np.vstack([h1] + pairs)
    return h1, h2


def main():
    bm, cc, errs = import_reference()
    g = {}

    # ---- bin_matrix: RREF on seeded shapes -----------------------------------
    rng = np.random.default_rng(20261004)
    shapes = [(1, 1), (3, 7), (17, 65), (64, 64), (63, 129), (100, 50), (128, 256), (256, 512),
              (5, 64), (65, 63), (2, 200)]
    names = []
    for idx, (m, n) in enumerate(shapes):
        a = rng.integers(0, 2, (m, n)).astype(np.int64)
        g["rref_in_%d" % idx] = a
        g["rref_out_%d" % idx] = bm.reduced_row_echelon_form(a)
        names.append(idx)
    # rank deficient, zero, zero-row, non-binary entries, small dtypes
    a = rng.integers(0, 2, (20, 40)).astype(np.int64)
    a[10:] = a[:10][::-1] ^ a[:10]
    g["rref_in_def"], g["rref_out_def"] = a, bm.reduced_row_echelon_form(a)
    a = np.zeros((6, 70), dtype=np.int64)
    g["rref_in_zero"], g["rref_out_zero"] = a, bm.reduced_row_echelon_form(a)
    a = np.zeros((0, 9), dtype=np.int64)
    g["rref_in_norows"], g["rref_out_norows"] = a, bm.reduced_row_echelon_form(a)
    a = rng.integers(0, 7, (9, 33)).astype(np.int64)
    g["rref_in_nonbin"], g["rref_out_nonbin"] = a, bm.reduced_row_echelon_form(a)
    a = rng.integers(0, 2, (12, 31)).astype(np.uint8)
    g["rref_in_u8"], g["rref_out_u8"] = a, bm.reduced_row_echelon_form(a)
    a = rng.integers(0, 2, (12, 31)).astype(np.int8)
    g["rref_in_i8"], g["rref_out_i8"] = a, bm.reduced_row_echelon_form(a)
    g["rref_shape_ids"] = np.array(names)

    # ---- vec_to_int / int_to_vec / weight_w_vectors -----------------------------------------
    vecs = [rng.integers(0, 2, L).astype(np.int64) for L in (1, 2, 5, 17, 31, 32, 33, 62, 63)]
    g["v2i_lens"] = np.array([v.size for v in vecs])
    g["v2i_bits"] = np.concatenate(vecs)
    g["v2i_vals"] = np.array([int(bm.vec_to_int(v)) for v in vecs], dtype=np.int64)
    g["v2i_allones64"] = np.array([int(bm.vec_to_int(np.ones(L, dtype=np.int64)))
                                   for L in (63, 64, 65, 100)], dtype=np.int64)
    big = (1 << 100) + 12345
    g["i2v_big"] = bm.int_to_vec(big, 101)
    for (n, w) in ((4, 2), (7, 0), (7, 1), (7, 2), (7, 3), (5, 5), (3, 4)):
        items = list(bm.weight_w_vectors(n, w))
        g["wwv_%d_%d" % (n, w)] = (np.array(items, dtype=np.int64).reshape(len(items), n))

    # ---- normalize_parity_check -------------------------------------------------------------
    def run_norm(tag, h, offset):
        work = np.array(h, dtype=np.int64)
        out, swaps = cc.normalize_parity_check(work, offset)
        g["norm_in_" + tag] = np.array(h, dtype=np.int64)
        g["norm_off_" + tag] = np.array(offset)
        g["norm_out_" + tag] = out
        g["norm_mut_" + tag] = work          # the argument is mutated in place (raw, un-reduced)
        g["norm_swaps_" + tag] = np.array(swaps, dtype=np.int64).reshape(len(swaps), 2)

    run_norm("steane0", steane_h(), 0)
    run_norm("steane3", steane_h(), 3)
    h1, h2 = rm15()
    run_norm("rm15_h1", h1, 0)
    run_norm("rm15_h2", h2, 4)
    tags = ["steane0", "steane3", "rm15_h1", "rm15_h2"]
    for idx, (r, n, off) in enumerate([(8, 20, 0), (8, 20, 12), (30, 70, 5), (64, 128, 64),
                                       (33, 130, 3), (40, 41, 0), (65, 200, 66)]):
        for attempt in range(200):
            h = rng.integers(0, 2, (r, n)).astype(np.int64)
            # sprinkle zero columns so that column swaps actually happen
            n_zero = min(3, n - off - r)
            if n_zero:
                h[:, rng.integers(off, off + r, n_zero)] = 0
            try:
                cc.normalize_parity_check(np.array(h), off)
            except errs.InvalidCodeError:
                continue
            break
        else:
            raise RuntimeError("no independent draw")
        run_norm("rand%d" % idx, h, off)
        tags.append("rand%d" % idx)
    g["norm_tags"] = np.array(tags)
    dep = np.array([[1, 1, 0, 0, 1], [1, 1, 0, 0, 1], [0, 0, 1, 1, 0]], dtype=np.int64)
    try:
        cc.normalize_parity_check(np.array(dep), 0)
        raise RuntimeError("expected InvalidCodeError")
    except errs.InvalidCodeError:
        g["norm_dep_in"] = dep
    try:
        cc.normalize_parity_check(np.zeros((3, 5), dtype=np.int64), 3)
        raise RuntimeError("expected ValueError")
    except ValueError:
        pass

    # ---- CSSCode attribute dumps: Steane and RM[[15,1,3]] --------------------------------------
    def dump_code(tag, a, b):
        code = cc.CSSCode(np.array(a), np.array(b))
        g[tag + "_in1"], g[tag + "_in2"] = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
        g[tag + "_h1"], g[tag + "_h2"] = code.parity_check_c1, code.parity_check_c2
        g[tag + "_nktr"] = np.array([code.n, code.k, code.t, code.r_1, code.r_2])
        g[tag + "_gates"] = np.array(sorted(code._transversal_gates))
        g[tag + "_zop"], g[tag + "_xop"] = code.z_operator_matrix(), code.x_operator_matrix()
        for which, tab in (("c1", code._c1_syndromes), ("c2", code._c2_syndromes)):
            keys, vals = table_arrays(tab)
            g["%s_%s_keys" % (tag, which)], g["%s_%s_errs" % (tag, which)] = keys, vals
        for which, h in (("h1", code.parity_check_c1), ("h2", code.parity_check_c2)):
            t, tab = cc.syndrome_table(h)
            keys, vals = table_arrays(tab)
            g["%s_tab_%s_t" % (tag, which)] = np.array(t)
            g["%s_tab_%s_keys" % (tag, which)] = keys
            g["%s_tab_%s_errs" % (tag, which)] = vals

    dump_code("steane", steane_h(), steane_h())
    dump_code("rm15", *rm15())

    # InvalidCodeError for k != 1 raised last (css_code.py:74-75): [[4,2,2]] code
    try:
        cc.CSSCode(np.array([[1, 1, 1, 1]]), np.array([[1, 1, 1, 1]]))
        raise RuntimeError("expected InvalidCodeError")
    except errs.InvalidCodeError:
        pass

    # ---- codes_equal / is_doubly_even -------------------------------------------------------
    a = rng.integers(0, 2, (6, 14)).astype(np.int64)
    mix = rng.integers(0, 2, (6, 6)).astype(np.int64)
    while int(round(abs(np.linalg.det(mix)))) % 2 == 0:
        mix = rng.integers(0, 2, (6, 6)).astype(np.int64)
    b = np.mod(mix @ a, 2)
    c = np.array(a)
    c[0, 0] ^= 1
    g["ceq_a"], g["ceq_b"], g["ceq_c"] = a, b, c
    g["ceq_res"] = np.array([cc.codes_equal(a, b), cc.codes_equal(a, c), cc.codes_equal(a, a[:5])])
    de = rng.integers(0, 2, (10, 24)).astype(np.int64)
    g["de_in"] = de
    g["de_rows"] = np.array([cc.is_doubly_even(de[i:i + 1]) for i in range(10)])

    # ---- syndrome products np.mod(np.matmul(H, e), 2) (css_code.py:728) ------------------------
    for tag, h in (("steane", g["steane_h1"]), ("rm15", g["rm15_h2"]),
                   ("r64x128", rng.integers(0, 2, (64, 128)).astype(np.int64)),
                   ("r70x200", rng.integers(0, 2, (70, 200)).astype(np.int64))):
        n = h.shape[1]
        e = rng.integers(0, 2, (50, n)).astype(np.int64)
        g["syn_h_" + tag] = h
        g["syn_e_" + tag] = e
        g["syn_s_" + tag] = np.array([np.mod(np.matmul(h, e[i]), 2) for i in range(50)])

    # ---- large shape digests (config 4 of BASELINE.json): 512x1024 here, full size hashed ------
    a = np.random.default_rng(1024).integers(0, 2, (512, 1024)).astype(np.int64)
    red = bm.reduced_row_echelon_form(a)
    g["big512_rref_sha"] = np.array(sha(pack_rows(red)))
    g["big512_rank"] = np.array(int(np.count_nonzero(red.any(axis=1))))
    if os.environ.get("GOLDEN_FULL", "1") == "1":
        a = np.random.default_rng(4096).integers(0, 2, (2048, 4096)).astype(np.int64)
        red = bm.reduced_row_echelon_form(a)
        g["big4096_rref_sha"] = np.array(sha(pack_rows(red)))
        g["big4096_rank"] = np.array(int(np.count_nonzero(red.any(axis=1))))
        work = np.array(a)
        out, swaps = cc.normalize_parity_check(work, 0)
        g["big4096_norm_sha"] = np.array(sha(pack_rows(out)))
        g["big4096_norm_swaps"] = np.array(swaps, dtype=np.int64).reshape(len(swaps), 2)
        e = np.random.default_rng(77).integers(0, 2, (32, 4096)).astype(np.int64)
        s = np.array([np.mod(np.matmul(a, e[i]), 2) for i in range(32)])
        g["big4096_syn_sha"] = np.array(sha(pack_rows(s)))

    np.savez_compressed(os.path.join(OUT, "reference_golden.npz"), **g)
    print("wrote", len(g), "arrays")


if __name__ == "__main__":
    main()
