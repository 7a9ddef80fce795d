"""
CPU-only checks of the boundary: libgf2hip.so loads, exports every symbol include/gf2hip.h declares,
the ctypes table matches the header, the host-side packing works, and compute calls fail loudly (no CPU
fallback) when no GPU is present.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from quantum_css_codes_amd import _native, bin_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "gf2hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"\b(?:int64_t|int|const char\*)\s+(gf2_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    return {name: [a.strip() for a in args.split(",")] if args.strip() != "void" else [] for name, args in decls}


def test_library_exports_every_declared_symbol():
    funcs = header_functions()
    assert len(funcs) >= 30
    handle = _native.lib()
    for name in funcs:
        assert hasattr(handle, name), "libgf2hip.so does not export %s" % name
    assert handle.gf2_version() >= 100


def test_ctypes_table_matches_header():
    funcs = header_functions()
    assert set(funcs) == set(_native.SIGNATURES), set(funcs) ^ set(_native.SIGNATURES)
    for name, args in funcs.items():
        assert len(args) == len(_native.SIGNATURES[name]), name


def test_flag_and_option_numbers_match_the_header():
    # the public header names the few routes and tunables a caller needs, csrc/gf2_tuning.h the ones of the tests and A/B scripts:
    # together they are what _native.py states, no number used twice
    public = open(os.path.join(ROOT, "include", "gf2hip.h")).read()
    assert len(re.findall(r"#define GF2_F_\w+", public)) == 3 and len(re.findall(r"#define GF2_OPT_\w+", public)) == 2
    text = public + open(os.path.join(ROOT, "quantum_css_codes_amd", "csrc", "gf2_tuning.h")).read()
    flags = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define GF2_F_(\w+)\s+\(1u << (\d+)\)", text)}
    assert len(flags) >= 16
    for name, bit in flags.items():
        assert getattr(_native, "F_" + name) == 1 << bit, name
    assert len({v for v in flags.values()}) == len(flags)                     # no bit used twice
    opts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define GF2_OPT_(\w+)\s+(\d+)", text)}
    count = opts.pop("COUNT")
    assert sorted(opts.values()) == list(range(count))
    for name, number in opts.items():
        assert getattr(_native, "OPT_" + name) == number, name


def test_host_packing_roundtrip():
    handle = _native.lib()
    rng = np.random.default_rng(1)
    for (m, n) in ((1, 1), (3, 7), (5, 64), (4, 65), (9, 200)):
        mat = rng.integers(-5, 9, (m, n)).astype(np.int64)            # non-binary: packing applies & 1
        want = _native.pack_rows(mat)
        ld = want.shape[1]
        got = np.zeros_like(want)
        assert handle.gf2_pack_rows_i64(mat.ctypes.data, m, n, n, got.ctypes.data, ld) == 0
        assert np.array_equal(got, want)
        back = np.zeros((m, n), dtype=np.int64)
        assert handle.gf2_unpack_rows_i64(got.ctypes.data, m, n, ld, back.ctypes.data, n) == 0
        assert np.array_equal(back, mat & 1)
        assert np.array_equal(_native.unpack_rows(want, n), mat & 1)
        m8 = (mat & 1).astype(np.uint8)
        got8 = np.zeros_like(want)
        assert handle.gf2_pack_rows_u8(m8.ctypes.data, m, n, n, got8.ctypes.data, ld) == 0
        assert np.array_equal(got8, want)
        back8 = np.zeros((m, n), dtype=np.uint8)
        assert handle.gf2_unpack_rows_u8(got8.ctypes.data, m, n, ld, back8.ctypes.data, n) == 0
        assert np.array_equal(back8, m8)
    assert handle.gf2_pack_rows_u8(None, 2, 2, 2, None, 1) == _native.GF2_E_ARG
    assert b"null" in handle.gf2_last_error()


def test_pack_layout_is_column_j_at_word_j_shift_6():
    mat = np.zeros((1, 130), dtype=int)
    mat[0, [0, 63, 64, 129]] = 1
    words = _native.pack_rows(mat)
    assert words.shape == (1, 3)
    assert int(words[0, 0]) == (1 << 63) | 1 and int(words[0, 1]) == 1 and int(words[0, 2]) == 2
    assert np.array_equal(_native.pack_rows(np.array([[True, False, True]])), [[5]])


def test_host_bit_vector_helpers(golden):
    # test/test_bin_matrix.py:22-31 and reference-generated values
    assert bin_matrix.vec_to_int(np.array([0, 1, 0, 1, 1])) == 11
    assert np.array_equal(bin_matrix.int_to_vec(11, 5), np.array([0, 1, 0, 1, 1]))
    with pytest.raises(ValueError, match="n is too small"):
        bin_matrix.int_to_vec(11, 3)
    pos = 0
    for length, want in zip(golden["v2i_lens"], golden["v2i_vals"]):
        vec = golden["v2i_bits"][pos:pos + length]
        pos += length
        assert bin_matrix.vec_to_int(vec) == int(want)
        assert np.array_equal(bin_matrix.int_to_vec(int(want), int(length)), vec)
    # exact beyond 63 bits (the reference wraps; documented divergence)
    assert bin_matrix.vec_to_int(np.ones(100, dtype=np.int64)) == (1 << 100) - 1
    assert np.array_equal(bin_matrix.int_to_vec((1 << 100) + 12345, 101), golden["i2v_big"])
    for (n, w) in ((4, 2), (7, 0), (7, 1), (7, 2), (7, 3), (5, 5), (3, 4)):
        items = list(bin_matrix.weight_w_vectors(n, w))
        want = golden["wwv_%d_%d" % (n, w)]
        assert len(items) == want.shape[0]
        if items:
            assert np.array_equal(np.array(items), want)
            sup = bin_matrix.weight_w_supports(n, w)
            assert np.array_equal(np.sort(np.nonzero(want)[1].reshape(sup.shape), axis=1), sup)
    first = next(bin_matrix.weight_w_vectors(4, 2))
    first[:] = 9
    assert np.array_equal(next(bin_matrix.weight_w_vectors(4, 2)), [1, 1, 0, 0])


@pytest.mark.skipif(_native.device_count() > 0, reason="a GPU is present")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(_native.GF2Error):
        bin_matrix.reduced_row_echelon_form(np.eye(3, dtype=int))
    out = ctypes.c_void_p()
    assert _native.lib().gf2_ctx_create(0, ctypes.byref(out)) == _native.GF2_E_HIP
    assert out.value is None


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "quantum_css_codes_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", text, flags=re.S).replace("# ", ""), name
