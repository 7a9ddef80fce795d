"""
The host-only translation unit of libgf2hip.so (csrc/gf2_host.cpp: error message, pack / unpack of dense arrays on up to 16
host threads) under ThreadSanitizer and AddressSanitizer + UBSan on the CPU box (sanitizers).  `make tsan`
/ `make asan` build that translation unit alone; a child interpreter loads it beside the sanitizer's runtime and runs the
packing round trips of tests/test_abi.py on arrays large enough for several threads, from two Python threads at once (the
message buffer is thread-local), and through the out-of-threads fallback (a test hook of the sanitizer builds makes the
k-th std::thread of a call fail).
"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quantum_css_codes_amd", "csrc")

CHILD = r"""
import ctypes, sys, threading
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
i64, p = ctypes.c_int64, ctypes.c_void_p
for name in ("gf2_pack_rows_u8", "gf2_pack_rows_i64", "gf2_unpack_rows_u8", "gf2_unpack_rows_i64"):
    getattr(lib, name).argtypes = [p, i64, i64, i64, p, i64]
for name in ("gf2_pack_rows_binary_u8", "gf2_pack_rows_binary_i64"):
    getattr(lib, name).argtypes = [p, i64, i64, i64, p, i64, ctypes.POINTER(ctypes.c_int)]
lib.gf2_last_error.restype = ctypes.c_char_p
lib.gf2_host_test_fail_after.argtypes = [ctypes.c_int]

def pack_numpy(mat):
    m, n = mat.shape
    ld = (n + 63) // 64
    bits = np.zeros((m, ld * 64), dtype=np.uint8)
    bits[:, :n] = mat & 1
    return np.packbits(bits, axis=1, bitorder="little").view("<u8").reshape(m, ld)

def round_trip(seed, shapes, errors):
    try:
        rng = np.random.default_rng(seed)
        for (m, n) in shapes:
            mat = rng.integers(-5, 9, (m, n)).astype(np.int64)          # non-binary entries: packing applies & 1
            want = pack_numpy(mat)
            ld = want.shape[1]
            got = np.zeros_like(want)
            assert lib.gf2_pack_rows_i64(mat.ctypes.data, m, n, n, got.ctypes.data, ld) == 0
            assert np.array_equal(got, want), ("pack i64", m, n)
            other = ctypes.c_int(0)
            assert lib.gf2_pack_rows_binary_i64(mat.ctypes.data, m, n, n, got.ctypes.data, ld, ctypes.byref(other)) == 0
            assert other.value == int(((mat != 0) & (mat != 1)).any()) and np.array_equal(got, want), ("binary test", m, n)
            back = np.zeros((m, n), dtype=np.int64)
            assert lib.gf2_unpack_rows_i64(got.ctypes.data, m, n, ld, back.ctypes.data, n) == 0
            assert np.array_equal(back, mat & 1), ("unpack i64", m, n)
            m8 = (mat & 1).astype(np.uint8)
            got8 = np.zeros_like(want)
            assert lib.gf2_pack_rows_binary_u8(m8.ctypes.data, m, n, n, got8.ctypes.data, ld, ctypes.byref(other)) == 0
            assert other.value == 0 and np.array_equal(got8, want)
            back8 = np.zeros((m, n), dtype=np.uint8)
            assert lib.gf2_unpack_rows_u8(got8.ctypes.data, m, n, ld, back8.ctypes.data, n) == 0
            assert np.array_equal(back8, m8)
        assert lib.gf2_pack_rows_u8(None, 2, 2, 2, None, 1) == -1 and b"null" in lib.gf2_last_error()
    except BaseException as err:        # noqa: BLE001 -- reported by the parent
        errors.append(repr(err))

small = [(1, 1), (3, 7), (5, 64), (4, 65), (9, 200)]                # tests/test_abi.py's shapes: one thread
large = [(2048, 4096), (1000, 3001), (4099, 700)]                    # 64 MiB / 24 MiB / 23 MiB of int64: several threads
errors = []
round_trip(1, small + large, errors)
workers = [threading.Thread(target=round_trip, args=(s, large, errors)) for s in (2, 3)]   # two callers at once
for w in workers: w.start()
for w in workers: w.join()
for k in (0, 1, 3):                                                   # the k-th thread of every call cannot be started
    lib.gf2_host_test_fail_after(k)
    round_trip(10 + k, large, errors)
lib.gf2_host_test_fail_after(-1)
if errors:
    print("\n".join(errors)); sys.exit(1)
print("host packing ok")
"""


def runtime_of(kind):
    out = subprocess.run(["g++", "-print-file-name=lib%s.so" % kind], capture_output=True, text=True).stdout.strip()
    if out and os.path.isabs(out) and os.path.exists(out):
        return os.path.realpath(out)
    hits = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/lib%s.so" % kind))
    return os.path.realpath(hits[-1]) if hits else None


@pytest.mark.parametrize("kind,target,marker", [("tsan", "tsan", "ThreadSanitizer"), ("asan", "asan", "AddressSanitizer")])
def test_host_packing_under_sanitizer(kind, target, marker, tmp_path):
    runtime = runtime_of(kind)
    if runtime is None:
        pytest.skip("lib%s is not installed" % kind)
    subprocess.run(["make", "-C", CSRC, target], check=True, capture_output=True)
    lib = os.path.join(CSRC, "build", "libgf2host_%s.so" % kind)
    # (the C++ runtime beside it: the interpreter is a C program, and the sanitizer's interceptor of __cxa_throw -- the fallback
    # test throws -- must find the real one when it is set up)
    stdcxx = subprocess.run(["g++", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    preload = runtime + (" " + os.path.realpath(stdcxx) if os.path.isabs(stdcxx) and os.path.exists(stdcxx) else "")
    env = dict(os.environ, LD_PRELOAD=preload, GF2_HOST_THREADS="8", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="exitcode=66 report_signal_unsafe=0",
               UBSAN_OPTIONS="halt_on_error=1 print_stacktrace=1")
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    run = subprocess.run([sys.executable, str(script), lib], env=env, capture_output=True, text=True, timeout=600)
    report = run.stdout[-2000:] + run.stderr[-4000:]
    assert run.returncode == 0, report
    assert "host packing ok" in run.stdout
    assert marker not in run.stderr and "runtime error" not in run.stderr, report
