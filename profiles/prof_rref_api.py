import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from quantum_css_codes_amd import _native, bin_matrix
rng = np.random.default_rng(4096)
a = rng.integers(0, 2, (2048, 4096))
ctx = _native.default_context()
for rep in range(4):
    t0 = time.perf_counter(); packed = _native.pack_rows(a); t1 = time.perf_counter()
    piv, rank = ctx.rref(packed, 2048, 4096); t2 = time.perf_counter()
    out = _native.unpack_rows(packed, 4096, dtype=a.dtype); t3 = time.perf_counter()
    print("pack %.2f ms  ctx.rref %.2f ms  unpack %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
for rep in range(3):
    t0 = time.perf_counter(); packed = _native.pack_rows(a); t1 = time.perf_counter()
    basis = ctx.nullspace(packed, 2048, 4096); t2 = time.perf_counter()
    out = _native.unpack_rows(basis, 4096, dtype='int'); t3 = time.perf_counter()
    print("nullspace: pack %.2f ms  ctx.nullspace %.2f ms  unpack %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
