import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from quantum_css_codes_amd import _native
ctx = _native.default_context()
n = 4096; count = 1 << 21
ex = ctx.alloc(count * 512); ez = ctx.alloc(count * 512)
p = 0.01 / 3
lib = _native.lib()
for rep in range(3):
    ctx.timer_start()
    _native.check(lib.gf2_sample_errors_dev(ctx.handle, n, 1, 0, count, p, p, p, ex.ptr, ez.ptr, 64, 0))
    ms = ctx.timer_stop()
print("sampler 2^21 samples: %.3f ms = %.3f ms per 2^20 (nostore=%s)" % (ms, ms / 2, os.environ.get("GF2_DBG_NOSTORE")))
