#!/usr/bin/env python3
"""
bench.py -- syndromes/s of the n = 4096 CSS Monte-Carlo hot path on N x MI355X (BASELINE.json metric), with the
GF(2) RREF GB/s beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher, WORLD_SIZE unset: starts its N ranks itself, one process per GPU,
                                         before this process has touched the GPU, and exits with their worst code)

Workload (BASELINE.json configs[4]): the random dual code of config 4 --
H1 = default_rng(4096) 2048 x 4096, H2 = first 2047 rows of nullspace(H1), both put in standard form by
CSSCode exactly as css_code.py:51-61 does -- and depolarising errors (p = 0.01) from the counter-based
sampler, pre-materialised in HBM before the timed region (2^27 samples per GPU by default, --batch-log2: 128 GiB of
packed errors, 512 times the 256 MiB Infinity Cache, so every step streams from HBM).

One step = one pass of the hot path over that batch: s_z = H1 . e_z and s_x = H2 . e_x for every sample and
the two syndrome-weight histograms.  Two implementations, same results bit for bit:
  --algo sparse (default)  LDS-slab pipeline (gf2_slabs.hip): the samples' set bits are compacted into column
                           records, every CU keeps one 512-row slab of the transposed check in LDS and XORs the
                           listed columns, partial weights are combined into the histogram; no syndromes are
                           written.  Work ~ error weight (about 27 set bits per component at p = 0.01).  The two
                           components of a step go to two contexts = two HIP streams (--one-stream: one).
  --algo dense             Four-Russians table kernel on tiled errors, slab-major syndromes written, then the
                           histogram kernel.  Data-independent.
Per-GPU work is fixed as N grows ("weak"; --total-samples T: T samples in all, 1/N of them per rank, "strong" -- T = 10^8 is
configs[4] to the letter); ranks never exchange data on the path; the histograms are summed once with one all-reduce inside
the timed region -- torch.distributed's all_reduce (RCCL) by default; --allreduce gf2: gf2_hist_allreduce, libgf2hip's own RCCL
communicator, from the device buffer the kernels accumulate into, on the compute context's stream, set up under a time limit
(guarded_comm: a rank whose communicator does not come up leaves, the launcher starts fresh ranks with --allreduce torch).
value = K * (samples of all ranks per step) / max-over-ranks time.

The JSON line also carries
  roofline      the path against the HBM roofline: algorithmic bytes per launch (n/8 read [+ r/8
                written when syndromes are stored] per sample and component) / mean launch time, measured live
                with HIP events; traffic = PMC HBM bytes per launch from the committed rocprofv3 passes
                (profiles/traffic.json).  With two streams a launch is one step (both components; events on the
                main stream around the timed region, the side stream joined before the stop event); with one
                stream it is one library call, timed on the context's stream (secondary.one_stream).
  cpu_baseline  the reference's NumPy path (oracle/cpu_ref.py restatement of css_code.py:728) timed on this
                host, 1 core, on a bounded sample of the same workload.
  secondary     (rank 0, N = 1) the end-to-end Monte-Carlo (sampler included), the other syndrome kernels (column gather,
                dense table) on the same workload, configs[1] and configs[2] (Steane, Reed-Muller), and RREF GB/s
                (2 * m * ceil(n/64) * 8 bytes / time) for one and for 256 resident 2048 x 4096 matrices, for one
                32768 x 65536 matrix, and for 256 MiB of 64 x 512 and of 128 x 512 matrices.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_QUBITS, R1, R2 = 4096, 2048, 2047
P_TOTAL = 0.01
SEED = 0xC55C0DE
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_code():
    """Config 4 of BASELINE.json on the GPU: returns the code and packed standard-form H1, H2 (uint64 words)."""
    from quantum_css_codes_amd import _native, bin_matrix
    from quantum_css_codes_amd.css_code import CSSCode
    seed = 4096
    while True:
        h1 = np.random.default_rng(seed).integers(0, 2, (R1, N_QUBITS)).astype(np.uint8)
        if bin_matrix.rank(h1) == R1:
            break
        seed += 1
    h2 = bin_matrix.nullspace(h1)[:R2]
    code = CSSCode(h1, h2, max_table_weight=0)       # syndrome_table is exponential: capped
    assert code.k == 1
    c1, c2 = _native.pack_rows(code.parity_check_c1), _native.pack_rows(code.parity_check_c2)
    # the timed kernels' inputs are the reference's own: digests of what its constructor (css_code.py:32-75) made of the
    # same H1, H2 (tests/golden/make_golden_config4.py ran it; the file holds data only)
    with np.load(os.path.join(ROOT, "tests", "golden", "config4_golden.npz"), allow_pickle=False) as g:
        digest = lambda arr: hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
        assert digest(_native.pack_rows(h1)) == str(g["h1_in_sha"]) and digest(_native.pack_rows(h2)) == str(g["h2_in_sha"])
        assert digest(c1) == str(g["c1_sha"]) and digest(c2) == str(g["c2_sha"]), "standard forms differ from the reference's"
        assert np.array_equal(code.z_operator_matrix(), g["zop"]) and np.array_equal(code.x_operator_matrix(), g["xop"])
    return code, c1, c2


def cpu_baseline(code, c1p, c2p, seconds_numpy=12.0, sample=256, seconds_c=8.0):
    """Two CPU legs on this host, both on a bounded sample of the same workload:
    value / cores / kind / sample: the reference-style path, np.mod(np.matmul(H, e), 2) per error vector on dense int64 arrays
        (css_code.py:728), both Pauli components, one core (NumPy's integer matmul is not threaded) -- 256 vectors;
    packed_c: the same products on packed 64-bit words (oracle/gf2_oracle.c: AND, XOR, popcount parity), OpenMP over the samples
        on every core the host gives -- the stronger baseline SURVEY.md 8d offers."""
    from oracle import cpu_ref, c_oracle
    h1 = np.array(code.parity_check_c1, dtype='int')
    h2 = np.array(code.parity_check_c2, dtype='int')
    exp, ezp = c_oracle.sample_errors(N_QUBITS, SEED, 0, sample, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3)
    ex, ez = c_oracle.unpack_rows(exp, N_QUBITS), c_oracle.unpack_rows(ezp, N_QUBITS)
    done, t0 = 0, time.perf_counter()
    while done < sample:
        cpu_ref.syndrome_product(h1, ez[done])
        cpu_ref.syndrome_product(h2, ex[done])
        done += 1
        if time.perf_counter() - t0 > seconds_numpy and done >= 16:
            break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "syndromes/s", "cores": 1, "kind": "port",
           "sample": "%d errors of the same n=4096 code, np.mod(np.matmul(H,e),2) for H1.e_z and H2.e_x "
                     "(oracle/cpu_ref.py restating css_code.py:728), %.1f s, host has %d cores"
                     % (done, dt, os.cpu_count())}
    # packed words, all cores
    count = 1 << 17
    bx, bz = c_oracle.sample_errors(N_QUBITS, SEED, 0, count, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3)
    c_oracle.syndrome_batch(c1p, R1, N_QUBITS, bz[:4096], 4096)           # threads started, pages touched
    reps, t0 = 0, time.perf_counter()
    while True:
        c_oracle.syndrome_batch(c1p, R1, N_QUBITS, bz, count)
        c_oracle.syndrome_batch(c2p, R2, N_QUBITS, bx, count)
        reps += 1
        if time.perf_counter() - t0 > seconds_c or reps >= 64:
            break
    dt = time.perf_counter() - t0
    out["packed_c"] = {"value": reps * count / dt, "unit": "syndromes/s", "cores": c_oracle.max_threads(), "kind": "port",
                       "sample": "%d x %d errors of the same code, packed-word AND / XOR / popcount products for H1.e_z and H2.e_x "
                                 "(oracle/gf2_oracle.c orc_syndrome_batch, OpenMP over the samples), %.1f s" % (reps, count, dt)}
    return out


class Path(object):
    """Resident buffers and the step function of one implementation of the hot path."""

    def __init__(self, ctx, algo, chk1, chk2, batch, first, ctx2=None):
        from quantum_css_codes_amd import _native
        self.ctx, self.algo, self.batch = ctx, algo, batch
        # second context = second HIP stream (own workspace): the sparse path issues H1.e_z on ctx and H2.e_x on ctx2, so
        # that the two slab pipelines, which stress HBM and the SIMDs at different moments, overlap; None: one stream
        self.ctx2 = ctx2 if (ctx2 is not None and algo == "sparse") else None
        self.ls1, self.ls2 = _native.words_for(R1), _native.words_for(R2)
        # one buffer for both histograms, so that the ranks' sum is one all-reduce of (R1 + 1) + (R2 + 1) words
        self.hist = ctx.alloc((R1 + 1 + R2 + 1) * 8)
        self.hz, self.hx = self.hist.view(0, (R1 + 1) * 8), self.hist.view((R1 + 1) * 8, (R2 + 1) * 8)
        p = P_TOTAL / 3
        if algo == "sparse":
            # sample-major packed errors resident in HBM; histogram-only output (no syndromes written)
            lde = _native.words_for(N_QUBITS)
            self.ex, self.ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
            for done in range(0, batch, 1 << 21):                        # the sampler writes 2^21 samples per call
                _native.check(_native.lib().gf2_sample_errors_dev(
                    ctx.handle, N_QUBITS, SEED, first + done, min(1 << 21, batch - done), p, p, p,
                    self.ex.ptr + done * lde * 8, self.ez.ptr + done * lde * 8, lde, _native.LAYOUT_SAMPLE_MAJOR))
            self.kernel = "slab_pipeline"                                # compact + gather + combine, gf2_slabs.hip
            self.alg_bytes_per_sample = N_QUBITS / 8.0                   # SURVEY.md 8d read-only variant, per component
            side = self.ctx2 if self.ctx2 is not None else ctx

            def step():
                ctx.syndrome_sparse_dev(chk1, self.ez, batch, lde, None, 0, self.hz, R1 + 1)
                side.syndrome_sparse_dev(chk2, self.ex, batch, lde, None, 0, self.hx, R2 + 1)

            def prefix_hist(count):
                a, b = ctx.alloc((R1 + 1) * 8).zero(), ctx.alloc((R2 + 1) * 8).zero()
                ctx.syndrome_sparse_dev(chk1, self.ez, count, lde, None, 0, a, R1 + 1)
                ctx.syndrome_sparse_dev(chk2, self.ex, count, lde, None, 0, b, R2 + 1)
                out = a.download((R1 + 1,), np.uint64), b.download((R2 + 1,), np.uint64)
                a.free(), b.free()
                return out
            self.prefix_hist = prefix_hist

            def prefix(count):
                a, b = ctx.alloc(count * self.ls1 * 8).zero(), ctx.alloc(count * self.ls2 * 8).zero()
                ctx.syndrome_sparse_dev(chk1, self.ez, count, lde, a, self.ls1)
                ctx.syndrome_sparse_dev(chk2, self.ex, count, lde, b, self.ls2)
                out = a.download((count, self.ls1), "<u8"), b.download((count, self.ls2), "<u8")
                a.free(), b.free()
                return out
        else:
            lde = _native.tiled_ld(N_QUBITS)
            tiled = _native.LAYOUT_TILED      # device-native error layout (include/gf2hip.h), written by the sampler
            words = _native.tiled_words(N_QUBITS, batch)
            self.ex, self.ez = ctx.alloc(words * 8), ctx.alloc(words * 8)
            self.s1, self.s2 = ctx.alloc(batch * self.ls1 * 8), ctx.alloc(batch * self.ls2 * 8)
            ctx.sample_errors_dev(N_QUBITS, SEED, first, batch, p, p, p, self.ex, self.ez, lde, tiled)
            self.kernel = "syndrome_tiled_kernel"
            self.alg_bytes_per_sample = N_QUBITS / 8.0 + (R1 + R2) / 2.0 / 8.0   # packed error read + syndrome written

            def step():
                # tiled errors in, slab-major syndromes out (word s of sample b at s*batch + b)
                ctx.syndrome_dev(chk1, self.ez, batch, lde, self.s1, batch, tiled)
                ctx.syndrome_dev(chk2, self.ex, batch, lde, self.s2, batch, tiled)
                ctx.histogram_dev(self.s1, batch, batch, R1, _native.HIST_WEIGHT, self.hz, R1 + 1, tiled)
                ctx.histogram_dev(self.s2, batch, batch, R2, _native.HIST_WEIGHT, self.hx, R2 + 1, tiled)

            def prefix(count):
                return (np.ascontiguousarray(self.s1.download((self.ls1, batch), "<u8")[:, :count].T),
                        np.ascontiguousarray(self.s2.download((self.ls2, batch), "<u8")[:, :count].T))
        self.step, self.prefix = step, prefix
        ctx.sync()

    def sync(self):
        self.ctx.sync()
        if self.ctx2 is not None:
            self.ctx2.sync()

    def contexts(self):
        return [self.ctx] + ([self.ctx2] if self.ctx2 is not None else [])

    def check_against_oracle(self, h1, h2, first):
        """A prefix of the resident batch through this path against the C oracle."""
        from oracle import c_oracle
        p = P_TOTAL / 3
        self.hz.zero(), self.hx.zero()
        self.ctx.sync()
        self.step()
        self.sync()
        want_z, want_x = c_oracle.mc(h1, R1, h2, R2, N_QUBITS, SEED, first, 512, p, p, p, 1)
        got1, got2 = self.prefix(512)
        assert np.array_equal(c_oracle.histogram(got1, 512, R1, 1, R1 + 1), want_z), "H1.e_z differs from the oracle"
        assert np.array_equal(c_oracle.histogram(got2, 512, R2, 1, R2 + 1), want_x), "H2.e_x differs from the oracle"
        assert int(self.hz.download((R1 + 1,), np.uint64).sum()) == self.batch
        if self.algo == "sparse":
            # the histogram-only route of the timed region (LDS-slab pipeline), forced onto the same 512-sample prefix
            from quantum_css_codes_amd import _native
            with self.ctx.flags(_native.F_SPARSE_SLABS):
                got_z, got_x = self.prefix_hist(512)
            assert np.array_equal(got_z, want_z) and np.array_equal(got_x, want_x), "slab pipeline differs from the oracle"

    def free(self):
        for name in ("ex", "ez", "s1", "s2", "hist"):
            buf = getattr(self, name, None)
            if buf is not None:
                buf.free()


def timed(ctx, path, steps, warmup):
    """Returns (stream ms, mean syndrome-kernel launch s, launches, histogram-kernel ms)."""
    from quantum_css_codes_amd import _native
    for _ in range(warmup):
        path.step()
    path.sync()
    path.hz.zero(), path.hx.zero()
    path.sync()
    for c in path.contexts():
        c.profile(True)
        c.profile_reset()
    ctx.timer_start()
    for _ in range(steps):
        path.step()
    if path.ctx2 is not None:
        path.ctx2.sync()                                     # the side stream's work ends inside the timed interval
    gpu_ms = ctx.timer_stop()
    syn_ms, syn_n, hist_ms = 0.0, 0, 0.0
    for c in path.contexts():
        ms, cnt = c.profile_get(_native.K_SYNDROME)
        syn_ms, syn_n = syn_ms + ms, syn_n + cnt
        hist_ms += c.profile_get(_native.K_HIST)[0]
        c.profile(False)
    return gpu_ms, syn_ms / 1e3 / max(1, syn_n), syn_n, hist_ms


def pmc_traffic(kernel):
    """(HBM-side bytes per 2^20-sample call of one component, where they come from): the committed rocprofv3 --pmc passes
    (profiles/traffic.json names the script and the commit they were taken at); not measured by this run."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    tj = json.load(open(tpath))
    return tj.get(kernel + "_bytes_per_launch"), "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, at " \
        "commit %s (profiles/r05_evidence.sh)" % tj.get("captured_at_commit", "?")


def traffic_stale():
    """True when the kernel sources are not the ones profiles/traffic.json was measured on (profiles/csrc_digest.py): the
    `traffic` figures of this line then describe an earlier build and want a new profiles/r05_evidence.sh run."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return True
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import csrc_digest
    return json.load(open(tpath)).get("captured_at_csrc_sha256") != csrc_digest.digest()


def roofline(path, mean_launch_s, launches):
    alg_bytes = path.batch * path.alg_bytes_per_sample
    achieved = alg_bytes / mean_launch_s / 1e9
    per, source = pmc_traffic(path.kernel)
    traffic = per * (path.batch >> 20) if per and path.batch % (1 << 20) == 0 else None
    return {"bound": "hbm", "kernel": path.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source, "algorithmic_bytes_per_launch": alg_bytes,
            "mean_launch_ms": mean_launch_s * 1e3, "launches": launches}


def roofline_of_steps(path, gpu_s, steps, per_call_s, calls):
    """Two-stream steps: the two components' pipelines overlap, so the duration that means something is a step's (HIP
    events on the main stream around the timed region, the side stream joined before the stop event).  A "launch" is one
    step = both components; the per-call durations on each stream (overlapped, hence longer than alone) are kept beside it."""
    alg_bytes = 2 * path.batch * path.alg_bytes_per_sample
    achieved = alg_bytes / (gpu_s / steps) / 1e9
    one, source = pmc_traffic(path.kernel)
    traffic = 2 * one * (path.batch >> 20) if one and path.batch % (1 << 20) == 0 else None
    return {"bound": "hbm", "kernel": path.kernel + " x2: H1.e_z and H2.e_x on two HIP streams, one launch = one step",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": source,
            "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_ms": gpu_s / steps * 1e3, "launches": steps,
            "mean_call_ms_on_its_stream": per_call_s * 1e3, "calls": calls}


def rref_numbers(ctx):
    """RREF GB/s = 2 * m * ceil(n/64) * 8 bytes / time on resident random matrices: one and 256 of
    2048 x 4096 (1 MiB each, L2-resident: latency-bound), and one 32768 x 65536 (256 MiB, streamed from HBM)."""
    from quantum_css_codes_amd import _native
    res = {}
    rng = np.random.default_rng(4096)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)).get("rref_bytes_per_call", {}) if os.path.exists(tpath) else {}

    def random_packed(m, n):
        ld = (n + 63) // 64
        return (rng.integers(0, 2**63, (m, ld), dtype=np.int64).view(np.uint64) << np.uint64(1)) | \
            rng.integers(0, 2, (m, ld), dtype=np.int64).view(np.uint64)

    for (m, n, batch, reps) in ((R1, N_QUBITS, 1, 3), (R1, N_QUBITS, 256, 3), (32768, 65536, 1, 1)):
        mats = [random_packed(m, n) for _ in range(batch)]              # `batch` DIFFERENT matrices (own pivot patterns each)
        a = mats[0]
        buf = ctx.alloc(batch * a.nbytes)
        piv, rk = ctx.alloc(batch * min(m, n) * 8), ctx.alloc(batch * 8)
        best = None
        for _ in range(reps):
            for b in range(batch):
                _native.check(_native.lib().gf2_h2d(ctx.handle, buf.ptr + b * a.nbytes, mats[b].ctypes.data, a.nbytes))
            ctx.timer_start()
            _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, m, n, a.shape[1], piv.ptr, rk.ptr))
            ms = ctx.timer_stop()
            best = ms if best is None else min(best, ms)
        assert int(rk.download((batch,), np.int64).min()) == min(m, n)
        del mats
        gbs = batch * 2 * a.nbytes / best / 1e6
        key = "%dx%d_x%d" % (m, n, batch)
        # integer work of Gauss-Jordan on packed words (what bin_matrix.py:27-29 does bit by bit): per pivot about m / 2 rows
        # take the pivot row over the 3/4 of the words that can still change = rank * m/2 * ld * 3/4 64-bit XORs, two 32-bit
        # lane-operations each, against 256 CUs x 64 lanes x 2.4 GHz; traffic = PMC bytes of the whole call (profiles/traffic.json)
        lane_ops = batch * 2.0 * min(m, n) * (m / 2.0) * a.shape[1] * 0.75
        moved = traffic.get(key)
        res[key] = {"ms": best, "GB/s": gbs, "frac_hbm_peak": gbs / HBM_PEAK_GBS, "algorithmic_bytes": batch * 2 * a.nbytes,
                    "matrices": "%d different random matrices" % batch if batch > 1 else "one random matrix",
                    "traffic": moved, "moved_GB/s": (moved / best / 1e6) if moved else None,
                    "int_op": {"lane_ops": lane_ops, "frac": lane_ops / (best / 1e3) / (256 * 64 * 2.4e9)}}
        buf.free(), piv.free(), rk.free()
    # many small matrices (one wavefront each, rows in registers): 256 MiB of 64 x 512, of 128 x 512 and (four pivots at a time) of
    # 64 x 1024 matrices, read once and written once
    for (m, n, batch) in ((64, 512, 65536), (128, 512, 32768), (64, 1024, 32768)):
        ld = n // 64
        some = random_packed(1024 * m, n).reshape(1024, m, ld)
        host = np.ascontiguousarray(np.tile(some, (batch // 1024, 1, 1)))
        buf = ctx.alloc(host.nbytes)
        piv, rk = ctx.alloc(batch * m * 8), ctx.alloc(batch * 8)
        best = None
        for _ in range(3):
            buf.upload(host)
            ctx.timer_start()
            _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, m, n, ld, piv.ptr, rk.ptr))
            ms = ctx.timer_stop()
            best = ms if best is None else min(best, ms)
        ranks = rk.download((batch,), np.int64)
        assert int(ranks.min()) >= m - 3 and int(ranks.max()) == m
        gbs = 2 * host.nbytes / best / 1e6
        res["%dx%d_x%d" % (m, n, batch)] = {"ms": best, "GB/s": gbs, "frac_hbm_peak": gbs / HBM_PEAK_GBS}
        buf.free(), piv.free(), rk.free()
    return res


def small_code_numbers(ctx):
    """BASELINE.json configs[1], configs[2]: the Steane code on 10^6 and the Reed-Muller [[15,1,3]] code on 10^7 Pauli
    errors (sampler + both syndromes + full histograms, one fused kernel, checked against the C oracle on a prefix), and
    the streaming syndrome kernel on 2^31 resident bit-sliced errors (working set beyond the Infinity Cache): exactly
    (n + r) / 8 bytes per sample and component move, so moved bytes = algorithmic bytes."""
    from oracle import c_oracle
    from quantum_css_codes_amd import _native
    from quantum_css_codes_amd.css_code import CSSCode
    steane = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
    cols = np.arange(1, 16)
    rm_h1 = np.array([(cols >> b) & 1 for b in range(4)])
    rm_h2 = np.vstack([rm_h1] + [rm_h1[a] & rm_h1[b] for a in range(4) for b in range(a + 1, 4)])
    res = {}
    p = P_TOTAL / 3
    for name, code, count in (("steane_7_1_3", CSSCode(steane, steane), 10**6), ("reed_muller_15_1_3", CSSCode(rm_h1, rm_h2), 10**7)):
        n, r1, r2 = code.n, code.r_1, code.r_2
        h1p, h2p = _native.pack_rows(code.parity_check_c1), _native.pack_rows(code.parity_check_c2)
        want_z, want_x = c_oracle.mc(h1p, r1, h2p, r2, n, SEED, 0, 20000, p, p, p, 0)
        got = code.monte_carlo(20000, p, p, p, seed=SEED)
        assert np.array_equal(got['hist_z'], want_z) and np.array_equal(got['hist_x'], want_x), name + " differs from the oracle"
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            out = code.monte_carlo(count, p, p, p, seed=SEED)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert int(out['hist_z'].sum()) == count
        chk = ctx.check_create(h2p, r2, n)
        big = 1 << 31
        bw = big // 64
        eb, sb = ctx.alloc(n * bw * 8), ctx.alloc(r2 * bw * 8)
        chunk = np.random.default_rng(1).integers(0, 2**63, 1 << 22, dtype=np.int64).view(np.uint64)
        for off in range(0, n * bw, chunk.size):                  # any bits will do for a bandwidth measurement
            _native.check(_native.lib().gf2_h2d(ctx.handle, eb.ptr + off * 8, chunk.ctypes.data,
                                                min(chunk.size, n * bw - off) * 8))
        ctx.syndrome_dev(chk, eb, big, bw, sb, bw, _native.LAYOUT_BIT_SLICED)
        ctx.sync()
        ctx.timer_start()
        for _ in range(5):
            ctx.syndrome_dev(chk, eb, big, bw, sb, bw, _native.LAYOUT_BIT_SLICED)
        ms = ctx.timer_stop() / 5
        moved = (n + r2) * bw * 8
        eb.free(), sb.free()
        res[name] = {"monte_carlo": {"samples": count, "value": count / best, "unit": "syndromes/s",
                                     "what": "CSSCode.monte_carlo: fused sampler + both syndromes + full histograms, host "
                                             "wall time incl. histogram download"},
                     "bit_sliced_stream": {"samples": big, "value": big / (ms / 1e3), "unit": "syndromes/s",
                                           "bytes_per_syndrome": (n + r2) / 8.0,
                                           "roofline": {"bound": "hbm", "kernel": "syndrome_sliced_kernel",
                                                        "achieved": moved / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                        "frac": moved / ms / 1e6 / HBM_PEAK_GBS}}}
    return res


COMM_TIMEOUT_EXIT = 75         # (EX_TEMPFAIL: nothing else in this program or in the interpreter exits with it) a rank leaves with this code when its RCCL communicator did not come up in time (guarded_comm)


def launch_ranks(gpus, argv=None, deadline_s=3600.0):
    """`python bench.py --gpus N` without a launcher: the N ranks as child processes of this one, which never touches the
    GPU (no re-exec of a process that has: the children are started first thing).  Rank 0 prints the JSON line on the
    inherited stdout; the exit code is the worst of the ranks'.  Every rank is the leader of a process group of its own:
    when one rank fails, when this process is told to stop (SIGTERM / SIGINT are forwarded) or when the deadline passes, the
    others are terminated and, if they have not gone within five seconds -- a rank inside an RCCL or HIP call does not see
    SIGTERM --, killed, so no rank outlives the launcher holding its GPU.  Ranks that leave with COMM_TIMEOUT_EXIT (the
    library's own communicator did not come up, --allreduce gf2) are started once more, fresh, with --allreduce torch."""
    import signal
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    stop = {"signal": None}

    def on_signal(signum, _frame):
        stop["signal"] = signum

    previous = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}

    def end(procs):
        for proc in procs:
            if proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGTERM)
                except OSError:
                    pass
        limit = time.monotonic() + 5.0
        for proc in procs:
            try:
                proc.wait(max(0.0, limit - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()

    def one_round(args):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        procs = []
        try:
            for r in range(gpus):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + args, env=env, start_new_session=True))
            worst, live, t_end = 0, list(procs), time.monotonic() + deadline_s
            while live:
                time.sleep(0.1)
                if stop["signal"] is not None:
                    return 128 + int(stop["signal"])
                if time.monotonic() > t_end:
                    sys.stderr.write("[bench] ranks still running after %.0f s: stopping them\n" % deadline_s)
                    return 124
                for proc in list(live):
                    rc = proc.poll()
                    if rc is None:
                        continue
                    live.remove(proc)
                    if rc != 0:
                        worst = worst or (rc if rc > 0 else 128 - rc)
                        return worst                       # the others are stopped below
            return worst
        finally:
            end(procs)                                     # exactly the process groups started above

    try:
        rc = one_round(argv)
        # (asked of the parsed arguments, not of the raw list: `--allreduce=gf2` is one token)
        wants_gf2 = any(a == "gf2" and i > 0 and argv[i - 1] == "--allreduce" for i, a in enumerate(argv)) or "--allreduce=gf2" in argv
        if rc == COMM_TIMEOUT_EXIT and wants_gf2:
            sys.stderr.write("[bench] a rank's RCCL communicator did not come up: fresh ranks with --allreduce torch\n")
            sys.stderr.flush()
            rc = one_round(argv + ["--allreduce", "torch"])
        return rc
    finally:
        for sig, handler in previous.items():
            signal.signal(sig, handler)


def guarded_comm(ctx, nbins, seconds=120.0):
    """libgf2hip's communicator over all ranks, proven by one all-reduce -- (comm, "") --, or (None, why) when it cannot be had:
    then every rank sums through torch.distributed instead.  The id exchange runs here (a collective of the process group);
    ncclCommInitRank and the first all-reduce run in a helper thread with a time limit.  What the ranks made of it they tell
    each other through the process group's key-value store (host side: no GPU collective while a helper may be stuck in one):
      every rank got through              -> the communicator;
      a rank's attempt came back failed   -> (None, why) on every rank; a rank that does hold a communicator closes it;
      a rank's helper has not come back   -> that thread sits inside RCCL on this context's device, so the context must not
                                             be used any further: every rank says why on stderr, flushes and leaves through
                                             os._exit(COMM_TIMEOUT_EXIT) -- an exit, never a re-exec; the stuck thread cannot
                                             hold the interpreter's teardown.  bench's own launcher then starts fresh ranks with
                                             --allreduce torch (launch_ranks); under another launcher the run is lost, which is
                                             why --allreduce gf2 is not the default for N > 1 before an N > 1 run has proven it."""
    import datetime
    import threading
    import torch.distributed as dist
    from quantum_css_codes_amd import _native
    grouped = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if grouped else (0, 1)
    ident = [None]
    if rank == 0:
        try:
            ident[0] = _native.Comm.unique_id()
        except _native.GF2Error as err:                # librccl does not load: every rank hears of it
            ident[0] = str(err)[:120]
    if grouped:
        dist.broadcast_object_list(ident, src=0)
    if not isinstance(ident[0], bytes):
        return None, " [gf2_comm_unique_id: %s]" % ident[0]
    box = {}

    def work():
        try:
            if os.environ.get("BENCH_FAULT") == "comm_timeout":        # tests/test_rccl.py: a communicator that never comes up
                time.sleep(3600)
            comm = _native.Comm(ctx, ident[0], world, rank)
            box["comm"] = comm
            probe = ctx.alloc(nbins * 8).upload(np.full(nbins, 1, dtype=np.uint64))
            comm.allreduce(probe, nbins)
            got = probe.download((nbins,), np.uint64)
            probe.free()
            if not (got == world).all():
                raise RuntimeError("gf2_hist_allreduce: wrong sum")
            box["ok"] = True
        except Exception as err:                       # noqa: BLE001 -- whatever it is, the fallback runs
            box["why"] = str(err)[:120]

    helper = threading.Thread(target=work, daemon=True)
    helper.start()
    helper.join(seconds)
    mine = "ok" if box.get("ok") else ("err:" + box["why"] if "why" in box else "timeout")
    verdicts = [mine]
    if grouped and world > 1:
        # the verdicts travel under keys of THIS call (a counter: a second call in the same process group must not read the first
        # one's); the store is the process group's own (there is no public accessor for it: without one the ranks agree through a
        # host-side all_gather_object instead, which a rank stuck in RCCL cannot hold up either -- gloo -- or would anyway -- nccl)
        guarded_comm.calls = getattr(guarded_comm, "calls", 0) + 1
        get_store = getattr(dist.distributed_c10d, "_get_default_store", None)
        if get_store is None:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            verdicts = gathered
        else:
            store = dist.PrefixStore("gf2_comm/%d/" % guarded_comm.calls, get_store())
            store.set(str(rank), mine)
            keys = [str(r) for r in range(world)]
            try:
                store.wait(keys, datetime.timedelta(seconds=seconds + 60.0))
                verdicts = [store.get(k).decode("utf-8", "replace") for k in keys]
            except Exception as err:                   # noqa: BLE001 -- a rank never answered: treated as stuck
                verdicts = [mine, "timeout (%s)" % str(err)[:60]]
    if any(v.startswith("timeout") for v in verdicts):
        sys.stderr.write("[bench] rank %d: gf2_comm_create / first gf2_hist_allreduce not back within %.0f s on %s: leaving with code "
                         "%d (the context cannot be used beside a thread that is inside RCCL)\n"
                         % (rank, seconds, "this rank" if mine == "timeout" else "another rank", COMM_TIMEOUT_EXIT))
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(COMM_TIMEOUT_EXIT)
    if all(v == "ok" for v in verdicts):
        return box["comm"], ""
    if "comm" in box and not helper.is_alive():
        try:
            box["comm"].close()                        # this rank's came up, another's did not
        except Exception:                              # noqa: BLE001
            pass
    why = next(v[4:] for v in verdicts if v.startswith("err:"))
    return None, " [gf2_comm_create: %s]" % why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-log2", type=int, default=27,
                    help="samples per GPU per step (2^k); 2^27 = 128 GiB of packed errors resident in HBM, 38 ms per step: the "
                         "driver's 20 steps are a timed region of 0.75 s (round 2 ran 2^24 = 16 GiB, 0.09 s; same rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-settle", action="store_true", help="skip the 0.1 s of untimed steps before the warm-up steps")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other syndrome kernel and the RREF timings")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo lets several ranks share one GPU to rehearse the multi-process path (histograms are "
                         "then all-reduced on the host); the driver's runs use nccl = RCCL")
    ap.add_argument("--allreduce", choices=("gf2", "torch"), default="torch",
                    help="the histogram all-reduce of the ranks on the nccl backend: torch = torch.distributed.all_reduce (RCCL through "
                         "torch; the default until an N > 1 run has proven the other), gf2 = gf2_hist_allreduce (libgf2hip's own RCCL "
                         "communicator, device buffer, the context's stream; set up under a time limit: guarded_comm)")
    ap.add_argument("--launch-deadline", type=float, default=3600.0,
                    help="seconds after which the self-launched ranks (no launcher: WORLD_SIZE unset) are stopped")
    ap.add_argument("--comm-timeout", type=float, default=120.0, help="seconds --allreduce gf2 waits for its communicator")
    ap.add_argument("--total-samples", type=int, default=None,
                    help="strong scaling: this many samples of the global stream in total, rank g of N taking the g-th of N contiguous "
                         "shards (montecarlo.shard_range) -- 100000000 is BASELINE.json configs[4] to the letter; the line then says "
                         "\"scaling\": \"strong\" and a step is one pass over the rank's shard.  Default: 2^batch-log2 samples per "
                         "GPU whatever N (weak scaling)")
    ap.add_argument("--algo", choices=("sparse", "dense"), default="sparse")
    ap.add_argument("--slab-pass-log2", type=int, default=None,
                    help="samples per pass of the slab pipeline through its workspace (GF2_OPT_SLAB_PASS_LOG2)")
    ap.add_argument("--ctx-flags", type=lambda v: int(v, 0), default=0, help="routing flags (GF2_F_*) of both contexts")
    ap.add_argument("--opt", action="append", default=[], metavar="K=V",
                    help="set a context tunable (GF2_OPT_* number = value), e.g. --opt 2=0")
    ap.add_argument("--one-stream", action="store_true",
                    help="issue both components of a step on one HIP stream (default: H2.e_x goes to a second context)")
    args = ap.parse_args()

    # without a launcher this process only supervises: N > 1 ranks, and also the single rank of an explicit --allreduce gf2,
    # which must be able to leave (guarded_comm) and be started afresh
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.allreduce == "gf2"):
        sys.exit(launch_ranks(args.gpus, deadline_s=args.launch_deadline))
    if os.environ.get("BENCH_FAULT") == "hang" and "WORLD_SIZE" in os.environ:
        # tests/test_rccl.py: a rank that sits in a call that does not see SIGTERM (what a rank inside RCCL looks like to its launcher)
        import signal
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(3600)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))

    import torch
    import torch.distributed as dist
    device = local_rank % max(1, torch.cuda.device_count())
    os.environ.setdefault("GF2_DEVICE", str(device))
    if world > 1:
        torch.cuda.set_device(device)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from quantum_css_codes_amd import _native
    from quantum_css_codes_amd.montecarlo import all_reduce_histograms
    ctx = _native.default_context()
    # the ranks' own RCCL communicator (its id travels over the process group's rendezvous); ranks that share a GPU
    # (--dist-backend gloo) cannot form one and sum on the host
    comm, comm_note = None, ""
    if args.dist_backend == "nccl" and args.allreduce == "gf2":
        comm, comm_note = guarded_comm(ctx, R1 + 1 + R2 + 1, args.comm_timeout)
    code, h1, h2 = build_code()
    chk1, chk2 = ctx.check_create(h1, R1, N_QUBITS), ctx.check_create(h2, R2, N_QUBITS)
    if args.total_samples is None:
        batch = 1 << args.batch_log2
        first = rank * batch           # this rank's shard of the global sample stream: sample i = f(seed, i)
        total_per_step = world * batch
    else:
        from quantum_css_codes_amd.montecarlo import shard_range
        first, batch = shard_range(0, args.total_samples, rank, world)
        total_per_step = args.total_samples
        if batch < 512:
            raise SystemExit("--total-samples %d leaves rank %d with %d samples: at least 512 per rank" % (args.total_samples, rank, batch))

    ctx2 = None if args.one_stream else _native.Context(ctx.device)
    for c in (ctx, ctx2):
        if c is not None and args.slab_pass_log2 is not None:
            c.set_option(_native.OPT_SLAB_PASS_LOG2, args.slab_pass_log2)
        if c is not None and args.ctx_flags:
            c.set_flags(args.ctx_flags)
        for kv in args.opt:
            if c is not None:
                c.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    path = Path(ctx, args.algo, chk1, chk2, batch, first, ctx2)
    if rank == 0:
        path.check_against_oracle(h1, h2, first)        # correctness of what is about to be timed

    def fence():
        path.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Part of the set-up, like the oracle check above: about 0.1 s of untimed steps bring the clocks to the state a long run
    # sees (from a cold start a short measurement reads up to 6 % low); reported as settle_steps.  The W warm-up steps of
    # the contract follow.
    settle_steps = 0
    if not args.no_settle:
        path.step()
        path.sync()
        t_one = time.perf_counter()
        path.step()
        path.sync()
        t_one = max(1e-5, time.perf_counter() - t_one)
        settle_steps = 2 + max(1, min(400, int(0.1 / t_one)))
        for _ in range(settle_steps - 2):
            path.step()
        path.sync()
    for _ in range(args.warmup):
        path.step()
    path.sync()
    if comm is not None:
        comm.allreduce(path.hist, R1 + 1 + R2 + 1)       # the first collective sets up the channels: untimed
    elif world > 1:
        all_reduce_histograms([path.hz.download((R1 + 1,), np.uint64), path.hx.download((R2 + 1,), np.uint64)])
    path.sync()
    path.hz.zero(), path.hx.zero()
    # per-call HIP events cost the two-stream step about 2.5 %: there the timed region runs without them (its roofline
    # needs the step time only) and the per-call durations come from a few further steps after it
    call_events = path.ctx2 is None
    for c in path.contexts():
        c.profile(call_events)
        c.profile_reset()
    fence()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        path.step()
    if path.ctx2 is not None:
        path.ctx2.sync()                                 # the side stream's work ends inside the timed interval
    gpu_ms = ctx.timer_stop()
    if comm is not None:
        comm.allreduce(path.hist, R1 + 1 + R2 + 1)       # in place, on the context's stream, behind the last step
    hist_z = path.hz.download((R1 + 1,), np.uint64)
    hist_x = path.hx.download((R2 + 1,), np.uint64)
    if world > 1 and comm is None:
        hist_z, hist_x = all_reduce_histograms([hist_z, hist_x])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total = args.steps * total_per_step
    assert int(hist_z.sum()) == total and int(hist_x.sum()) == total
    if not call_events:
        for c in path.contexts():
            c.profile(True)
            c.profile_reset()
        for _ in range(max(2, min(20, (1 << 24) // batch))):
            path.step()
        path.sync()
    syn_ms, syn_n, hist_ms = 0.0, 0, 0.0
    for c in path.contexts():
        ms, cnt = c.profile_get(_native.K_SYNDROME)
        syn_ms, syn_n = syn_ms + ms, syn_n + cnt
        hist_ms += c.profile_get(_native.K_HIST)[0]
        c.profile(False)

    # every rank's shard and its own roofline figure (both components' algorithmic bytes of its steps over its stream time), for the
    # N > 1 line: the driver computes scaling from `value`, this shows which GPU held it back
    shards = None
    if world > 1:
        dev = "cuda" if args.dist_backend == "nccl" else "cpu"
        mine = torch.tensor([float(first), float(batch), float(gpu_ms)], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        shards = []
        for r, v in enumerate(gathered):
            f_r, n_r, ms_r = (float(x) for x in v.tolist())
            gbs = 2.0 * n_r * path.alg_bytes_per_sample * args.steps / (ms_r * 1e-3) / 1e9
            shards.append({"rank": r, "first": int(f_r), "count": int(n_r), "stream_ms": ms_r,
                           "roofline": {"achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}})
    if rank == 0:
        roof = (roofline_of_steps(path, gpu_ms / 1e3, args.steps, syn_ms / 1e3 / max(1, syn_n), syn_n)
                if path.ctx2 is not None else roofline(path, syn_ms / 1e3 / max(1, syn_n), syn_n))
        # the chip's streaming ceiling measured in this run (plain 16-byte loads over the resident errors), next to the
        # 8 TB/s specification the fraction is quoted against
        probe_bytes = min(batch * 512, 1 << 32)
        measured = ctx.membw_probe(path.ex, probe_bytes)
        roof["traffic_stale"] = traffic_stale()
        roof["measured_read_peak_GBs"] = measured
        roof["frac_of_measured_peak"] = roof["achieved"] / measured
        # integer work of the algorithm as it is run: per sample and component (w + 2) * r / 32 32-bit XOR
        # and popcount lane-operations, w = expected listed columns (n - r)(p_x + p_y); peak = CUs x 64 lanes x 2.4 GHz
        w_mean = (N_QUBITS - R1) * (2.0 * P_TOTAL / 3.0)
        lane_ops = 2.0 * batch * (w_mean + 2.0) * (R1 / 32.0)
        int_peak = 256 * 64 * 2.4e9
        roof["int_op"] = {"lane_ops_per_launch": lane_ops, "achieved_per_s": lane_ops / (roof["mean_launch_ms"] / 1e3 * (1 if path.ctx2 is not None else 2)),
                          "peak_per_s": int_peak, "frac": lane_ops / (roof["mean_launch_ms"] / 1e3 * (1 if path.ctx2 is not None else 2)) / int_peak,
                          "what": "32-bit XOR/popcount lane-operations of the sparse algorithm ((w + 2) r / 32 per sample and "
                                  "component, w = %.2f listed columns) against 256 CUs x 64 lanes x 2.4 GHz" % w_mean}
        out = {
            "metric": "syndromes/sec (n=4096 CSS)", "value": total / elapsed, "unit": "syndromes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle_steps,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.total_samples is None else "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[4]: n=4096 CSS Monte-Carlo, random dual code (H1 2048x4096, H2 2047x4096, "
                                   "standard form), depolarising p=0.01, errors resident in HBM",
                       "algo": args.algo, "samples_per_gpu_per_step": batch, "global_samples_per_step": total_per_step,
                       "streams": 2 if path.ctx2 is not None else 1,
                       "parallelism": "sample-range shards, 1 histogram all-reduce"
                                      + ("" if world == 1 else " (gf2_hist_allreduce over librccl)" if comm is not None else
                                         " (torch.distributed, %s)%s" % (args.dist_backend, comm_note))},
            "roofline": roof,
            "collective": None if world == 1 else
                          {"call": "gf2_hist_allreduce (C ABI, librccl)" if comm is not None else "torch.distributed.all_reduce",
                           "backend": "rccl" if (comm is not None or args.dist_backend == "nccl") else args.dist_backend,
                           "rccl_ranks": world if (comm is not None or args.dist_backend == "nccl") else 0,
                           "bins": R1 + 1 + R2 + 1, "dtype": "uint64 sum"},
            "shards": shards,
            "checks": {"histogram_total": int(hist_z.sum()), "expected_total": int(total),
                       "histogram_sha256": hashlib.sha256(hist_z.tobytes() + hist_x.tobytes()).hexdigest(),
                       "oracle_prefix": "512 samples of this rank's batch through the timed path == oracle/gf2_oracle.c",
                       "inputs": "H1, H2, both standard forms and logical operators == digests of the reference's own "
                                 "constructor (tests/golden/config4_golden.npz)"},
            "kernel_ms": {"syndrome_calls": syn_ms, "syndrome_calls_counted": syn_n, "histogram": hist_ms,
                          "stream_total": gpu_ms},
        }
        if world == 1 and not args.no_cpu_baseline:               # timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(code, h1, h2)          # (h1, h2: the packed standard forms)
        if world == 1 and not args.no_secondary:
            single_z = (hist_z // np.uint64(args.steps)).astype(np.uint64)      # same batch every step
            # end to end: nothing resident, gf2_mc_run draws the errors itself (record sampler, then gather / combine / misfits per
            # component, in line on one stream); its histogram over this rank's batch must be the timed path's
            mc_count = max(batch, 1 << 24)
            # (the first call of this size allocates the context's workspaces: untimed)
            ctx.mc_run(chk1, chk2, SEED, 0, mc_count, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3, _native.HIST_WEIGHT)
            t_mc = None
            for _ in range(2):
                t0 = time.perf_counter()
                mc_z, _ = ctx.mc_run(chk1, chk2, SEED, first, mc_count, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3, _native.HIST_WEIGHT)
                t1 = time.perf_counter() - t0
                t_mc = t1 if t_mc is None else min(t_mc, t1)
            assert int(mc_z.sum()) == mc_count
            if mc_count == batch:
                assert np.array_equal(mc_z, single_z), "end-to-end Monte-Carlo and the resident-error path disagree"
            else:
                assert np.array_equal(ctx.mc_run(chk1, chk2, SEED, first, batch, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3,
                                                 _native.HIST_WEIGHT)[0], single_z)
            # BASELINE.json configs[4] to the letter, on this one GPU: 10^8 samples (the first 10^8 of the resident batch), both
            # components on the two streams of the timed region, histograms included; `--total-samples 100000000 --gpus N` is
            # the same over N ranks (1/N of it each)
            literal = None
            if args.algo == "sparse" and batch >= 10**8:
                lde = _native.words_for(N_QUBITS)
                side = path.ctx2 if path.ctx2 is not None else ctx
                lit_z, lit_x = ctx.alloc((R1 + 1) * 8).zero(), ctx.alloc((R2 + 1) * 8).zero()

                def literal_step():
                    ctx.syndrome_sparse_dev(chk1, path.ez, 10**8, lde, None, 0, lit_z, R1 + 1)
                    side.syndrome_sparse_dev(chk2, path.ex, 10**8, lde, None, 0, lit_x, R2 + 1)
                literal_step()
                path.sync()
                lit_z.zero(), lit_x.zero()
                path.sync()
                ctx.timer_start()
                for _ in range(5):
                    literal_step()
                if path.ctx2 is not None:
                    path.ctx2.sync()
                lit_ms = ctx.timer_stop() / 5
                assert int(lit_z.download((R1 + 1,), np.uint64).sum()) == 5 * 10**8
                literal = {"samples": 10**8, "ms": lit_ms, "value": 10**8 / (lit_ms / 1e3), "unit": "syndromes/s",
                           "frac_hbm_peak": 10**8 * 2 * path.alg_bytes_per_sample / lit_ms / 1e6 / HBM_PEAK_GBS,
                           "what": "configs[4]'s 10^8 error samples, resident, H1.e_z and H2.e_x + both weight histograms, one GPU"}
                lit_z.free(), lit_x.free()
            path.free()
            # the other implementations on 2^20 resident samples of the same stream (one stream, per-call rooflines)
            sec = min(batch, 1 << 20)
            gather, single, stored = None, None, None
            plain = Path(ctx, "sparse", chk1, chk2, sec, first)
            s_ms, s_launch, s_n, _ = timed(ctx, plain, 50, 5)
            sec_z = (plain.hz.download((R1 + 1,), np.uint64) // np.uint64(50)).astype(np.uint64)
            assert int(sec_z.sum()) == sec
            single = {"value": 50 * sec / (s_ms / 1e3), "unit": "syndromes/s", "ms_per_step": s_ms / 50,
                      "roofline": roofline(plain, s_launch, s_n)}
            with ctx.flags(_native.F_SPARSE_GATHER):
                g_ms, g_launch, g_n, _ = timed(ctx, plain, 10, 2)
                assert np.array_equal(plain.hz.download((R1 + 1,), np.uint64), sec_z * np.uint64(10)), \
                    "column-gather kernel and slab pipeline disagree"
            plain.kernel = "syndrome_sparse_kernel"
            gather = {"value": 10 * sec / (g_ms / 1e3), "unit": "syndromes/s", "ms_per_step": g_ms / 10,
                      "roofline": roofline(plain, g_launch, g_n)}
            # SURVEY.md 8d's read + write variant: the same pipeline with the syndromes stored (every gather workgroup its slab's
            # 64-byte piece; round 3 went to the column-gather kernel for this), 1536 B per sample (2 x 512 read, 256 + 255.9
            # written) instead of 1024 read
            s1, s2 = ctx.alloc(sec * plain.ls1 * 8), ctx.alloc(sec * plain.ls2 * 8)
            lde = _native.words_for(N_QUBITS)
            for _ in range(2):
                ctx.syndrome_sparse_dev(chk1, plain.ez, sec, lde, s1, plain.ls1)
                ctx.syndrome_sparse_dev(chk2, plain.ex, sec, lde, s2, plain.ls2)
            ctx.sync()
            ctx.timer_start()
            for _ in range(10):
                ctx.syndrome_sparse_dev(chk1, plain.ez, sec, lde, s1, plain.ls1)
                ctx.syndrome_sparse_dev(chk2, plain.ex, sec, lde, s2, plain.ls2)
            rw_ms = ctx.timer_stop() / 10
            from oracle import c_oracle
            e_head = plain.ez.download((64, lde), "<u8")
            assert np.array_equal(s1.download((64, plain.ls1), "<u8"), c_oracle.syndrome_batch(h1, R1, N_QUBITS, e_head, 64)), \
                "stored syndromes differ from the oracle"
            rw_bytes = sec * (2 * N_QUBITS / 8.0 + (R1 + R2) / 8.0)
            stored = {"value": sec / (rw_ms / 1e3), "unit": "syndromes/s", "ms_per_step": rw_ms,
                      "bytes_per_syndrome": 2 * N_QUBITS / 8.0 + (R1 + R2) / 8.0,
                      "roofline": {"bound": "hbm", "kernel": "slab_pipeline (syndromes stored, no histogram)", "achieved": rw_bytes / rw_ms / 1e6,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rw_bytes / rw_ms / 1e6 / HBM_PEAK_GBS}}
            s1.free(), s2.free()
            plain.free()
            other = Path(ctx, "dense" if args.algo == "sparse" else "sparse", chk1, chk2, sec, first)
            other.check_against_oracle(h1, h2, first)
            o_ms, o_launch, o_n, o_hist = timed(ctx, other, 10, 2)
            other_z = other.hz.download((R1 + 1,), np.uint64)
            assert np.array_equal(other_z, sec_z * np.uint64(10)), "the two syndrome kernels disagree"
            out["secondary"] = {
                "monte_carlo_end_to_end": {"value": mc_count / t_mc, "unit": "syndromes/s",
                                           "what": "gf2_mc_run: record sampler (records and identity words, no packed rows), gather, combine, misfit "
                                                   "kernels in line on one stream, chunks of 2^22 samples, no resident input, host wall time incl. "
                                                   "histogram download, %d samples, best of two calls after one untimed call" % mc_count},
                other.algo + "_kernel": {"value": 10 * sec / (o_ms / 1e3), "unit": "syndromes/s", "ms_per_step": o_ms / 10,
                                         "roofline": roofline(other, o_launch, o_n), "histogram_ms_per_step": o_hist / 10},
                "rref": rref_numbers(ctx)}
            if literal is not None:
                out["secondary"]["configs4_literal_1e8"] = literal
            out["secondary"]["column_gather_kernel"] = gather
            out["secondary"]["one_stream"] = single
            out["secondary"]["read_write_1536B"] = stored
            out["secondary"]["small_codes"] = small_code_numbers(ctx)
            other.free()
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
