#!/usr/bin/env python3
"""
bench.py -- syndromes/s of the n = 4096 CSS Monte-Carlo hot path on N x MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[4]): the random dual code of config 4 --
H1 = default_rng(4096) 2048 x 4096, H2 = first 2047 rows of nullspace(H1), both put in standard form by
CSSCode exactly as css_code.py:51-61 does -- and depolarising errors (p = 0.01) from the counter-based
sampler, pre-materialised in HBM before the timed region (2^20 samples per GPU, 1 GiB of packed errors:
larger than the 256 MiB Infinity Cache, so every step streams from HBM).

One step = one pass of the hot path over that batch: s_z = H1 . e_z and s_x = H2 . e_x for every sample
(two launches of the dominant kernel) plus the two weight histograms.  Per-GPU work is fixed as N grows
("weak"); ranks never exchange data on the path; the histograms are summed once with one all-reduce
(RCCL) inside the timed region.  value = N * K * batch / max-over-ranks time.

The JSON line also carries
  roofline      dominant kernel (syndrome_tiled_kernel) against the HBM roofline: algorithmic bytes per
                launch = batch * (n/8 read + r/8 written) (1536 B per sample over the two
                launches) / its mean launch time, measured live with HIP events on the kernel's stream.
  cpu_baseline  the reference's NumPy path (oracle/cpu_ref.py restatement of css_code.py:728) timed on this
                host, 1 core, on a bounded sample of the same workload.
"""
import argparse
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
    """Config 4 of BASELINE.json on the GPU: returns packed standard-form H1, H2 (uint64 words)."""
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
    return code, _native.pack_rows(code.parity_check_c1), _native.pack_rows(code.parity_check_c2)


def cpu_baseline(code, seconds_target=12.0):
    """Reference-style CPU path: np.mod(np.matmul(H, e), 2) per error vector on dense int64 arrays
    (css_code.py:728), both Pauli components, single thread.  Bounded sample, extrapolated rate."""
    from oracle import cpu_ref, c_oracle
    h1 = np.array(code.parity_check_c1, dtype='int')
    h2 = np.array(code.parity_check_c2, dtype='int')
    ex, ez = c_oracle.sample_errors(N_QUBITS, SEED, 0, 512, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3)
    ex, ez = c_oracle.unpack_rows(ex, N_QUBITS), c_oracle.unpack_rows(ez, N_QUBITS)
    done, t0 = 0, time.perf_counter()
    while done < 512:
        cpu_ref.syndrome_product(h1, ez[done])
        cpu_ref.syndrome_product(h2, ex[done])
        done += 1
        if time.perf_counter() - t0 > seconds_target and done >= 16:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "syndromes/s", "cores": 1, "kind": "port",
            "sample": "%d errors of the same n=4096 code, np.mod(np.matmul(H,e),2) for H1.e_z and H2.e_x "
                      "(oracle/cpu_ref.py restating css_code.py:728), %.1f s, host has %d cores"
                      % (done, dt, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-log2", type=int, default=20, help="samples per GPU per step (2^k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time RREF and the sampler-inclusive pipeline")
    ap.add_argument("--algo", choices=("sparse", "dense"), default="sparse",
                    help="sparse: one wavefront per sample XORs the check's column for each set error bit, weight "
                         "histogram fused (work ~ error weight; the Monte-Carlo path at p=0.01).  dense: Four-Russians "
                         "table kernel on tiled errors + histogram kernel (data-independent)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    os.environ.setdefault("GF2_DEVICE", str(local_rank))

    import torch
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from quantum_css_codes_amd import _native
    ctx = _native.default_context()
    code, h1, h2 = build_code()
    chk1, chk2 = ctx.check_create(h1, R1, N_QUBITS), ctx.check_create(h2, R2, N_QUBITS)

    batch = 1 << args.batch_log2
    ls1, ls2 = _native.words_for(R1), _native.words_for(R2)
    hz, hx = ctx.alloc((R1 + 1) * 8), ctx.alloc((R2 + 1) * 8)
    # this rank's shard of the global sample stream (sample i is a function of (seed, i) only)
    first = rank * batch
    if args.algo == "sparse":
        # sample-major packed errors resident in HBM; histogram-only output (no syndromes written)
        lde = _native.words_for(N_QUBITS)
        ex, ez = ctx.alloc(batch * lde * 8), ctx.alloc(batch * lde * 8)
        s1 = s2 = None
        ctx.sample_errors_dev(N_QUBITS, SEED, first, batch, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3, ex, ez, lde)
        kernel_name = "syndrome_sparse_kernel"
        alg_bytes_per_sample = N_QUBITS / 8.0                       # SURVEY.md 8d read-only variant, per component

        def step():
            ctx.syndrome_sparse_dev(chk1, ez, batch, lde, None, 0, hz, R1 + 1)
            ctx.syndrome_sparse_dev(chk2, ex, batch, lde, None, 0, hx, R2 + 1)

        def prefix_syndromes(count):
            a, b = ctx.alloc(count * ls1 * 8).zero(), ctx.alloc(count * ls2 * 8).zero()
            ctx.syndrome_sparse_dev(chk1, ez, count, lde, a, ls1)
            ctx.syndrome_sparse_dev(chk2, ex, count, lde, b, ls2)
            return a.download((count, ls1), "<u8"), b.download((count, ls2), "<u8")
    else:
        lde = _native.tiled_ld(N_QUBITS)
        tiled = _native.LAYOUT_TILED      # device-native error layout (include/gf2hip.h), written by the sampler
        ex, ez = ctx.alloc(_native.tiled_words(N_QUBITS, batch) * 8), ctx.alloc(_native.tiled_words(N_QUBITS, batch) * 8)
        s1, s2 = ctx.alloc(batch * ls1 * 8), ctx.alloc(batch * ls2 * 8)
        ctx.sample_errors_dev(N_QUBITS, SEED, first, batch, P_TOTAL / 3, P_TOTAL / 3, P_TOTAL / 3, ex, ez, lde, tiled)
        kernel_name = "syndrome_tiled_kernel"
        alg_bytes_per_sample = N_QUBITS / 8.0 + (R1 + R2) / 2.0 / 8.0   # packed error read + packed syndrome written

        def step():
            # tiled errors in, slab-major syndromes out (word s of sample b at s*batch + b)
            ctx.syndrome_dev(chk1, ez, batch, lde, s1, batch, tiled)
            ctx.syndrome_dev(chk2, ex, batch, lde, s2, batch, tiled)
            ctx.histogram_dev(s1, batch, batch, R1, _native.HIST_WEIGHT, hz, R1 + 1, tiled)
            ctx.histogram_dev(s2, batch, batch, R2, _native.HIST_WEIGHT, hx, R2 + 1, tiled)

        def prefix_syndromes(count):
            return (np.ascontiguousarray(s1.download((ls1, batch), "<u8")[:, :count].T),
                    np.ascontiguousarray(s2.download((ls2, batch), "<u8")[:, :count].T))
    ctx.sync()

    # ---- correctness of what is about to be timed: a prefix of the batch against the oracle ----------------
    hz.zero(), hx.zero()
    step()
    ctx.sync()
    if rank == 0:
        from oracle import c_oracle
        want_z, want_x = c_oracle.mc(h1, R1, h2, R2, N_QUBITS, SEED, first, 512, P_TOTAL / 3, P_TOTAL / 3,
                                     P_TOTAL / 3, 1)
        got_s1, got_s2 = prefix_syndromes(512)
        assert np.array_equal(c_oracle.histogram(got_s1, 512, R1, 1, R1 + 1), want_z), "H1.e_z differs from the oracle"
        assert np.array_equal(c_oracle.histogram(got_s2, 512, R2, 1, R2 + 1), want_x), "H2.e_x differs from the oracle"
    single = hz.download((R1 + 1,), np.uint64)
    assert int(single.sum()) == batch

    for _ in range(args.warmup):
        step()
    ctx.sync()
    hz.zero(), hx.zero()
    ctx.sync()
    ctx.profile(True)
    ctx.profile_reset()

    def fence():
        ctx.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    fence()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    gpu_ms = ctx.timer_stop()
    hist_z = hz.download((R1 + 1,), np.uint64)
    hist_x = hx.download((R2 + 1,), np.uint64)
    if world > 1:
        from quantum_css_codes_amd.montecarlo import all_reduce_histograms
        hist_z, hist_x = all_reduce_histograms([hist_z, hist_x])
    fence()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(hist_z.sum()) == world * args.steps * batch and int(hist_x.sum()) == world * args.steps * batch

    syn_ms, syn_launches = ctx.profile_get(_native.K_SYNDROME)
    hist_ms, hist_launches = ctx.profile_get(_native.K_HIST)
    ctx.profile(False)

    total = world * args.steps * batch
    value = total / elapsed
    out = None
    if rank == 0:
        # dominant kernel: one launch handles `batch` samples of one Pauli component
        alg_bytes = batch * alg_bytes_per_sample
        mean_launch_s = syn_ms / 1e3 / max(1, syn_launches)
        achieved = alg_bytes / mean_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(kernel_name + "_bytes_per_launch")
        out = {
            "metric": "syndromes/sec (n=4096 CSS)", "value": value, "unit": "syndromes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[4]: n=4096 CSS Monte-Carlo, random dual code (H1 2048x4096, H2 2047x4096, "
                                   "standard form), depolarising p=0.01, errors resident in HBM",
                       "algo": args.algo, "samples_per_gpu_per_step": batch, "global_samples_per_step": batch * world,
                       "parallelism": "sample-range shards, 1 histogram all-reduce"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_ms": mean_launch_s * 1e3,
                         "launches": syn_launches},
            "kernel_ms": {"syndrome": syn_ms, "histogram": hist_ms, "stream_total": gpu_ms},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(code)
        if args.extras:
            out["extras"] = extras(ctx, h1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def extras(ctx, h1):
    """RREF GB/s (2 * m * ld * 8 bytes / time) on the resident 2048 x 4096 matrix, single and
    batched, and the sampler-inclusive Monte-Carlo pipeline."""
    from quantum_css_codes_amd import _native
    import ctypes
    res = {}
    rng = np.random.default_rng(4096)
    a = _native.pack_rows(rng.integers(0, 2, (R1, N_QUBITS)).astype(np.uint8))
    for batch in (1, 64):
        buf = ctx.alloc(batch * a.nbytes)
        piv, rk = ctx.alloc(batch * R1 * 8), ctx.alloc(batch * 8)
        best = None
        for _ in range(3):
            for b in range(batch):
                _native.check(_native.lib().gf2_h2d(ctx.handle, buf.ptr + b * a.nbytes, a.ctypes.data, a.nbytes))
            ctx.timer_start()
            _native.check(_native.lib().gf2_rref_batch_dev(ctx.handle, buf.ptr, batch, R1, N_QUBITS, N_QUBITS // 64,
                                                           piv.ptr, rk.ptr))
            ms = ctx.timer_stop()
            best = ms if best is None else min(best, ms)
        res["rref_2048x4096_batch%d" % batch] = {"ms": best, "GB/s": batch * 2 * a.nbytes / best / 1e6,
                                                 "frac_hbm": batch * 2 * a.nbytes / best / 1e6 / HBM_PEAK_GBS}
        buf.free(), piv.free(), rk.free()
    return res


if __name__ == "__main__":
    main()
