"""
The one collective of the path: the sum of the ranks' histograms over RCCL.

CPU part: librccl loads through the C ABI (gf2_rccl_version) and the communicator entry points refuse bad arguments without
a GPU.  GPU part (one MI355X): RCCL runs for real with one rank -- gf2_comm_create_all / gf2_comm_create + gf2_hist_allreduce in
this process, torch.distributed's "nccl" backend in a fresh child process under montecarlo.run_sharded -- and bench.py starts its
own ranks when no launcher did.  More than one rank over RCCL needs more than one GPU: the driver's scaling run.
"""
import json
import os
import socket
import subprocess
import time
import sys

import numpy as np
import pytest

from quantum_css_codes_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_loads_through_the_abi():
    try:
        version = _native.rccl_version()
    except _native.GF2Error as err:                        # a CPU box without ROCm's librccl: nothing to load
        if err.code == _native.GF2_E_RCCL:
            pytest.skip("librccl cannot be loaded here: %s" % err.message)
        raise
    assert version >= 20000                                # NCCL-style version number, e.g. 22707
    with open("/proc/self/maps") as maps:
        assert "librccl" in maps.read()


def test_comm_argument_errors_without_a_gpu():
    lib = _native.lib()
    import ctypes
    out = ctypes.c_void_p()
    assert lib.gf2_comm_unique_id(None, 128) == _native.GF2_E_ARG
    small = ctypes.create_string_buffer(64)
    assert lib.gf2_comm_unique_id(small, 64) == _native.GF2_E_ARG
    assert lib.gf2_comm_create(None, small, 1, 0, ctypes.byref(out)) == _native.GF2_E_ARG
    assert lib.gf2_comm_create_all(None, 1, ctypes.byref(out)) == _native.GF2_E_ARG
    assert lib.gf2_hist_allreduce(None, None, 8) == _native.GF2_E_ARG
    assert lib.gf2_comm_destroy(None) == _native.GF2_OK
    assert b"gf2_hist_allreduce" in lib.gf2_last_error() or b"bad argument" in lib.gf2_last_error()


@pytest.mark.gpu
def test_hist_allreduce_one_rank_in_this_process():
    # RCCL initialises on the device, all-reduces the 32 KiB of configs[4]'s two weight histograms in place on the context's
    # stream; with one rank the sum is the input
    ctx = _native.default_context()
    nbins = 2049 + 2048
    rng = np.random.default_rng(3)
    bins = rng.integers(0, 2**62, nbins, dtype=np.int64).view(np.uint64)
    for make in (lambda: _native.Comm.all_local([ctx]), lambda: _native.Comm(ctx, _native.Comm.unique_id(), 1, 0)):
        comm = make()
        buf = ctx.alloc(nbins * 8).upload(bins)
        comm.allreduce(buf, nbins)
        comm.allreduce(buf, nbins)
        assert np.array_equal(buf.download((nbins,), np.uint64), bins)
        parts = comm.allreduce_host([bins[:2049], bins[2049:]])
        assert np.array_equal(parts[0], bins[:2049]) and np.array_equal(parts[1], bins[2049:])
        buf.free()
        comm.close()
    with open("/proc/self/maps") as maps:
        assert "librccl" in maps.read()
    with pytest.raises(_native.GF2Error):
        _native.Comm.all_local([ctx, ctx])                  # one communicator cannot hold a device twice


NCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from quantum_css_codes_amd import _native, montecarlo
from quantum_css_codes_amd.css_code import CSSCode
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
h = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
code = CSSCode(h, h)
want = montecarlo.run_local(code, 200001, 0.02, 0.01, 0.03, seed=5, first_sample=17)
# torch.distributed's all_reduce on the nccl backend (= RCCL), the buffer on the compute context's device
got = montecarlo.run_sharded(code, 200001, 0.02, 0.01, 0.03, seed=5, first_sample=17)
assert got['shard'] == (17, 200001)
assert np.array_equal(got['hist_z'], want['hist_z']) and np.array_equal(got['hist_x'], want['hist_x'])
# libgf2hip's own communicator, its id carried by the process group
comm = montecarlo.rccl_comm()
again = montecarlo.run_sharded(code, 200001, 0.02, 0.01, 0.03, seed=5, first_sample=17, comm=comm)
assert np.array_equal(again['hist_z'], want['hist_z']) and np.array_equal(again['hist_x'], want['hist_x'])
big = np.arange(4097, dtype=np.uint64) * np.uint64(1 << 40)            # 32 KiB, values beyond 2^32
assert np.array_equal(montecarlo.all_reduce_histograms([big])[0], big)
assert np.array_equal(montecarlo.all_reduce_histograms([big], comm=comm)[0], big)
comm.close()
# bench.py's way in: the communicator proven by one all-reduce under a time limit, the ranks agreeing on the outcome; and
# its way out when the library's communicator cannot be had
import bench
ctx = _native.default_context()
comm, note = bench.guarded_comm(ctx, 4097)
assert comm is not None and note == "", note
comm.close()
class Broken(_native.Comm):
    def __init__(self, *a):
        raise _native.GF2Error(-7, "ncclCommInitRank refused")
keep, _native.Comm = _native.Comm, Broken
comm, note = bench.guarded_comm(ctx, 4097)
_native.Comm = keep
assert comm is None and "refused" in note, note
if os.environ.get("CHILD_STUCK_COMM"):
    # the TIMEOUT branch: a communicator that does not come up within the limit.  The helper thread is then inside RCCL for good, so
    # guarded_comm must not return: reason on stderr, flush, os._exit(3) -- nothing below runs
    import time
    class Stuck(_native.Comm):
        def __init__(self, *a):
            time.sleep(3600)
    _native.Comm = Stuck
    print("about to hang", flush=True)
    bench.guarded_comm(ctx, 4097, seconds=2.0)
    print("guarded_comm returned from a stuck communicator")
    sys.exit(0)
dist.destroy_process_group()
maps = open("/proc/self/maps").read()
assert "librccl" in maps and "libgf2hip.so" in maps
print("nccl child ok", int(got['hist_z'].sum()))
"""


@pytest.mark.gpu
def test_world_size_one_nccl_process_group_in_a_child_process(tmp_path):
    script = tmp_path / "nccl_child.py"
    script.write_text(NCCL_CHILD % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), GF2_DEVICE="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    done = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    assert "nccl child ok 200001" in done.stdout


@pytest.mark.gpu
def test_guarded_comm_leaves_the_process_when_the_communicator_never_comes_up(tmp_path):
    # on timeout the rank must END within a bound (the helper thread stays inside RCCL), by an exit, with
    # what it had printed flushed
    script = tmp_path / "nccl_child_stuck.py"
    script.write_text(NCCL_CHILD % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), GF2_DEVICE="0", CHILD_STUCK_COMM="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.monotonic()
    done = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=600, text=True)
    assert done.returncode == 75, (done.returncode, done.stderr[-3000:])       # bench.COMM_TIMEOUT_EXIT
    assert "about to hang" in done.stdout and "returned from a stuck" not in done.stdout
    assert "not back within 2 s" in done.stderr
    assert time.monotonic() - t0 < 300


@pytest.mark.gpu
def test_bench_with_its_own_communicator_and_without_when_it_never_comes_up():
    # `--allreduce gf2` with the one rank a single GPU has: the supervising launcher starts the rank, the communicator is proven and
    # used for the histogram sum.  Then the same with a communicator that hangs (BENCH_FAULT): the rank leaves with code 75 after
    # --comm-timeout, the launcher starts a fresh one with --allreduce torch, ONE line comes out and the exit code is 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GF2_DEVICE"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--allreduce", "gf2", "--comm-timeout", "3", "--steps", "2",
           "--warmup", "1", "--batch-log2", "17", "--no-cpu-baseline", "--no-secondary", "--no-settle"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    assert bench_line(done)["checks"]["histogram_total"] == 2 << 17
    t0 = time.monotonic()
    done = subprocess.run(cmd, cwd=ROOT, env=dict(env, BENCH_FAULT="comm_timeout"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    assert bench_line(done)["checks"]["histogram_total"] == 2 << 17
    assert "not back within 3 s" in done.stderr and "fresh ranks with --allreduce torch" in done.stderr
    assert time.monotonic() - t0 < 300


def children_of(pid):
    psutil = pytest.importorskip("psutil")                  # (not a dependency of the package: without it the test is skipped)
    try:
        return psutil.Process(pid).children(recursive=True)
    except psutil.NoSuchProcess:
        return []


@pytest.mark.parametrize("how", ["sigterm", "deadline"])
def test_self_launched_ranks_do_not_outlive_the_launcher(how):
    # ranks that ignore SIGTERM (BENCH_FAULT=hang: what a rank inside RCCL looks like) must be gone when the launcher is --
    # whether it is told to stop or its own deadline passes (no GPU needed: the ranks hang before they touch one)
    import signal
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_FAULT"] = "hang"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + (["--launch-deadline", "2"] if how == "deadline" else [])
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kids = []
    for _ in range(100):
        time.sleep(0.1)
        kids = children_of(proc.pid)
        if len(kids) >= 2:
            break
    assert len(kids) >= 2, "the launcher did not start its two ranks"
    if how == "sigterm":
        time.sleep(0.5)
        proc.send_signal(signal.SIGTERM)
    try:
        _, err = proc.communicate(timeout=60)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == (128 + signal.SIGTERM if how == "sigterm" else 124), (proc.returncode, err[-2000:])
    for kid in kids:
        assert not kid.is_running() or kid.status() == "zombie", "rank %d outlived the launcher" % kid.pid


def bench_line(done):
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_when_no_launcher_did():
    # `python bench.py --gpus 2` with WORLD_SIZE unset: two child ranks started before anything touches the GPU (they share this
    # box's one GPU, hence gloo; the nccl default needs one GPU per rank)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GF2_DEVICE"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch-log2", "17", "--no-cpu-baseline", "--no-secondary", "--no-settle"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    out = bench_line(done)
    assert out["n_gpus"] == 2 and out["config"]["global_samples_per_step"] == 2 << 17
    assert out["checks"]["histogram_total"] == 2 * (2 << 17)


@pytest.mark.gpu
def test_bench_one_rank_under_the_launcher_uses_the_nccl_backend():
    # the driver's launch line with N = 1: torch.distributed.run, nccl backend; a world of one has no process group, and the line
    # is the plain one
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch-log2", "17", "--no-cpu-baseline", "--no-secondary", "--no-settle"]
    done = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    out = bench_line(done)
    assert out["n_gpus"] == 1 and out["checks"]["histogram_total"] == 2 << 17


@pytest.mark.gpu
def test_bench_total_samples_is_strong_scaling():
    # --total-samples T: T samples in all, rank g of N taking the g-th of N contiguous shards (configs[4] to the letter is T = 10^8);
    # one rank, then two ranks sharing this box's GPU over gloo with a T that does not divide evenly: the histogram totals say that
    # every sample was counted once per step
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GF2_DEVICE"] = "0"
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-settle"]
    done = subprocess.run(base + ["--gpus", "1", "--total-samples", "300000"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    out = bench_line(done)
    assert out["scaling"] == "strong" and out["checks"]["histogram_total"] == 2 * 300000
    assert out["config"]["global_samples_per_step"] == 300000 and out["config"]["samples_per_gpu_per_step"] == 300000
    done = subprocess.run(base + ["--gpus", "2", "--dist-backend", "gloo", "--total-samples", "300001"], cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    out = bench_line(done)
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["checks"]["histogram_total"] == 2 * 300001
    assert out["config"]["global_samples_per_step"] == 300001 and out["config"]["samples_per_gpu_per_step"] == 150001


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=120, text=True)
    assert done.returncode != 0 and "WORLD_SIZE=3" in done.stderr


def test_bench_self_launch_reports_a_failing_rank():
    # `python bench.py --gpus 2` without a launcher starts its two ranks itself; where they cannot run (no GPU in the CPU container: the
    # first compute call raises) the launcher must come back with a non-zero code instead of waiting for the other rank
    if _native.device_count() > 0:
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1",
                           "--warmup", "0", "--batch-log2", "15", "--no-cpu-baseline", "--no-secondary", "--no-settle"],
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert done.returncode != 0
    assert not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
