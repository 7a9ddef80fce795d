"""
Monte-Carlo syndrome histograms [build-defined, SURVEY.md 8a x3 and 8e].

Every sample is independent and sample i is a pure function of (seed, i), so the global index range
[first, first + count) is cut into contiguous shards, one per GPU / process, with no exchange on the
data path.  The only collective is the final sum of the histograms: one all-reduce of at most
(r_1 + 1) + (r_2 + 1) uint64 bins (about 32 KiB at n = 4096) -- RCCL over xGMI when torch.distributed
runs on the "nccl" backend, gloo on CPU.  The result is identical for every number of shards.
"""
import numpy as np

from . import _native


def shard_range(first, count, rank, world):
    """Contiguous shard of [first, first + count) for `rank` of `world`: the first (count % world)
    ranks take one extra sample.  Returns (shard_first, shard_count)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(int(count), int(world))
    mine = base + (1 if rank < extra else 0)
    start = int(first) + rank * base + min(rank, extra)
    return start, mine


def pick_mode(r_1, r_2, mode=None):
    if mode is None:
        mode = 'full' if (r_1 <= 24 and r_2 <= 24) else 'weight'
    if mode not in ('full', 'weight'):
        raise ValueError("mode must be 'full' or 'weight'")
    if mode == 'full' and (r_1 > 24 or r_2 > 24):
        raise ValueError("full histograms need r_1, r_2 <= 24")
    return mode


def run_local(code, num_samples, p_x, p_y, p_z, seed=0, first_sample=0, mode=None):
    """Histograms of samples [first_sample, first_sample + num_samples) on this process's GPU."""
    mode = pick_mode(code.r_1, code.r_2, mode)
    chk1, chk2 = code._device_checks()
    ctx = _native.default_context()
    hist_z, hist_x = ctx.mc_run(chk1, chk2, int(seed), int(first_sample), int(num_samples), float(p_x), float(p_y),
                                float(p_z), _native.HIST_FULL if mode == 'full' else _native.HIST_WEIGHT)
    return {'hist_z': hist_z, 'hist_x': hist_x, 'mode': mode}


def rccl_comm(group=None, ctx=None):
    """A communicator of libgf2hip's own (gf2_comm_create over librccl) spanning the ranks of the torch.distributed group:
    rank 0 makes the id, the group's rendezvous carries its 128 bytes to the others (any backend; gloo needs no GPU
    runtime in torch).  The histogram all-reduce then runs on the compute context's stream from device memory, with no
    second GPU runtime in between.  None when no process group is initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        # the id, or the reason there is none (librccl does not load here): every rank must hear of it, or the others would
        # sit in the broadcast for ever
        try:
            box[0] = _native.Comm.unique_id()
        except _native.GF2Error as err:
            box[0] = (err.code, err.message)
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if not isinstance(box[0], bytes):
        raise _native.GF2Error(*box[0])                     # on every rank alike
    return _native.Comm(ctx or _native.default_context(), box[0], world, rank)


def all_reduce_histograms(hists, group=None, device=None, comm=None):
    """Sum a list of uint64 histograms over the ranks (one all-reduce of the concatenation).  With `comm` (a _native.Comm,
    see rccl_comm) the sum runs over libgf2hip's RCCL communicator; otherwise over torch.distributed -- RCCL on the
    "nccl" backend, gloo on CPU; counts stay below 2^63, so the int64 transport is exact.  With neither a communicator nor
    a process group the input is returned unchanged."""
    if comm is not None:
        return comm.allreduce_host(hists)
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [np.array(h, dtype=np.uint64) for h in hists]
    sizes = [int(h.size) for h in hists]
    flat = np.concatenate([np.asarray(h, dtype=np.uint64).view(np.int64) for h in hists])
    if device is None:
        # RCCL needs the buffer on THIS rank's GPU: the one the compute context runs on (GF2_DEVICE / LOCAL_RANK), not
        # torch's current device, which is cuda:0 in every rank unless the caller set it
        device = torch.device('cuda', _native.default_context().device) if dist.get_backend(group) == 'nccl' else 'cpu'
    buf = torch.from_numpy(flat.copy()).to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    total = buf.cpu().numpy().view(np.uint64)
    out, pos = [], 0
    for size in sizes:
        out.append(total[pos:pos + size].copy())
        pos += size
    return out


def run_sharded(code, num_samples, p_x, p_y, p_z, seed=0, first_sample=0, mode=None, group=None, local_fn=None, comm=None):
    """This rank's shard of the global range, then the histogram all-reduce (over `comm` when given, see
    all_reduce_histograms).  `local_fn` (same signature as run_local) replaces the GPU computation; the CPU tests of the
    sharding use it."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    start, mine = shard_range(first_sample, num_samples, rank, world)
    fn = local_fn or run_local
    part = fn(code, mine, p_x, p_y, p_z, seed=seed, first_sample=start, mode=mode)
    hist_z, hist_x = all_reduce_histograms([part['hist_z'], part['hist_x']], group=group, comm=comm)
    return {'hist_z': hist_z, 'hist_x': hist_x, 'mode': part['mode'], 'shard': (start, mine)}


DECODE_FIELDS = ('logical_x', 'logical_z', 'logical_any', 'uncorrectable_x', 'uncorrectable_z')


def dense_table(table, r, n):
    """A syndrome table (dict: vec_to_int(syndrome) -> error vector, css_code.py:715-735) as 2^r packed words
    indexed by the key; entries the table does not have are ~0."""
    out = np.full(1 << r, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    if len(table):
        keys = np.fromiter((int(key) for key in table), dtype=np.int64, count=len(table))
        errs = np.array(list(table.values()), dtype=np.uint8).reshape(len(table), n)
        out[keys] = _native.pack_rows(errs)[:, 0]                    # (n <= 63: one word per error)
    return out


def packed_word(vec):
    word = 0
    for j in np.flatnonzero(np.asarray(vec)):
        word |= 1 << int(j)
    return word


def table_entries(table, r, n):
    """A syndrome table (dict: vec_to_int(syndrome) -> error vector) as the arrays gf2_mc_decode_hashed takes: keys (entries x 1
    word for r <= 63, x 2 words, low first, beyond) and the packed errors (entries x 2 words, n <= 128)."""
    kw = 1 if r <= 63 else 2
    keys = np.zeros((len(table), kw), dtype=np.uint64)
    mask = (1 << 64) - 1
    if kw == 1:
        keys[:, 0] = np.fromiter((int(key) for key in table), dtype=np.uint64, count=len(table))
    else:
        for i, key in enumerate(table):
            key = int(key)
            keys[i, 0] = key & mask
            keys[i, 1] = key >> 64
    errs = np.array(list(table.values()), dtype=np.uint8).reshape(len(table), n)
    corr = np.zeros((len(table), 2), dtype=np.uint64)
    packed = _native.pack_rows(errs) if len(table) else np.zeros((0, 1), dtype=np.uint64)
    corr[:, :packed.shape[1]] = packed[:, :2]
    return keys, corr


def decode_local(code, num_samples, p_x, p_y, p_z, seed=0, first_sample=0, hashed=None):
    """Table decode + logical-error tally of samples [first_sample, first_sample + num_samples) on this GPU: gf2_mc_decode
    (dense tables of 2^r words) for n <= 63 and r_1, r_2 <= 20, gf2_mc_decode_hashed (the tables' entries in hash tables on the
    device) for every other code of at most 128 qubits.  Returns a dict of counts (DECODE_FIELDS) plus 'samples'."""
    if code.n > 128 or max(code.r_1, code.r_2) > 127 or min(code.r_1, code.r_2) < 1:
        raise ValueError("table decode needs n <= 128 and 1 <= r_1, r_2 <= 127")
    ctx = _native.default_context()
    if hashed or code.n > 63 or code.r_1 > 20 or code.r_2 > 20:      # (hashed=True: the hash-table kernel for a small code too)
        two = lambda vec: np.pad(_native.pack_rows(np.asarray(vec).reshape(1, -1))[0], (0, 2))[:2]
        # (the tables as arrays: made once per code object -- 0.1 s of Python for a table of 350 000 entries)
        cached = getattr(code, "_hashed_table_arrays", None)
        if cached is None or cached[0] is not code._c1_syndromes or cached[1] is not code._c2_syndromes:
            cached = (code._c1_syndromes, code._c2_syndromes, table_entries(code._c1_syndromes, code.r_1, code.n),
                      table_entries(code._c2_syndromes, code.r_2, code.n))
            code._hashed_table_arrays = cached
        (keys1, corr1), (keys2, corr2) = cached[2], cached[3]
        counts = ctx.mc_decode_hashed(code.n, _native.pack_rows(code.parity_check_c1), code.r_1, keys1, corr1,
                                      _native.pack_rows(code.parity_check_c2), code.r_2, keys2, corr2,
                                      two(code.x_operator_matrix()[0]), two(code.z_operator_matrix()[0]),
                                      int(seed), int(first_sample), int(num_samples), float(p_x), float(p_y), float(p_z))
        out = {name: int(v) for name, v in zip(DECODE_FIELDS, counts)}
        out['samples'] = int(num_samples)
        return out
    chk1, chk2 = code._device_checks()
    cached = getattr(code, "_dense_table_arrays", None)             # (made once per code object, like the hash tables' arrays)
    if cached is None or cached[0] is not code._c1_syndromes or cached[1] is not code._c2_syndromes:
        cached = (code._c1_syndromes, code._c2_syndromes, dense_table(code._c1_syndromes, code.r_1, code.n),
                  dense_table(code._c2_syndromes, code.r_2, code.n))
        code._dense_table_arrays = cached
    counts = ctx.mc_decode(chk1, chk2, cached[2], cached[3],
                           packed_word(code.x_operator_matrix()[0]), packed_word(code.z_operator_matrix()[0]),
                           int(seed), int(first_sample), int(num_samples), float(p_x), float(p_y), float(p_z))
    out = {name: int(v) for name, v in zip(DECODE_FIELDS, counts)}
    out['samples'] = int(num_samples)
    return out


def decode_sharded(code, num_samples, p_x, p_y, p_z, seed=0, first_sample=0, group=None, local_fn=None):
    """This rank's shard of the global range, then one all-reduce of the five counts."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    start, mine = shard_range(first_sample, num_samples, rank, world)
    part = (local_fn or decode_local)(code, mine, p_x, p_y, p_z, seed=seed, first_sample=start)
    total, = all_reduce_histograms([np.array([part[f] for f in DECODE_FIELDS], dtype=np.uint64)], group=group)
    out = {name: int(v) for name, v in zip(DECODE_FIELDS, total)}
    out['samples'] = int(num_samples)
    return out
