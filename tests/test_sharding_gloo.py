"""
The N > 1 path on CPU: world_size-2 torch.distributed (gloo) processes each take their shard of the global
sample range and all-reduce the histograms; the result must equal the unsharded histogram.  The GPU kernel
is replaced by the C oracle here (run_sharded's local_fn hook) -- what is under test is the sharding
arithmetic and the collective, which are the same code on RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from quantum_css_codes_amd.montecarlo import shard_range
    for (first, count, world) in ((0, 10, 3), (5, 100, 8), (7, 3, 4), (0, 0, 2), (11, 10**8, 8)):
        pos, total = first, 0
        for rank in range(world):
            start, mine = shard_range(first, count, rank, world)
            assert start == pos and mine >= 0
            pos += mine
            total += mine
        assert total == count
    with pytest.raises(ValueError):
        shard_range(0, 10, 2, 2)


def test_decode_table_arrays():
    # the syndrome tables as the arrays the decode kernels take (host only): dense tables of 2^r words (n <= 63) and the hash
    # tables' key / error arrays, with one-word keys up to 63 checks and two-word keys beyond
    from quantum_css_codes_amd.montecarlo import dense_table, table_entries
    rng = np.random.default_rng(8)
    n, r = 40, 9
    errs = [np.zeros(n, dtype=int)] + [(rng.random(n) < 0.1).astype(int) for _ in range(30)]
    keys = [0] + [int(k) for k in rng.choice(np.arange(1, 1 << r), 30, replace=False)]
    table = dict(zip(keys, errs))
    dense = dense_table(table, r, n)
    assert dense.shape == (1 << r,) and dense.dtype == np.uint64
    for key in range(1 << r):
        want = sum(1 << j for j in np.flatnonzero(table[key])) if key in table else 0xFFFFFFFFFFFFFFFF
        assert int(dense[key]) == want
    n, r = 100, 70
    big_keys = [0, 5, (1 << 64) + 3, (1 << 69) | 1]
    big_errs = [(rng.random(n) < 0.05).astype(int) for _ in big_keys]
    k, c = table_entries(dict(zip(big_keys, big_errs)), r, n)
    assert k.shape == (4, 2) and c.shape == (4, 2)
    for i, key in enumerate(big_keys):
        assert int(k[i, 0]) | (int(k[i, 1]) << 64) == key
        assert int(c[i, 0]) | (int(c[i, 1]) << 64) == sum(1 << int(j) for j in np.flatnonzero(big_errs[i]))
    k1, c1 = table_entries({7: big_errs[0][:50], 9: big_errs[1][:50]}, 40, 50)
    assert k1.shape == (2, 1) and [int(v) for v in k1[:, 0]] == [7, 9] and int(c1[0, 1]) == 0
    assert table_entries({}, 30, 50)[0].shape == (0, 1)


def test_pick_mode():
    from quantum_css_codes_amd.montecarlo import pick_mode
    assert pick_mode(3, 3) == 'full' and pick_mode(2048, 2047) == 'weight' and pick_mode(3, 3, 'weight') == 'weight'
    with pytest.raises(ValueError):
        pick_mode(30, 3, 'full')
    with pytest.raises(ValueError):
        pick_mode(3, 3, 'other')


class _Code(object):
    def __init__(self, h1, h2):
        self.parity_check_c1, self.parity_check_c2 = h1, h2
        self.r_1, self.r_2, self.n = h1.shape[0], h2.shape[0], h1.shape[1]


def _oracle_local(code, num_samples, p_x, p_y, p_z, seed=0, first_sample=0, mode=None):
    from oracle import c_oracle
    from quantum_css_codes_amd.montecarlo import pick_mode
    mode = pick_mode(code.r_1, code.r_2, mode)
    hz, hx = c_oracle.mc(c_oracle.pack_rows(code.parity_check_c1), code.r_1, c_oracle.pack_rows(code.parity_check_c2),
                         code.r_2, code.n, seed, first_sample, num_samples, p_x, p_y, p_z, 0 if mode == 'full' else 1)
    return {'hist_z': hz, 'hist_x': hx, 'mode': mode}


def _oracle_decode_local(code, num_samples, p_x, p_y, p_z, seed=0, first_sample=0):
    from oracle import c_oracle
    from quantum_css_codes_amd.montecarlo import DECODE_FIELDS, dense_table, packed_word
    counts = c_oracle.mc_decode(c_oracle.pack_rows(code.parity_check_c1), code.r_1, c_oracle.pack_rows(code.parity_check_c2),
                                code.r_2, code.n, dense_table(code._c1_syndromes, code.r_1, code.n),
                                dense_table(code._c2_syndromes, code.r_2, code.n),
                                packed_word(code.x_operator_matrix()[0]), packed_word(code.z_operator_matrix()[0]),
                                seed, first_sample, num_samples, p_x, p_y, p_z)
    out = {name: int(v) for name, v in zip(DECODE_FIELDS, counts)}
    out['samples'] = num_samples
    return out


def _steane_oracle_code():
    from oracle import cpu_ref
    h = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])
    return cpu_ref.CSSCode(h, h)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from quantum_css_codes_amd import montecarlo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1)
    code = _Code(rng.integers(0, 2, (5, 70)), rng.integers(0, 2, (66, 70)))
    res = montecarlo.run_sharded(code, 5001, 0.05, 0.02, 0.03, seed=9, first_sample=100, mode='weight',
                                 local_fn=_oracle_local)
    dec = montecarlo.decode_sharded(_steane_oracle_code(), 30001, 0.05, 0.02, 0.03, seed=4, first_sample=7,
                                    local_fn=_oracle_decode_local)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), hist_z=res['hist_z'], hist_x=res['hist_x'],
             shard=np.array(res['shard']), decode=np.array([dec[f] for f in montecarlo.DECODE_FIELDS]))
    dist.destroy_process_group()


def test_two_rank_histogram_allreduce(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(1)
    code = _Code(rng.integers(0, 2, (5, 70)), rng.integers(0, 2, (66, 70)))
    whole = _oracle_local(code, 5001, 0.05, 0.02, 0.03, seed=9, first_sample=100, mode='weight')
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert list(r0["shard"]) == [100, 2501] and list(r1["shard"]) == [2601, 2500]
    for r in (r0, r1):                                    # every rank holds the global histogram
        assert np.array_equal(r["hist_z"], whole['hist_z']) and np.array_equal(r["hist_x"], whole['hist_x'])
    assert int(whole['hist_z'].sum()) == 5001
    from quantum_css_codes_amd.montecarlo import DECODE_FIELDS
    whole_dec = _oracle_decode_local(_steane_oracle_code(), 30001, 0.05, 0.02, 0.03, seed=4, first_sample=7)
    for r in (r0, r1):
        assert list(r["decode"]) == [whole_dec[f] for f in DECODE_FIELDS]


def test_eight_rank_histogram_allreduce(tmp_path):
    # the world size of the target node (8 GPUs, BASELINE.json configs[4]) rehearsed on the CPU: eight gloo ranks, the same
    # sharding arithmetic and the same collective call as on RCCL; every rank ends with the one-rank histogram bin for bin
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    rng = np.random.default_rng(1)
    code = _Code(rng.integers(0, 2, (5, 70)), rng.integers(0, 2, (66, 70)))
    whole = _oracle_local(code, 5001, 0.05, 0.02, 0.03, seed=9, first_sample=100, mode='weight')
    from quantum_css_codes_amd.montecarlo import DECODE_FIELDS, shard_range
    whole_dec = _oracle_decode_local(_steane_oracle_code(), 30001, 0.05, 0.02, 0.03, seed=4, first_sample=7)
    for rank in range(8):
        r = np.load(tmp_path / ("rank%d.npz" % rank))
        assert tuple(r["shard"]) == shard_range(100, 5001, rank, 8)
        assert np.array_equal(r["hist_z"], whole['hist_z']) and np.array_equal(r["hist_x"], whole['hist_x'])
        assert list(r["decode"]) == [whole_dec[f] for f in DECODE_FIELDS]
