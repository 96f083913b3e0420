"""world_size-2 CPU (gloo) test of the multi-GPU path: rank partitioning covers every pair exactly once,
the packed-weight broadcast delivers rank 0's blob bit-exactly, max-over-ranks timing and result
gathering work.  (The data path itself has no collective: pairs are independent.)"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from mdgat_matcher_amd import MDGAT, shard, synth, pack
    r, w, _ = shard.init_distributed(world, backend='gloo')
    L = 2
    net = MDGAT(synth.default_config(L=L, k=[]))
    if r == 0:
        net.load_state_dict(synth.make_state_dict(L=L, seed=5))
    blob = shard.broadcast_weights(net, 'cpu', r, w)
    first, count = shard.partition(11, r, w)
    local = torch.arange(first, first + count, dtype=torch.int64)
    t = shard.max_over_ranks(1.0 + r, 'cpu', w)
    first_e, count_e = shard.partition(12, r, w)
    gathered = shard.gather_matches(torch.arange(first_e, first_e + count_e, dtype=torch.int64)[:, None], w)
    shard.barrier(w)
    q.put((r, blob.numpy().copy(), local.numpy(), t, gathered.numpy()))
    shard.finalize(w)


def test_two_rank_gloo():
    world, port = 2, 29611 + (os.getpid() % 200)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from mdgat_matcher_amd import MDGAT, synth
    ref_net = MDGAT(synth.default_config(L=2, k=[]))
    ref_net.load_state_dict(synth.make_state_dict(L=2, seed=5))
    expect = ref_net.packed_weights()
    np.testing.assert_array_equal(res[0][1], expect)
    np.testing.assert_array_equal(res[1][1], expect)          # rank 1 got rank 0's blob bit-exactly
    assert sorted(np.concatenate([res[0][2], res[1][2]]).tolist()) == list(range(11))
    assert res[0][3] == res[1][3] == 2.0
    np.testing.assert_array_equal(res[0][4][:, 0], np.arange(12))


def test_partition_properties():
    sys.path.insert(0, ROOT)
    from mdgat_matcher_amd import shard
    for n in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            parts = [shard.partition(n, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
