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


def _worker_f64(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from mdgat_matcher_amd import MDGAT, shard, synth
    r, w, _ = shard.init_distributed(world, backend='gloo')
    # (no 'arithmetic' key: the float64 module is the request, as for the reference's callers; rank 1 stays float32 until after the
    # broadcast - a module whose arithmetic follows its dtype receives both blobs either way)
    net = MDGAT(synth.default_config(L=2))
    if r == 0:
        net = net.double()
        net.load_state_dict(synth.make_state_dict(L=2, seed=5))
    got64 = []
    blob = shard.broadcast_weights(net, 'cpu', r, w, out64=got64)
    shard.barrier(w)
    q.put((r, blob.numpy().copy(), got64[0].numpy().copy()))
    shard.finalize(w)


def test_two_rank_gloo_exact_mode_broadcasts_both_blobs():
    """The exact mode (float64 module / arithmetic='fp64' / a module that may be cast later): the fp64 blob (the folded weights before their rounding) travels next to the fp32 one, and a rank that
    never loaded a checkpoint receives rank 0's, bit for bit."""
    world, port = 2, 29711 + (os.getpid() % 200)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_f64, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from mdgat_matcher_amd import MDGAT, synth
    ref = MDGAT(synth.default_config(L=2, arithmetic='fp64')).double()
    ref.load_state_dict(synth.make_state_dict(L=2, seed=5))
    e32, e64 = ref.packed_weights(), ref.packed_weights(np.float64)
    for r in range(2):
        np.testing.assert_array_equal(res[r][1], e32)
        np.testing.assert_array_equal(res[r][2], e64)
        assert res[r][2].dtype == np.float64
    np.testing.assert_array_equal(e64.astype(np.float32), e32)


def _install_worker(rank, world, port, q):
    """The per-rank control flow of bench.py (init -> rank 0 loads the checkpoint -> broadcast of the packed blob ->
    load_packed -> net.double().eval() -> forward on the rank's shard) with the HIP library replaced at the _lib
    boundary by a recorder, and 'is on the GPU' answered with yes: what is exercised is the host logic that decides
    WHICH weights reach mdgat_load_weights on a rank that never loaded a checkpoint."""
    import contextlib
    import ctypes as C
    from unittest import mock
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from mdgat_matcher_amd import MDGAT, _lib, pack, shard, synth

    class FakeLib:
        def __init__(self):
            self.calls, self.loaded = [], None
        def mdgat_create(self, cfg, idx, handle):
            self.calls.append(('create', idx))
            handle._obj.value = 4242                      # (handle is ctypes.byref(c_void_p()))
            return 0
        def mdgat_blob_floats(self, L):
            return pack.blob_layout(L)['total']
        def mdgat_load_weights(self, handle, ptr, n, on_device):
            addr = ptr.value if hasattr(ptr, 'value') else C.cast(ptr, C.c_void_p).value
            self.loaded = np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_float)), shape=(n,)).copy()
            self.calls.append(('load', int(n), int(on_device)))
            return 0
        def mdgat_workspace_bytes(self, handle, B, N, M):
            return 256
        def mdgat_forward(self, *a):
            self.calls.append(('forward', a[1], a[2], a[3]))
            return 0
        def mdgat_destroy(self, handle):
            self.calls.append(('destroy',))
        def mdgat_last_error(self):
            return b''

    fake = FakeLib()
    stream = mock.Mock(cuda_stream=0)
    r, w, _ = shard.init_distributed(world, backend='gloo')
    L = 2
    torch.manual_seed(100 + r)                       # every rank starts from its own random init
    # (arithmetic='fp32': the throughput path of a sharded job, as bench.py pins it - the casts of test.py:193 then change nothing)
    net = MDGAT(synth.default_config(L=L, k=[16, None], arithmetic='fp32')).eval()
    if r == 0:
        net.load_state_dict(synth.make_state_dict(L=L, seed=5, dtype=torch.float32))
    blob = shard.broadcast_weights(net, 'cpu', r, w)  # gloo broadcast of rank 0's packed blob (a CPU tensor: not installed)
    first, count = shard.partition(6, r, w)
    d = synth.make_batch(count, 40, 48, first_pair=first)
    # from here on the tensors pose as device memory (only around the calls under test: gloo must not see it)
    with mock.patch.object(_lib, 'load', lambda: fake), \
            mock.patch.object(torch.Tensor, 'is_cuda', new=property(lambda self: True)), \
            mock.patch.object(torch.cuda, 'current_device', lambda: 0), \
            mock.patch.object(torch.cuda, 'device', lambda d: contextlib.nullcontext()), \
            mock.patch.object(torch.cuda, 'current_stream', lambda d=None: stream):
        net.load_packed(blob)                            # what broadcast_weights does with a blob on the GPU
        net.double().eval()                              # test.py:193, before every forward
        net._run(d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
        net.double().eval()
        net._run(d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
        net._invalidate()                                # the recorder's handle must not reach the real mdgat_destroy
    shard.barrier(w)
    q.put((r, fake.calls, fake.loaded))
    shard.finalize(w)


def test_two_rank_weight_install_with_stubbed_library():
    world, port = 2, 29411 + (os.getpid() % 200)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_install_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from mdgat_matcher_amd import MDGAT, pack, synth
    ref_net = MDGAT(synth.default_config(L=2, k=[16, None]))
    ref_net.load_state_dict(synth.make_state_dict(L=2, seed=5, dtype=torch.float32))
    expect = ref_net.packed_weights()
    n = pack.blob_layout(2)['total']
    for r, calls, loaded in res:
        # exactly one handle and ONE weight install per rank - the broadcast blob, handed over as device memory - and
        # the casts before the forwards neither re-pack the rank's own (random) parameters nor rebuild the handle
        assert [c for c in calls if c[0] in ('create', 'load', 'destroy')] == [('create', 0), ('load', n, 1), ('destroy',)], (r, calls)
        assert [c for c in calls if c[0] == 'forward'] == [('forward', 3, 40, 48)] * 2
        np.testing.assert_array_equal(loaded, expect)


def test_partition_properties():
    sys.path.insert(0, ROOT)
    from mdgat_matcher_amd import shard
    for n in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            parts = [shard.partition(n, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def _one_rank_worker(port, q):
    """A process started by a launcher with ONE rank (torch.distributed.run --nproc-per-node 1): the process group exists and
    the helpers run their collectives through it - the CPU twin of tests/test_gpu_rccl.py (there: nccl on device tensors)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from mdgat_matcher_amd import MDGAT, shard, synth
    r, w, _ = shard.init_distributed(1, backend='gloo')
    assert dist.is_initialized() and w == 1
    net = MDGAT(synth.default_config(L=1, k=[]))
    net.load_state_dict(synth.make_state_dict(L=1, seed=5))
    blob = shard.broadcast_weights(net, 'cpu', r, w)
    ok = blob is not None and np.array_equal(blob.numpy(), net.packed_weights())
    shard.barrier(w)
    t = shard.max_over_ranks(2.5, 'cpu', w)
    g = shard.gather_matches(torch.arange(6)[:, None], w)
    shard.finalize(w)
    q.put((ok, t, g[:, 0].tolist(), dist.is_initialized()))


def test_one_rank_launched_process_runs_the_collectives():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(29811 + (os.getpid() % 100), q))
    p.start()
    ok, t, g, still = q.get(timeout=180)
    p.join(60)
    assert p.exitcode == 0 and ok and t == 2.5 and g == list(range(6)) and not still


def test_share_device_hook_is_announced_and_needs_a_single_gpu(capsys, monkeypatch):
    sys.path.insert(0, ROOT)
    from mdgat_matcher_amd import shard
    for k in ('RANK', 'MASTER_ADDR', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('MDGAT_SHARE_DEVICE', '1')
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert shard.init_distributed(1) == (0, 1, 0)                 # every rank on device 0 ...
    assert 'share_device test hook active' in capsys.readouterr().err     # ... and it says so
    monkeypatch.setattr('torch.cuda.is_available', lambda: True)
    monkeypatch.setattr('torch.cuda.device_count', lambda: 8)
    import pytest
    with pytest.raises(RuntimeError, match='single-GPU test hook'):
        shard.init_distributed(1)


def test_bench_eight_ranks_gloo_stub(tmp_path):
    """bench.py --gpus 8 --config 3 the way the driver launches it (torch.distributed.run, 8 ranks) - on CPU: gloo stands for
    RCCL and a recorder for the library (tests/bench_stub_runner.py), everything else is bench.py's own code: BASELINE
    configs[3] = 4096 pairs sharded 8-way = 512 per rank, ONE JSON line from rank 0, value = all pairs / the slowest rank's
    window, the communicator's world size and every rank's pair count and own time in the line."""
    import json
    import subprocess
    port = 29611 + (os.getpid() % 300)
    log = str(tmp_path / 'calls')
    env = dict(os.environ, MDGAT_BENCH_STUB_LOG=log, OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'bench_stub_runner.py'), '--gpus', '8', '--config', '3',
           '--steps', '3', '--warmup', '1', '--windows', '2']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout                                     # rank 0 only
    d = json.loads(lines[0])
    assert d['stub'] is True and d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['steps'] == 3
    c = d['config']
    assert c['rccl_world_size'] == 8 and c['collectives'] == 'gloo' and c['pairs_per_rank'] == [512] * 8 and c['baseline_config'] == 3
    t = d['timing']
    assert len(t['per_rank_ms_per_step']) == 8 and max(range(8), key=lambda r: t['per_rank_ms_per_step'][r]) == 3     # the straggler
    assert abs(d['value'] - 4096 * 3 / (sorted(t['window_ms'])[len(t['window_ms']) // 2] * 1e-3)) < 1e-3 * d['value']      # (window_ms is rounded)
    assert d['ms_per_step'] >= max(t['per_rank_ms_per_step']) * 0.999        # the whole job waits for its slowest rank
    assert d['status']['sinkhorn_fallback'] is False and d['status']['range_violation'] is False
    # the parity fields are COMPUTED in the run (here on the recorder's zeros: presence and shape, not values)
    pm = d['parity']['modes']
    assert set(pm) == {'fp32', 'fp64'} and 'cfg_n512_L9_S100' in d['parity']['fixture']
    for mode in pm.values():       # (the 40-pair batch the reference ran as one: tests/golden/cfg_n512_L9_S100_b40.npz)
        assert mode['pairs'] == 40 and 0 <= mode['pairs_within_1e-4'] <= 40 and isinstance(mode['matches_identical'], bool)
        assert isinstance(mode['max_abs_dZ'], float) and isinstance(mode['max_abs_d_mscores'], float)
        assert mode['arg_maxes'] == 40 * 1024 and mode['matches_differing'] >= 0
    assert d['status']['literal_1e-4_pairs'] == {m: f"{pm[m]['pairs_within_1e-4']}/40" for m in pm} and 'measured' in d['status']['parity_source']
    # every rank ran its own 512 pairs, and only those: 8 + warmup + windows x steps forwards of 512 x 512 x 512
    for r in range(8):
        calls = open(f'{log}.{r}').read().split('\n')[:-1]
        timed = calls[:8 + 1 + 2 * 3]
        assert timed and set(timed) == {'512 512 512'} and len(timed) == 8 + 1 + 2 * 3, (r, len(calls))
        # behind them, on rank 0 only: the parity block's two forwards over the 40 reference-held pairs (one per arithmetic mode)
        assert calls[len(timed):] == (['40 512 512'] * 2 if r == 0 else []), (r, calls[len(timed):])


def test_bench_plain_form_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 8 --config 3` WITHOUT a launcher must produce eight ranks (it used to run one and
    print n_gpus 1): bench.py re-executes itself under torch.distributed.run.  And a launcher whose world differs from --gpus is
    refused instead of mislabelled.  (Stub runner, gloo, CPU.)"""
    import json
    import subprocess
    log = str(tmp_path / 'calls')
    env = dict(os.environ, MDGAT_BENCH_STUB_LOG=log, OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY'):
        env.pop(k, None)
    runner = os.path.join(ROOT, 'tests', 'bench_stub_runner.py')
    p = subprocess.run([sys.executable, runner, '--gpus', '8', '--config', '3', '--steps', '2', '--warmup', '1', '--windows', '1'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['rccl_world_size'] == 8 and d['config']['pairs_per_rank'] == [512] * 8
    assert 'starting 8 ranks' in p.stderr
    for r in range(8):
        assert set(open(f'{log}.{r}').read().split()) == ({'512', '40'} if r == 0 else {'512'})     # ('40': rank 0's parity block)
    # a launcher that started 2 ranks for a --gpus 4 request: every rank refuses, nothing is printed
    port = 29911 + (os.getpid() % 80)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), runner, '--gpus', '4', '--steps', '1', '--warmup', '0', '--windows', '1'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert 'WORLD_SIZE=2' in p.stderr


def test_env_defaults_for_rccl_on_this_pool(monkeypatch):
    """bench.py and shard.init_distributed put HSA_ENABLE_IPC_MODE_LEGACY=0 into the environment when nobody chose (RCCL on
    this pool's hosts needs it: INTEGRATION.md) and never override a choice."""
    import importlib
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != 'HSA_ENABLE_IPC_MODE_LEGACY'}
    code = ('import os, sys; sys.argv=["bench.py", "--help"]\n'
            'try:\n    import bench\nexcept SystemExit:\n    pass\n'
            'print("IPC", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))')
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300).stdout
    assert 'IPC 0' in out, out
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY='1'), capture_output=True,
                         text=True, timeout=300).stdout
    assert 'IPC 1' in out, out
