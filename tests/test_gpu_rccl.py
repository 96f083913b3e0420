"""The RCCL (`nccl` backend) code path of shard.py on real hardware, before an 8-GPU job meets it for the first time.

The GPU boxes of the test pool have ONE GPU, and RCCL refuses two ranks on one device, so what can run here is a
1-rank `nccl` process group: communicator set-up (the part that depends on the box: HSA IPC mode, network interface
discovery, xGMI topology files), a broadcast of the packed weight blob from rank 0, barriers, the max-over-ranks
all-reduce and the all-gather of results - all on DEVICE tensors, through the same helpers bench.py uses.  Replaces
`DataParallel.replicate` of the reference (test.py:158).  The world-size-2 logic of the same helpers runs on gloo in
tests/test_shard_gloo.py (CPU) and test_bench_two_ranks_one_gpu (GPU box, both ranks on device 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['MDGAT_ROOT'])
import torch.distributed as dist
from mdgat_matcher_amd import MDGAT, shard, synth
rank, world, local = shard.init_distributed(1)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1, (dist.is_initialized(), world)
dev = torch.device('cuda', local)
L = 3
cfg = synth.default_config(L=L, k=[64, None, 32, None], sinkhorn_iterations=30, arithmetic='fp32')      # the throughput path of a sharded job (bench.py)
sd = synth.make_state_dict(L=L, seed=7)
src = MDGAT(cfg); src.load_state_dict(sd); src = src.eval().to(dev)
d = synth.make_batch(3, 200, 256, device=dev, dtype=torch.float32)
args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
ref = src.match(*args, return_scores=True)
# a rank that never loaded a checkpoint: runs on what the RCCL broadcast delivers
net = MDGAT(cfg).eval().to(dev)
expect = torch.from_numpy(src.packed_weights()).to(dev)
blob = expect.clone()
shard.broadcast_blob(blob, 0)                       # ncclBroadcast on a device tensor
assert torch.equal(blob, expect)
net.load_state_dict(sd)                             # rank 0 of bench.py loads, then broadcasts and installs the blob
got = shard.broadcast_weights(net, dev, rank, world)
assert got is not None and got.is_cuda and torch.equal(got, expect)
net = net.double().eval()                           # test.py:193 - the installed blob must survive the cast
out = net.match(*args, return_scores=True)
for a, b in zip(ref, out):
    assert torch.equal(a, b)
# a module whose arithmetic follows its dtype (no 'arithmetic' key, like the reference's callers): BOTH blobs travel, and the cast of
# test.py:193 after the broadcast switches the rank to the reference-exact mode on rank 0's weights
cfg_auto = {k: v for k, v in cfg.items() if k != 'arithmetic'}
src64 = MDGAT(cfg_auto).double(); src64.load_state_dict(sd); src64 = src64.eval().to(dev)
assert src64.exact()
ref64 = src64.match(*args, return_scores=True)
net2 = MDGAT(cfg_auto).eval().to(dev)               # random init, float32
if rank == 0:
    net2 = net2.double(); net2.load_state_dict(sd)
got64 = []
shard.broadcast_weights(net2, dev, rank, world, out64=got64)
assert len(got64) == 1 and got64[0].dtype == torch.float64 and got64[0].is_cuda
net2 = net2.double().eval()
assert net2.exact()
out64 = net2.match(*args, return_scores=True)
for a, b in zip(ref64, out64):
    assert torch.equal(a, b)
net2.check(dev)
shard.barrier(world)
t = shard.max_over_ranks(1.25, dev, world)          # all-reduce(MAX) on a device tensor
assert t == 1.25
g = shard.gather_matches(out[0], world)             # all-gather on a device tensor
assert g.is_cuda and torch.equal(g, out[0])
torch.cuda.synchronize()
shard.finalize(world)
assert not dist.is_initialized()
print('RCCL_ONE_RANK_OK')
'''


def _env(port):
    return dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                MDGAT_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')


def test_one_rank_nccl_group_runs_the_shard_helpers():
    out = subprocess.run([sys.executable, '-c', _WORKER], cwd=ROOT, env=_env(29650 + os.getpid() % 100), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert 'RCCL_ONE_RANK_OK' in out.stdout


def test_bench_under_torchrun_one_rank_nccl():
    """bench.py launched the way the driver launches the N > 1 runs (torch.distributed.run), with one rank: the process
    group is RCCL, the weight broadcast, both barriers and the max-over-ranks reduction run through it.  BASELINE
    configs[3]'s per-GPU workload (512 pairs per rank), 3 steps."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'MDGAT_SHARE_DEVICE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(29750 + os.getpid() % 100), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--config', '3',
           '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--windows', '1']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['config']['pairs_per_gpu'] == 512
    assert d['config']['collectives'] == 'nccl'            # the run went through RCCL, not around it
    assert d['value'] > 1000
