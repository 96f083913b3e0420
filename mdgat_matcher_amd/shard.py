"""Multi-GPU sharding of frame-pair batches: one process per GPU, no data-path collective.

The reference's only multi-GPU mechanism is single-process ``torch.nn.DataParallel``
(``test.py:158``, ``train.py:192-196``): it re-broadcasts all parameters and scatters/gathers the batch
on EVERY forward.  Frame pairs are independent (no exchange step inside a pair), so here every rank owns
a contiguous slice of the pair list and the only collective is a one-time RCCL broadcast (over xGMI) of
the packed fp32 weight blob (~11 MB) from rank 0 - a single latency-bound message, after which the
steady state has no cross-GPU dependency at all.  ``gather_matches`` is optional result consolidation
(~12 KB per pair)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def _group_active() -> bool:
    return dist.is_available() and dist.is_initialized()


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def init_distributed(n_gpus_requested: int = 1, backend: str | None = None, share_device: bool | None = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, local).

    A process group is created whenever the process was started by a launcher (RANK and MASTER_ADDR in the environment),
    also with ONE rank: `python -m torch.distributed.run --nproc-per-node 1 bench.py` then runs the same RCCL
    communicator set-up, broadcast, barriers and reduction as an 8-GPU job (tests/test_gpu_rccl.py), and the helpers
    below run their collective whenever a group exists.  Started without a launcher there is no group and no collective.

    ``share_device`` (tests only; the environment variable MDGAT_SHARE_DEVICE stands for it when the argument is None):
    every rank drives device 0 and the collectives run on gloo - the N > 1 control flow on a box with a single GPU, where
    RCCL refuses two ranks on one device.  It is refused on a box with more than one GPU and announced on stderr."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if share_device is None:
        share_device = bool(os.environ.get('MDGAT_SHARE_DEVICE'))
    if share_device:
        if torch.cuda.is_available() and torch.cuda.device_count() > 1:
            raise RuntimeError('MDGAT_SHARE_DEVICE / share_device is a single-GPU test hook: this box has '
                               f'{torch.cuda.device_count()} GPUs, every rank must drive its own')
        import sys
        print(f'[mdgat shard] rank {rank}: share_device test hook active - all {world} ranks on device 0, gloo collectives',
              file=sys.stderr, flush=True)
        local, backend = 0, 'gloo'
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    if (world > 1 or launched) and not _group_active():
        # the host driver of this pool shares device memory between processes through dmabuf only (RCCL communicator set-up
        # fails with `hipIpcGetMemHandle: invalid argument` otherwise); harmless elsewhere, and only set when nobody chose
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            # (a launcher always sets it; without one the ranks of a multi-rank job cannot agree on a free port by themselves,
            # a single rank can)
            os.environ['MASTER_PORT'] = str(_free_port()) if world == 1 else '29531'
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # 'nccl' IS RCCL on ROCm
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def partition(n_pairs: int, rank: int, world: int):
    """Contiguous shard [first, first + count) of the pair list for ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(n_pairs, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    if _group_active():
        if blob.is_cuda and dist.get_backend() != 'nccl':      # gloo: through host memory
            host = blob.cpu()
            dist.broadcast(host, src=src)
            blob.copy_(host)
        else:
            dist.broadcast(blob, src=src)
    return blob


def broadcast_weights(net, device, rank: int, world: int, out64=None):
    """Rank 0 packs its parameters (BN fold etc., pack.py); the packed blob travels once by RCCL broadcast
    and every rank installs it into its library handle.  Without a process group the module just packs lazily.
    A module that runs the reference-exact mode - or may, once the caller casts it: a float64 module, arithmetic='fp64', and
    every module whose arithmetic follows its dtype ('auto'; test.py:193 casts before every forward) - also needs the blob before
    its rounding to fp32: it travels in a second broadcast (``out64``, a list, receives it - tests).  Only arithmetic='fp32'
    (bench.py's headline) skips it."""
    if world == 1 and not _group_active():
        return None
    from . import pack
    n = pack.blob_layout(net.config['L'])['total']
    if rank == 0:
        blob = torch.from_numpy(net.packed_weights()).to(device)
    else:
        blob = torch.empty(n, dtype=torch.float32, device=device)
    broadcast_blob(blob, 0)
    blob64 = None
    if net.exact() or getattr(net, 'arithmetic', 'auto') == 'auto':
        # the reference-exact mode loads the folded weights before their rounding to fp32 as well (29 MB at L = 9)
        import numpy as np
        if rank == 0:
            blob64 = torch.from_numpy(net.packed_weights(np.float64)).to(device)
        else:
            blob64 = torch.empty(n, dtype=torch.float64, device=device)
        broadcast_blob(blob64, 0)
        if out64 is not None:
            out64.append(blob64)
    if blob.is_cuda:
        net.load_packed(blob, blob64)
    return blob


def barrier(world: int):
    if world > 1 or _group_active():
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device, world: int) -> float:
    if world == 1 and not _group_active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_matches(local: torch.Tensor, world: int):
    """Optional: concatenate per-rank results on every rank (equal shard sizes)."""
    if world == 1 and not _group_active():
        return local
    if local.is_cuda and dist.get_backend() != 'nccl':         # gloo: through host memory
        host = local.cpu()
        outs = [torch.empty_like(host) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, host)
        return torch.cat(outs, dim=0).to(local.device)
    outs = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, local)
    return torch.cat(outs, dim=0)


def finalize(world: int):
    if _group_active():
        dist.destroy_process_group()
