"""``models.mdgat`` as the reference's callers import it (``test.py:12``, ``test_registration_metric.py:12``:
``from models.mdgat import MDGAT``), resolved to the MI355X implementation.

``MDGAT`` is ``mdgat_matcher_amd.MDGAT``: same constructor config, parameter names, ``forward(dict) -> dict``.  The free
functions of ``/root/reference/models/mdgat.py`` that have a kernel behind them are exported under their reference
names and signatures (channel-major ``[B, dh, H, N]`` tensors in and out, like ``mdgat.py:190-210``); they run on the
gfx950 library through ``mdgat_matcher_amd.ops`` - there is no CPU path here either.  As in ``MDGAT`` itself the dtype of
the tensors is the arithmetic request: float64 q / k / v (what the reference's ``net.double()`` produces) run the fp64
kernels of ``csrc/f64.hip`` - ``dynamic_attention`` then keeps exactly the keys ``torch.topk`` keeps on the fp64 logits -
float32 tensors the fp32-class kernels.
"""
from __future__ import annotations

import torch

from mdgat_matcher_amd import ops as _ops
from mdgat_matcher_amd.mdgat import MDGAT, match  # noqa: F401
from mdgat_matcher_amd.mdgat import _mlp_modules as _mlp_modules

__all__ = ['MDGAT', 'match', 'MLP', 'attention', 'dynamic_attention', 'log_optimal_transport', 'knn', 'get_graph_feature']


def MLP(channels: list, do_bn=True):
    """mdgat.py:34-46: the Conv1d(k=1) / BatchNorm1d / ReLU stack as a parameter container with the reference's
    Sequential indices (the encoders and propagation MLPs of ``MDGAT`` hold their weights in these)."""
    return _mlp_modules(list(channels))


def _to_lib(q, k, v, min_rows=0):
    """[B, dh, H, N] / [B, dh, H, M] (mdgat.py:227-232 after the view) -> the library's [B, N' + M, 3, H, dh] with the
    queries as frame 0 and the keys / values as frame 1 (a cross layer: frame 0 attends to frame 1).  N' = max(N,
    min_rows): the library's layer form checks a dynamic k against BOTH frames, zero query rows pad frame 0 up to k."""
    B, dh, H, N = q.shape
    M = k.shape[3]
    Np = max(N, min_rows)
    qkv = torch.zeros(B, Np + M, 3, H, dh, dtype=torch.float64 if q.dtype == torch.float64 else torch.float32, device=q.device)
    qkv[:, :N, 0] = q.permute(0, 3, 2, 1)
    qkv[:, Np:, 1] = k.permute(0, 3, 2, 1)
    qkv[:, Np:, 2] = v.permute(0, 3, 2, 1)
    return qkv, Np, M


def _from_lib(msg, N, like):
    B = msg.shape[0]
    return msg[:, :N].reshape(B, N, 4, 32).permute(0, 3, 2, 1).to(like.dtype)


def attention(query, key, value):
    """mdgat.py:190-194.  Returns ``(message [B, dh, H, N], None)``: the probability tensor the reference returns second
    is write-only there (``self.prob.append``, mdgat.py:236) and is never materialised here."""
    qkv, Np, M = _to_lib(query, key, value)
    op = _ops.attention_f64 if qkv.dtype == torch.float64 else _ops.attention
    return _from_lib(op(qkv, Np, M, cross=True, topk=0), query.shape[3], query), None


def dynamic_attention(query, key, value, k):
    """mdgat.py:196-210: attention over the k largest logits of every query.  ``k`` larger than the number of keys
    raises, as ``torch.topk`` does there."""
    if int(k) > key.shape[3]:
        raise RuntimeError(f'selected index k out of range: k={int(k)} exceeds the number of keys {key.shape[3]}')
    qkv, Np, M = _to_lib(query, key, value, min_rows=int(k))
    op = _ops.attention_f64 if qkv.dtype == torch.float64 else _ops.attention
    return _from_lib(op(qkv, Np, M, cross=True, topk=int(k)), query.shape[3], query), None


def log_optimal_transport(scores, alpha, iters: int):
    """mdgat.py:288-308: scores [B, N, M] -> log assignment matrix [B, N+1, M+1] (log-domain Sinkhorn)."""
    return _ops.sinkhorn(scores, float(alpha), int(iters)).to(scores.dtype)


def knn(x, src, k):
    """mdgat.py:8-15: indices [B, N, k] of the k nearest ``src`` columns of every ``x`` column (x [B, C, N], src [B, C, M])."""
    return _ops.knn(x, src, int(k))


def get_graph_feature(x, src, k=20, idx=None):
    """mdgat.py:17-32: dense 0/1 int64 adjacency [B, N, M] of the kNN graph."""
    return _ops.knn(x, src, int(k), adjacency=True)[1]
