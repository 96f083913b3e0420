"""Deterministic synthetic weights and frame pairs for parity tests and the bench.

There is no pretrained checkpoint in the reference tree (``pre-trained/best_model.pth`` is a
missing blob) and no KITTI keypoint files, so every fixture and bench input is generated here
from ``numpy.random.RandomState`` (the frozen legacy stream: identical on every box).

* ``make_state_dict`` emits tensors under the reference's parameter/buffer names
  (``/root/reference/models/mdgat.py:325-360``: ``kenc.encoder.N``, ``denc.encoder.N``,
  ``gnn.layers.i.attn.{merge,proj.j}``, ``gnn.layers.i.mlp.N``, ``final_proj``, ``bin_score``)
  with non-trivial BatchNorm running statistics so that BN folding is actually exercised.
* ``make_pair`` / ``make_batch`` emit keypoints, saliency and L2-normalised 33-D FPFH rows in
  the layout the reference's loader hands to the model
  (``/root/reference/load_data.py:152-165`` record split, ``290-292`` normalisation,
  ``299-321`` dict keys).
"""
from __future__ import annotations

import numpy as np
import torch

DEFAULT_K = [128, None, 128, None, 64, None, 64, None]


def effective_cpu_count() -> int:
    """Cores this process may really use: min(os.cpu_count(), affinity mask, cgroup v2 CPU quota).
    (The GPU boxes report 256 CPUs but run under a 16-core quota; 256 OpenMP threads there are
    ~2000x slower than 16.)"""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def default_config(L=9, k=None, sinkhorn_iterations=100, **over):
    """Config dict with the keys ``test.py:137-151`` passes to ``MDGAT(config)``."""
    cfg = {
        'sinkhorn_iterations': sinkhorn_iterations,
        'match_threshold': 0.2,
        'lr': 1e-4,
        'loss_method': 'triplet_loss',
        'k': list(DEFAULT_K) if k is None else list(k),
        'descriptor': 'FPFH',
        'mutual_check': False,
        'triplet_loss_gamma': 0.5,
        'train_step': 3,
        'L': L,
    }
    cfg.update(over)
    return cfg


def _conv(rs, cout, cin, zero_bias=False, gain=1.0):
    w = rs.standard_normal((cout, cin, 1)) * (gain / np.sqrt(cin))
    b = np.zeros(cout) if zero_bias else 0.1 * rs.standard_normal(cout)
    return w, b


def _bn(rs, c):
    return {
        'weight': rs.uniform(0.5, 1.5, c),
        'bias': 0.1 * rs.standard_normal(c),
        'running_mean': 0.1 * rs.standard_normal(c),
        'running_var': rs.uniform(0.5, 1.5, c),
        'num_batches_tracked': np.asarray(7, dtype=np.int64),
    }


def _mlp(rs, sd, prefix, channels):
    """Sequential index convention of the reference's MLP(): conv at 3*i, BN at 3*i+1 (all but last)."""
    n = len(channels)
    for i in range(1, n):
        idx = 3 * (i - 1)
        w, b = _conv(rs, channels[i], channels[i - 1])
        sd[f'{prefix}.{idx}.weight'] = w
        sd[f'{prefix}.{idx}.bias'] = b
        if i < n - 1:
            for k, v in _bn(rs, channels[i]).items():
                sd[f'{prefix}.{idx + 1}.{k}'] = v


def make_state_dict(L=9, seed=0, bin_score=1.0, dtype=torch.float64, feature_dim=128,
                    keypoint_encoder=(32, 64, 128), descriptor_encoder=(64, 128),
                    logit_std=4.0, score_std=1.0):
    """Reference-named state dict (348 entries at L=9) from a seed.

    The scales are chosen so that a random-weight network behaves like a trained one
    numerically: keypoint coordinates (tens of metres) are brought to O(1) by the first
    keypoint-encoder conv, the BatchNorm running variances of the GNN MLPs track the slowly
    growing variance of the residual stream (as trained statistics would), attention logits have
    a standard deviation of about ``logit_std`` in every layer (peaked but not one-hot rows, so
    the top-k selection and the softmax tail both matter) and pre-OT scores of unrelated
    keypoints have a standard deviation of about ``score_std``."""
    rs = np.random.RandomState(seed)
    sd = {}
    D = feature_dim
    _mlp(rs, sd, 'kenc.encoder', [4, *keypoint_encoder, D])
    sd['kenc.encoder.0.weight'][:, :3] /= 20.0
    _mlp(rs, sd, 'denc.encoder', [33, *descriptor_encoder, D])
    sd['denc.encoder.0.weight'] *= 4.0           # L2-normalised 33-D rows have entries ~0.17
    var_x = 0.5                                   # variance of the encoder sum; grows by ~dv per layer
    dv = 0.3
    for i in range(2 * L):
        p = f'gnn.layers.{i}'
        qk_gain = float(np.sqrt(logit_std / (0.5 + 0.09 * i)))   # measured stream variance, see docstring
        for name, gain in (('attn.merge', 1.0), ('attn.proj.0', qk_gain), ('attn.proj.1', qk_gain),
                           ('attn.proj.2', 1.0)):
            w, b = _conv(rs, D, D, gain=gain)
            sd[f'{p}.{name}.weight'] = w
            sd[f'{p}.{name}.bias'] = b
        _mlp(rs, sd, f'{p}.mlp', [2 * D, 2 * D, D])
        sd[f'{p}.mlp.1.running_var'] *= var_x
        sd[f'{p}.mlp.1.running_mean'] *= float(np.sqrt(var_x))
        sd[f'{p}.mlp.3.weight'] = sd[f'{p}.mlp.3.weight'] * 0.5
        var_x += dv
    w, b = _conv(rs, D, D, gain=float(np.sqrt(score_std / (0.5 + 0.09 * 2 * L))))
    sd['final_proj.weight'] = w
    sd['final_proj.bias'] = b
    sd['bin_score'] = np.asarray(bin_score, dtype=np.float64)
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).reshape(np.shape(v))
        out[k] = t if t.dtype == torch.int64 else t.to(dtype)
    return out


def make_frame(n, seed, dtype=np.float64):
    """One synthetic frame in the loader's 37-float record layout: xyz(3) | saliency(1) | FPFH(33)."""
    rs = np.random.RandomState(seed)
    kpts = 20.0 * rs.standard_normal((n, 3))
    sigma = rs.uniform(0.0, 1.0, n)
    fpfh = rs.uniform(0.0, 1.0, (n, 33))
    fpfh = fpfh / np.linalg.norm(fpfh, axis=1, keepdims=True)
    return kpts.astype(dtype), sigma.astype(dtype), fpfh.astype(dtype)


def make_pair(n, m, pair_index=0, dtype=np.float64, base_seed=1234, correlated=True):
    """Frame pair ``pair_index``.  With ``correlated`` the second frame re-observes a subset of
    the first frame's keypoints (rigid motion + noise) so some rows really match and others go
    to the dustbin - both extraction branches are then exercised."""
    k0, s0, f0 = make_frame(n, base_seed + 2 * pair_index, dtype=np.float64)
    k1, s1, f1 = make_frame(m, base_seed + 2 * pair_index + 1, dtype=np.float64)
    if correlated:
        rs = np.random.RandomState(base_seed + 7919 * (pair_index + 1))
        nshare = min(n, m) // 2
        src = rs.permutation(n)[:nshare]
        dst = rs.permutation(m)[:nshare]
        th = 0.1 * rs.standard_normal()
        R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        t = rs.standard_normal(3)
        k1[dst] = k0[src] @ R.T + t + 0.05 * rs.standard_normal((nshare, 3))
        f = f0[src] + 0.02 * rs.standard_normal((nshare, 33))
        f1[dst] = np.abs(f) / np.linalg.norm(f, axis=1, keepdims=True)
        s1[dst] = np.clip(s0[src] + 0.02 * rs.standard_normal(nshare), 0, 1)
    return tuple(a.astype(dtype) for a in (k0, s0, f0, k1, s1, f1))


def make_batch(B, n, m, first_pair=0, dtype=torch.float64, device='cpu', base_seed=1234):
    """Collated dict with the keys the hot path reads (``mdgat.py:372, 390-393``)."""
    cols = [[] for _ in range(6)]
    for b in range(B):
        for c, a in zip(cols, make_pair(n, m, first_pair + b, base_seed=base_seed)):
            c.append(a)
    k0, s0, f0, k1, s1, f1 = [torch.from_numpy(np.stack(c)).to(dtype).to(device) for c in cols]
    return {
        'keypoints0': k0, 'scores0': s0, 'descriptors0': f0,
        'keypoints1': k1, 'scores1': s1, 'descriptors1': f1,
        'gt_matches0': torch.full((B, n), -1, dtype=torch.int16, device=device),
        'gt_matches1': torch.full((B, m), -1, dtype=torch.int16, device=device),
    }
