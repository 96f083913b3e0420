"""ORACLE - test infrastructure, not product code.

CPU restatement (PyTorch, dtype-generic, meant to be run in fp64) of the inference hot path of
``/root/reference/models/mdgat.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped package
(``mdgat_matcher_amd``) never does and has no CPU fallback.

Parity pin: the reference publishes no tests or golden vectors for this path,
so the oracle is pinned against *outputs of the reference itself*: ``tools/make_goldens.py``
imports ``/root/reference/models/mdgat.py`` in the build container, runs it in fp64 on seeded
synthetic weights/inputs and commits the stage tensors under ``tests/golden/``;
``tools/make_goldens_aux.py`` does the same for the steps either side of the matcher
(``utils/utils_test.py``: solve_icp / calculate_error; ``load_data.py``: SparseDataset.__getitem__).
``tests/test_oracle_golden.py`` checks every function below against those fixtures.

Every function cites the reference lines it restates.  The state dict uses the reference's
parameter names, so a real checkpoint (``checkpoint['net']`` with the ``module.`` prefix
stripped) can be fed in unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

BN_EPS = 1e-5          # torch.nn.BatchNorm1d default used by mdgat.py:43
NUM_HEADS = 4          # mdgat.py:255


# ----------------------------------------------------------------------------- per-point MLPs
def _pointwise(w: torch.Tensor, b: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Conv1d(kernel_size=1) on channel-major [B, Cin, N] (mdgat.py:39-40)."""
    return torch.matmul(w[:, :, 0], x) + b[None, :, None]


def _batchnorm_eval(sd, prefix, x):
    """BatchNorm1d in eval mode: running statistics, affine (mdgat.py:43)."""
    mean = sd[f'{prefix}.running_mean'][None, :, None]
    var = sd[f'{prefix}.running_var'][None, :, None]
    g = sd[f'{prefix}.weight'][None, :, None]
    beta = sd[f'{prefix}.bias'][None, :, None]
    return (x - mean) / torch.sqrt(var + BN_EPS) * g + beta


def mlp(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, n_layers: int) -> torch.Tensor:
    """The reference's ``MLP`` stack (mdgat.py:34-46): conv at index 3i, BN at 3i+1, ReLU; the
    last conv has neither BN nor ReLU."""
    for i in range(n_layers):
        x = _pointwise(sd[f'{prefix}.{3 * i}.weight'], sd[f'{prefix}.{3 * i}.bias'], x)
        if i < n_layers - 1:
            x = torch.relu(_batchnorm_eval(sd, f'{prefix}.{3 * i + 1}', x))
    return x


def _count_convs(sd, prefix):
    n = 0
    while f'{prefix}.{3 * n}.weight' in sd:
        n += 1
    return n


def keypoint_encoder(sd, kpts, sigma):
    """KeypointEncoder.forward (mdgat.py:184-188): MLP over [x, y, z, saliency]."""
    inp = torch.cat([kpts.transpose(1, 2), sigma[:, None, :]], dim=1)
    return mlp(sd, 'kenc.encoder', inp, _count_convs(sd, 'kenc.encoder'))


def descriptor_encoder(sd, fpfh):
    """DescriptorEncoder.forward (mdgat.py:152-155): MLP over the 33-D FPFH row."""
    return mlp(sd, 'denc.encoder', fpfh.transpose(1, 2), _count_convs(sd, 'denc.encoder'))


def encode(sd, kpts, sigma, fpfh):
    """mdgat.py:392-393: descriptor encoding + keypoint encoding -> [B, D, N]."""
    return descriptor_encoder(sd, fpfh) + keypoint_encoder(sd, kpts, sigma)


# ----------------------------------------------------------------------------- attention
def attention(q, k, v):
    """mdgat.py:190-194.  q [B, dh, H, N], k/v [B, dh, H, M] -> message [B, dh, H, N], prob."""
    dh = q.shape[1]
    logits = torch.einsum('bdhn,bdhm->bhnm', q, k) / dh ** 0.5
    prob = torch.softmax(logits, dim=-1)
    return torch.einsum('bhnm,bdhm->bdhn', prob, v), prob


def dynamic_attention(q, k, v, topk: int, forced=None, report=None):
    """mdgat.py:196-210: per (batch, head, query) keep the ``topk`` largest logits, softmax over
    those, zero probability elsewhere.  ``topk > M`` raises, as ``torch.topk`` does there.

    Attribution of top-k flips (test infrastructure, no reference counterpart): ``forced`` is a boolean mask
    [B, H, N, M] of the keys ANOTHER implementation kept for the same layer; the softmax then runs over that
    selection instead of this function's own ``topk`` (softmax over a gathered set == masked softmax, 205-209), and
    ``report`` (a list) receives one dict per call describing every row where the two selections differ:
    ``rows`` = number of such rows (``rows_per_pair``: the same count per batch element), ``max_gap`` = largest
    |logit - k-th largest logit| over the keys in the symmetric difference (how far from a tie the disagreement is, in
    the units of the logits; ``max_rel_gap``: relative to the row's largest |logit|, at least 1), ``bad_count`` = rows
    whose forced selection does not hold exactly ``topk`` keys."""
    dh = q.shape[1]
    m = k.shape[3]
    if topk > m:
        raise RuntimeError(f'selected index k out of range: k={topk} > {m} keys')
    logits = torch.einsum('bdhn,bdhm->bhnm', q, k) / dh ** 0.5
    top = logits.topk(topk, dim=3, largest=True, sorted=True)
    if forced is None:
        prob = torch.zeros_like(logits)
        prob.scatter_(3, top.indices, torch.softmax(top.values, dim=-1))
        if report is not None:      # this function's own selection, in the mask form `forced` takes
            report.append({'own': torch.zeros_like(logits, dtype=torch.bool).scatter_(3, top.indices, True)})
    else:
        forced = forced.to(torch.bool)
        assert forced.shape == logits.shape, (forced.shape, logits.shape)
        prob = torch.softmax(logits.masked_fill(~forced, -math.inf), dim=-1)
        if report is not None:
            own = torch.zeros_like(forced)
            own.scatter_(3, top.indices, True)
            diff = own ^ forced
            kth = top.values[..., -1:]
            gap = torch.where(diff, (logits - kth).abs(), torch.zeros_like(logits))
            scale = logits.abs().amax(-1, keepdim=True).clamp(min=1.0)      # (of the row: for networks with large logits)
            report.append({'rows': int(diff.any(-1).sum()), 'total_rows': diff[..., 0].numel(),
                           'rows_per_pair': diff.any(-1).sum(dim=(1, 2)),
                           'max_gap': float(gap.max()), 'max_rel_gap': float((gap / scale).max()),
                           'bad_count': int((forced.sum(-1) != topk).sum()),
                           'row_gaps': gap.amax(-1)[diff.any(-1)]})
    return torch.einsum('bhnm,bdhm->bdhn', prob, v), prob


def multi_head_attention(sd, prefix, x, source, topk: Optional[int], forced=None, report=None):
    """MultiHeadedAttention.forward (mdgat.py:223-237).  The ``view(B, dh, H, -1)`` at 227/231
    sends channel c to (d = c // H, h = c % H)."""
    b, d_model, _ = x.shape
    dh = d_model // NUM_HEADS
    q = _pointwise(sd[f'{prefix}.proj.0.weight'], sd[f'{prefix}.proj.0.bias'], x).view(b, dh, NUM_HEADS, -1)
    k = _pointwise(sd[f'{prefix}.proj.1.weight'], sd[f'{prefix}.proj.1.bias'], source).view(b, dh, NUM_HEADS, -1)
    v = _pointwise(sd[f'{prefix}.proj.2.weight'], sd[f'{prefix}.proj.2.bias'], source).view(b, dh, NUM_HEADS, -1)
    if topk is None:
        msg, _ = attention(q, k, v)
    else:
        msg, _ = dynamic_attention(q, k, v, topk, forced, report)
    msg = msg.contiguous().view(b, d_model, -1)
    return _pointwise(sd[f'{prefix}.merge.weight'], sd[f'{prefix}.merge.bias'], msg)


def attentional_propagation(sd, prefix, x, source, topk, forced=None, report=None):
    """AttentionalPropagation.forward (mdgat.py:246-248): MLP([x ; message])."""
    message = multi_head_attention(sd, f'{prefix}.attn', x, source, topk, forced, report)
    return mlp(sd, f'{prefix}.mlp', torch.cat([x, message], dim=1), 2)


def layer_topk_schedule(L: int, k_list: List[Optional[int]]) -> List[Optional[int]]:
    """mdgat.py:268-272: layer i (0-based over 2L) is dynamic with k_list[i - 2L + len] iff
    i > 2L - 1 - len(k_list); ``None`` entries mean full attention."""
    sched = []
    for i in range(2 * L):
        if i > 2 * L - 1 - len(k_list):
            sched.append(k_list[i - 2 * L + len(k_list)])
        else:
            sched.append(None)
    return sched


def attentional_gnn(sd, desc0, desc1, k_list, L, capture=None, forced_topk=None):
    """AttentionalGNN.forward (mdgat.py:259-276): alternating self/cross layers (352-353); both
    frames use the same layer weights and the pre-update descriptors (270 before 274).

    ``forced_topk`` (test infrastructure): {layer index: (mask0, mask1)} selections for dynamic_attention, see there;
    the per-layer reports land in ``capture['topk_report'][layer] = [frame-0 report, frame-1 report]`` (an empty dict
    makes every dynamic layer report its own selection)."""
    sched = layer_topk_schedule(L, k_list)
    for i in range(2 * L):
        cross = (i % 2 == 1)
        src0, src1 = (desc1, desc0) if cross else (desc0, desc1)
        p = f'gnn.layers.{i}'
        f0 = f1 = rep = None
        if forced_topk is not None and sched[i] is not None:
            f0, f1 = forced_topk.get(i, (None, None))
            rep = [] if capture is not None else None
        delta0 = attentional_propagation(sd, p, desc0, src0, sched[i], f0, rep)
        delta1 = attentional_propagation(sd, p, desc1, src1, sched[i], f1, rep)
        if rep is not None:
            capture.setdefault('topk_report', {})[i] = rep
        desc0, desc1 = desc0 + delta0, desc1 + delta1
        if capture is not None:
            capture[f'layer{i}_desc0'] = desc0
            capture[f'layer{i}_desc1'] = desc1
    return desc0, desc1


# ----------------------------------------------------------------------------- optimal transport
def log_sinkhorn_iterations(Z, log_mu, log_nu, iters: int):
    """mdgat.py:279-285: u-update then v-update, ``iters`` times, log domain."""
    u = torch.zeros_like(log_mu)
    v = torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v[:, None, :], dim=2)
        v = log_nu - torch.logsumexp(Z + u[:, :, None], dim=1)
    return Z + u[:, :, None] + v[:, None, :]


def log_optimal_transport(scores, alpha, iters: int):
    """mdgat.py:288-308: border the [B, N, M] scores with the dustbin score, run Sinkhorn with
    marginals mu = [1..1, M] / (N + M), nu = [1..1, N] / (N + M), then subtract the norm."""
    b, n, m = scores.shape
    alpha = torch.as_tensor(alpha, dtype=scores.dtype, device=scores.device)
    couplings = alpha * torch.ones(b, n + 1, m + 1, dtype=scores.dtype, device=scores.device)
    couplings[:, :n, :m] = scores
    norm = -math.log(n + m)
    log_mu = torch.full((n + 1,), norm, dtype=scores.dtype, device=scores.device)
    log_mu[n] = math.log(m) + norm
    log_nu = torch.full((m + 1,), norm, dtype=scores.dtype, device=scores.device)
    log_nu[m] = math.log(n) + norm
    Z = log_sinkhorn_iterations(couplings, log_mu[None].expand(b, -1), log_nu[None].expand(b, -1), iters)
    return Z - norm


# ----------------------------------------------------------------------------- match extraction
def extract_matches(Z, loss_method='triplet_loss', mutual_check=False, match_threshold=0.2):
    """mdgat.py:441-483, all four branches.  Returns (matches0, matches1, mscores0, mscores1).

    * ``loss_method == 'superglue'`` (444-458): arg-max over the inner N x M block, validity by
      ``exp(max) > match_threshold`` (optionally after a mutual-nearest check).
    * otherwise (461-481): arg-max including the dustbin column/row; a row is valid iff its
      arg-max is not the dustbin (optionally also mutual).
    ``torch.max`` returns the first maximal index on ties.  The reference's all-dustbin quirk
    (465-467: integer zero scores) is reproduced as floating zeros of Z's dtype.
    """
    n, m = Z.shape[1] - 1, Z.shape[2] - 1
    zero = Z.new_zeros(())
    if loss_method == 'superglue':
        inner = Z[:, :n, :m]
        max0, max1 = inner.max(2), inner.max(1)
        idx0, idx1 = max0.indices, max1.indices
        e0, e1 = max0.values.exp(), max1.values.exp()
        if mutual_check:
            ar0 = torch.arange(n, device=Z.device)[None]
            ar1 = torch.arange(m, device=Z.device)[None]
            mutual0 = ar0 == idx1.gather(1, idx0)
            mutual1 = ar1 == idx0.gather(1, idx1)
            ms0 = torch.where(mutual0, e0, zero)
            ms1 = torch.where(mutual1, ms0.gather(1, idx1), zero)
            valid0 = mutual0 & (ms0 > match_threshold)
            valid1 = mutual1 & valid0.gather(1, idx1)
        else:
            valid0, valid1 = e0 > match_threshold, e1 > match_threshold
            ms0, ms1 = torch.where(valid0, e0, zero), torch.where(valid1, e1, zero)
    else:
        max0, max1 = Z[:, :n, :].max(2), Z[:, :, :m].max(1)
        idx0, idx1 = max0.indices, max1.indices
        valid0, valid1 = idx0 < m, idx1 < n
        e0, e1 = max0.values.exp(), max1.values.exp()
        if int(valid0.sum()) == 0:
            ms0, ms1 = torch.zeros_like(e0), torch.zeros_like(e1)
        elif mutual_check:
            # 469-478: among valid rows, keep those whose partner points back
            safe0 = torch.where(valid0, idx0, torch.zeros_like(idx0))
            safe1 = torch.where(valid1, idx1, torch.zeros_like(idx1))
            ar0 = torch.arange(n, device=Z.device)[None].expand_as(idx0)
            ar1 = torch.arange(m, device=Z.device)[None].expand_as(idx1)
            mutual0 = valid0 & (ar0 == idx1.gather(1, safe0))
            mutual1 = valid1 & (ar1 == idx0.gather(1, safe1))
            ms0, ms1 = torch.where(mutual0, e0, zero), torch.where(mutual1, e1, zero)
        else:
            ms0, ms1 = torch.where(valid0, e0, zero), torch.where(valid1, e1, zero)
    matches0 = torch.where(valid0, idx0, torch.full_like(idx0, -1))
    matches1 = torch.where(valid1, idx1, torch.full_like(idx1, -1))
    return matches0, matches1, ms0, ms1


# ----------------------------------------------------------------------------- dead-code kNN helpers
def knn(x, src, k: int):
    """mdgat.py:8-15: indices of the k nearest ``src`` columns for every ``x`` column
    (x [B, C, N], src [B, C, M]) by negative squared distance, ``topk`` order."""
    inner = -2.0 * torch.matmul(x.transpose(2, 1), src)
    xx = (x ** 2).sum(dim=1, keepdim=True)
    ss = (src ** 2).sum(dim=1, keepdim=True)
    neg_dist = -xx.transpose(2, 1) - inner - ss
    return neg_dist.topk(k=k, dim=-1)[1]


def knn_adjacency(x, src, k: int, idx=None):
    """mdgat.py:17-32 (get_graph_feature): dense 0/1 int64 adjacency [B, N, M] of the kNN graph."""
    b, _, n = x.shape
    m = src.shape[2]
    if idx is None:
        idx = knn(x, src, k)
    adj = torch.zeros(b, n, m, dtype=torch.int64, device=x.device)
    adj.scatter_(2, idx, 1)
    return adj


# ----------------------------------------------------------------------------- frame decode (loader)
def decode_frames(records):
    """SparseDataset.__getitem__'s record handling (load_data.py:152-165, 290-295) for a batch of raw keypoint
    frames ``[B, N, 37]`` float32: xyz = [:, :3], saliency = [:, 3], FPFH = [:, 4:] scaled by 1 / ||FPFH||_2
    (computed in float32 like numpy does there), everything then cast to float64.
    Returns (keypoints [B, N, 3], scores [B, N], descriptors [B, N, 33])."""
    import numpy as np
    rec = np.asarray(records, dtype=np.float32)
    kp = rec[..., :3]
    score = rec[..., 3]
    desc = rec[..., 4:]
    norm = np.linalg.norm(desc, axis=-1)[..., None]
    desc = np.multiply(desc, 1 / norm)
    return (torch.tensor(kp, dtype=torch.double), torch.tensor(score, dtype=torch.double),
            torch.tensor(desc, dtype=torch.double))


def fpfh_normalise_float32_steps(desc):
    """The float32 operations behind ``np.multiply(descs, 1 / np.linalg.norm(descs, axis=1))`` (load_data.py:290-292) one by one,
    for rows of 33 values: squares rounded to float32, summed in the order of numpy's pairwise reduction for 8 <= n <= 128 (eight
    strided partial sums over the first 32 values, combined as a balanced tree, the 33rd added last), square root, reciprocal and
    products each rounded once.  This is the order ``assemble_frames_f64_kernel`` (csrc/f64.hip) implements for the exact
    mode's record entry; tests/test_oracle_golden.py holds it to numpy's own result and to the reference loader's outputs."""
    import numpy as np
    f = np.float32
    d = np.asarray(desc, dtype=f)
    assert d.shape[-1] == 33
    sq = (d * d).astype(f)
    r = [sq[..., j].copy() for j in range(8)]
    for i in (8, 16, 24):
        for j in range(8):
            r[j] = (r[j] + sq[..., i + j]).astype(f)
    s = (((r[0] + r[1]).astype(f) + (r[2] + r[3]).astype(f)).astype(f) + ((r[4] + r[5]).astype(f) + (r[6] + r[7]).astype(f)).astype(f)).astype(f)
    s = (s + sq[..., 32]).astype(f)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = (f(1) / np.sqrt(s).astype(f)).astype(f)
        return (d * inv[..., None]).astype(f)


# ----------------------------------------------------------------------------- pose from matches (evaluation)
def solve_icp(P, Q):
    """utils/utils_test.py:73-110 (numpy, float64): rigid transform taking P onto Q from one SVD, R = U Vh, no
    reflection fix."""
    import numpy as np
    P = np.asarray(P, dtype=np.float64)
    Q = np.asarray(Q, dtype=np.float64)
    up, uq = P.mean(axis=0), Q.mean(axis=0)
    U, _, Vh = np.linalg.svd(np.dot((Q - uq).T, P - up), full_matrices=True, compute_uv=True)
    R = np.dot(U, Vh)
    T = np.zeros((4, 4))
    T[:3, :3] = R
    T[:3, 3] = uq - np.dot(R, up)
    T[3, 3] = 1.0
    return T


def pose_from_matches(kpts0, kpts1, matches0, T_gt=None, inlier_dist=1.0):
    """test.py:213-216 (selection of the matched keypoints) + calculate_error (utils_test.py:41-71) for one pair.
    Returns (T, matches, inliers, inlier_ratio, trans_error, rot_error)."""
    import numpy as np
    kpts0 = np.asarray(kpts0, dtype=np.float64)
    kpts1 = np.asarray(kpts1, dtype=np.float64)
    matches0 = np.asarray(matches0)
    valid = matches0 > -1
    mk0, mk1 = kpts0[valid], kpts1[matches0[valid]]
    T = solve_icp(mk1, mk0)
    w = (T[:3, :3] @ mk1.T).T + T[:3, 3]
    inl = int((np.linalg.norm(w - mk0, axis=1) < inlier_dist).sum())
    te = re = float('nan')
    if T_gt is not None:
        E = np.linalg.inv(T) @ np.asarray(T_gt, dtype=np.float64)
        te = float(np.linalg.norm(E[:3, 3]))
        with np.errstate(invalid='ignore'):
            re = float(np.arccos((E[0, 0] + E[1, 1] + E[2, 2] - 1) * 0.5))
    return T, int(valid.sum()), inl, inl / max(int(valid.sum()), 1), te, re


def frame_transforms(pose0, pose1, T_cam0_velo):
    """load_data.py:231-239 for one pair of frames: the sensor->world transform of each frame (pose . T_cam0_velo, the
    einsum at 238-239) and T_gt = inv(T_cam0_velo) inv(pose0) pose1 T_cam0_velo (235), which maps frame-1 points onto
    frame 0.  poses are the 4x4 camera-0 poses of KITTI/poses/NN.txt, T_cam0_velo the `Tr` row of calib.txt.
    Returns (T0, T1, T_gt) as float64 arrays."""
    import numpy as np
    p0, p1, tcv = (np.asarray(x, dtype=np.float64) for x in (pose0, pose1, T_cam0_velo))
    T_gt = np.linalg.inv(tcv) @ np.linalg.inv(p0) @ p1 @ tcv
    return p0 @ tcv, p1 @ tcv, T_gt


def gt_matches(kp0, kp1, T0=None, T1=None, threshold=0.5, mutual=False):
    """load_data.py:238-285 for one pair: world-frame keypoints, cdist, arg-mins both ways, threshold, optional
    mutual check.  Returns (match0 [N], match1 [M], rep)."""
    import numpy as np
    from scipy.spatial.distance import cdist
    def world(kp, T):
        kp = np.asarray(kp, dtype=np.float64)
        if T is None:
            return kp
        T = np.asarray(T, dtype=np.float64)
        return (T[:3, :3] @ kp.T).T + T[:3, 3]
    a, b = world(kp0, T0), world(kp1, T1)
    dists = cdist(a, b)
    min1 = np.argmin(dists, axis=0)
    min2 = np.argmin(dists, axis=1)
    min1v = np.min(dists, axis=1)
    min1f = min2[min1v < threshold]
    rep = len(min1f)
    match1, match2 = -1 * np.ones(len(a), dtype=np.int64), -1 * np.ones(len(b), dtype=np.int64)
    if mutual:
        xx = np.where(min2[min1] == np.arange(min1.shape[0]))[0]
        matches = np.intersect1d(min1f, xx)
        match1[min1[matches]] = matches
        match2[matches] = min1[matches]
    else:
        match1[min1v < threshold] = min1f
        min2v = np.min(dists, axis=0)
        match2[min2v < threshold] = min1[min2v < threshold]
    return match1, match2, rep


# ----------------------------------------------------------------------------- whole forward
def mdgat_forward(sd: Dict[str, torch.Tensor], config: dict, data: dict, capture: Optional[dict] = None,
                  forced_topk: Optional[dict] = None):
    """MDGAT.forward for ``descriptor == 'FPFH'`` (mdgat.py:369-483, 596-603), loss excluded.

    ``capture`` (optional dict) receives the stage tensors the golden fixtures hold; ``forced_topk`` replaces the
    top-k selections of the dynamic layers (see attentional_gnn / dynamic_attention)."""
    dtype = sd['bin_score'].dtype
    kpts0, kpts1 = data['keypoints0'].to(dtype), data['keypoints1'].to(dtype)
    if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:          # mdgat.py:374-382
        s0, s1 = kpts0.shape[:-1], kpts1.shape[:-1]
        return {
            'matches0': torch.full(s0, -1, dtype=torch.int32)[0],
            'matches1': torch.full(s1, -1, dtype=torch.int32)[0],
            'matching_scores0': torch.zeros(s0, dtype=dtype)[0],
            'matching_scores1': torch.zeros(s1, dtype=dtype)[0],
            'skip_train': True,
        }
    L = config['L']
    d_model = config.get('descriptor_dim', 128)
    desc0 = encode(sd, kpts0, data['scores0'].to(dtype), data['descriptors0'].to(dtype))
    desc1 = encode(sd, kpts1, data['scores1'].to(dtype), data['descriptors1'].to(dtype))
    if capture is not None:
        capture['enc0'], capture['enc1'] = desc0, desc1
    desc0, desc1 = attentional_gnn(sd, desc0, desc1, config['k'], L, capture, forced_topk)
    mdesc0 = _pointwise(sd['final_proj.weight'], sd['final_proj.bias'], desc0)       # mdgat.py:397
    mdesc1 = _pointwise(sd['final_proj.weight'], sd['final_proj.bias'], desc1)
    scores = torch.einsum('bdn,bdm->bnm', mdesc0, mdesc1) / d_model ** 0.5           # mdgat.py:430-431
    Z = log_optimal_transport(scores, sd['bin_score'], config.get('sinkhorn_iterations', 100))
    if capture is not None:
        capture['mdesc0'], capture['mdesc1'] = mdesc0, mdesc1
        capture['scores'], capture['Z'] = scores, Z
    m0, m1, s0, s1 = extract_matches(Z, config.get('loss_method', 'triplet_loss'),
                                     config.get('mutual_check', False),
                                     config.get('match_threshold', 0.2))
    return {'matches0': m0, 'matches1': m1, 'matching_scores0': s0, 'matching_scores1': s1}
