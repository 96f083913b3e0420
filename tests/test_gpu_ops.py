"""GPU parity of every kernel, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs and against the committed golden vectors produced by the real reference.

Tolerances: index outputs bit-exact; floating outputs within 1e-4 absolute of the fp64 oracle (the
north-star bound for the soft-assignment matrix), most stages far tighter as stated per test."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mdgat_matcher_amd import _lib, ops  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402

DEV = 'cuda:0'


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('rows,cout,K', [(128, 128, 128), (300, 384, 128), (1000, 64, 32), (77, 256, 256), (64, 128, 64),
                                         (5, 40, 96)])
def test_pointwise(rows, cout, K):
    rs = np.random.RandomState(rows + cout + K)
    A = torch.from_numpy(rs.standard_normal((rows, K)))
    W = torch.from_numpy(rs.standard_normal((cout, K)) / np.sqrt(K))
    b = torch.from_numpy(rs.standard_normal(cout))
    R = torch.from_numpy(rs.standard_normal((rows, cout)))
    ref = A @ W.T + b
    out = ops.pointwise(A.to(DEV), W.to(DEV), b.to(DEV)).cpu().double()
    assert (out - ref).abs().max() < 2e-5
    ref2 = torch.relu(ref) + R
    out2 = ops.pointwise(A.to(DEV), W.to(DEV), b.to(DEV), relu=True, residual=R.to(DEV)).cpu().double()
    assert (out2 - ref2).abs().max() < 2e-5
    out3 = ops.pointwise(A.to(DEV), W.to(DEV)).cpu().double()
    assert (out3 - A @ W.T).abs().max() < 2e-5


def test_pointwise_is_transpose_safe():
    # asymmetric operand: a swapped C/D fragment mapping cannot pass
    A = torch.zeros(64, 32, dtype=torch.float64)
    A[torch.arange(32), torch.arange(32)] = 1.0   # rows 0..31 = identity, rows 32..63 zero
    W = torch.arange(64 * 32, dtype=torch.float64).reshape(64, 32) / 100.0
    out = ops.pointwise(A.to(DEV), W.to(DEV)).cpu().double()
    assert (out - A @ W.T).abs().max() < 1e-6


def _to_lib_qkv(q, k, v):
    """reference [B, dh, H, N] tensors -> library [B, N, 3, H, dh] (self-attention frame layout helper)"""
    return torch.stack([t.permute(0, 3, 2, 1) for t in (q, k, v)], dim=2).contiguous()


def _ref_msg_to_lib(msg):
    """reference message [B, dh, H, N] -> library [B, N, H*dh]"""
    b, dh, h, n = msg.shape
    return msg.permute(0, 3, 2, 1).reshape(b, n, h * dh)


@pytest.mark.parametrize('N,M', [(64, 64), (40, 56), (128, 96), (512, 512), (257, 130), (33, 31), (600, 700), (1024, 1024),
                                 (2048, 1300), (128, 320), (192, 64), (2048, 1344)])   # multiples of 64: the streamed kernel
@pytest.mark.parametrize('cross', [False, True])
def test_attention_full(N, M, cross):
    rs = np.random.RandomState(N * 7 + M)
    B = 2 if N <= 1024 else 1
    qkv = torch.from_numpy(rs.standard_normal((B, N + M, 3, 4, 32)) * 1.3)
    out = ops.attention(qkv.to(DEV), N, M, cross).cpu().double()
    # oracle per frame
    for side, (lo, hi) in enumerate(((0, N), (N, N + M))):
        slo, shi = ((N, N + M) if side == 0 else (0, N)) if cross else (lo, hi)
        q = qkv[:, lo:hi, 0].permute(0, 3, 2, 1)      # [B, dh, H, n]
        k = qkv[:, slo:shi, 1].permute(0, 3, 2, 1)
        v = qkv[:, slo:shi, 2].permute(0, 3, 2, 1)
        ref, _ = O.attention(q, k, v)
        err = (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max()
        assert err < 1e-5, (side, err)


@pytest.mark.parametrize('B,N,M,cross', [(3, 512, 512, False), (2, 256, 512, True), (1, 1024, 768, False), (2, 64, 64, True)])
def test_qk_phase_probes_agree(B, N, M, cross):
    """bench.py's roofline_qk legs: the standalone Q K^T phase kernels with 32 and 64 queries per wave leave the row maxima of
    the base-2 logits in msg[:, :, head * 32] - bit-identical to each other, the oracle's (fp64) to fp32 resolution; the Q K^T
    phase of the shipped streamed kernel (softmax and P.V knocked out) leaves its RUNNING softmax reference there, which
    follows the maximum only on jumps beyond 4 octaves: within [max - 4, max]."""
    rs = np.random.RandomState(B + N + M)
    qkv = torch.from_numpy(rs.standard_normal((B, N + M, 3, 4, 32)) * 1.3)
    probe = ops.QkProbe(qkv.to(DEV), N, M)
    outs = [probe.run(cross, nq_sets=s).clone()[:, :, ::32] for s in (0, 1, 2)]        # [B, N + M, 4 heads]
    torch.cuda.synchronize()
    assert torch.equal(outs[1], outs[2])
    assert (outs[0] <= outs[1]).all() and (outs[0] >= outs[1] - 4.0).all()
    for lo, hi, slo, shi in ((0, N, *((N, N + M) if cross else (0, N))), (N, N + M, *((0, N) if cross else (N, N + M)))):
        q = qkv[:, lo:hi, 0].permute(0, 3, 2, 1)
        k = qkv[:, slo:shi, 1].permute(0, 3, 2, 1)
        ref = (torch.einsum('bdhn,bdhm->bhnm', q, k) / 32 ** 0.5 * 1.4426950408889634).amax(-1).permute(0, 2, 1)   # [B, n, H]
        assert (outs[1][:, lo:hi].cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize('N,topk', [(512, 0), (512, 128), (320, 0), (100, 30), (1024, 0), (1024, 128)])
@pytest.mark.parametrize('case', ['small_q_large_k', 'small_v', 'small_everything', 'large'])
def test_attention_mismatched_magnitudes(N, topk, case):
    """The q / k / v planes carry an UNSCALED residual, lo = f16(x - hi) (common.hpp): its rounding is absolute (2^-25, an
    f16 denormal below |x| = 0.25), so the products are fp32-class (2^-22 relative) for operands between ~0.02 and 6e4 and
    carry an absolute error of ~3e-8 per term below that.  Trained checkpoints need not sit at order one: queries ~1e-3
    against keys ~30, values ~1e-3, everything tiny, logits of several hundred.  What is asserted is that statement,
    measured against the fp64 oracle (tools/magnitude_probe.py): message error <= |v| (3e-6 + 1.5e-7 max|logit|) + 4e-8 -
    order-one operands 1-2e-6, logits of 400 5e-5 |v| (their own fp32 resolution), values of 1e-3 2.3e-8 absolute
    (2e-5 relative: the denormal floor) - and a dynamic layer selects the fp64 keys up to near-ties at that resolution."""
    rs = np.random.RandomState(N + topk + len(case))
    qkv = rs.standard_normal((2, 2 * N, 3, 4, 32))
    sq, sk, sv = {'small_q_large_k': (1e-3 * 32 ** 0.5, 30.0, 1.0), 'small_v': (1.3, 1.3, 1e-3),
                  'small_everything': (2e-2, 2e-2, 1e-3), 'large': (8.0, 8.0, 100.0)}[case]
    qkv[:, :, 0] *= sq
    qkv[:, :, 1] *= sk
    qkv[:, :, 2] *= sv
    qkv = torch.from_numpy(qkv)
    out = ops.attention(qkv.to(DEV), N, N, False, topk=topk).cpu().double()
    for lo, hi in ((0, N), (N, 2 * N)):
        q, kk, v = (qkv[:, lo:hi, i].permute(0, 3, 2, 1) for i in range(3))
        logits = torch.einsum('bdhn,bdhm->bhnm', q, kk) / 32 ** 0.5
        Lmax = float(logits.abs().max())
        if topk:
            ref, _ = O.dynamic_attention(q, kk, v, topk)
            top = logits.topk(topk + 1, dim=3).values
            # not a near-tie at the resolution of the logits: 2^-22 relative, plus the absolute floor of the denormal
            # residuals of tiny queries (2^-25 per term) times the size of the keys
            res = 5e-7 * max(Lmax, 0.2) + 2e-7 * float(kk.abs().max())
            ok = ((top[..., topk - 1] - top[..., topk]) >= res).permute(0, 2, 1)                       # [B, n, H]
            assert ok.double().mean() > 0.9
        else:
            ref, _ = O.attention(q, kk, v)
            ok = torch.ones(2, N, 4, dtype=torch.bool)
        err = (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().reshape(2, N, 4, 32).amax(3)
        tol = sv * (3e-6 + 1.5e-7 * Lmax) + 4e-8
        assert err[ok].max() < tol, (case, float(err[ok].max()), tol)


@pytest.mark.parametrize('N,M,k', [(64, 64, 16), (64, 64, 1), (64, 64, 63), (40, 56, 8), (512, 512, 128), (512, 512, 64),
                                   (256, 256, 128), (100, 70, 70), (48, 64, 16), (300, 500, 64), (1024, 1024, 128),
                                   (2048, 2048, 64), (700, 600, 100), (1500, 520, 64), (513, 513, 512)])
def test_attention_topk(N, M, k):
    rs = np.random.RandomState(N + 13 * M + k)
    B = 2 if N <= 1024 else 1
    qkv = torch.from_numpy(rs.standard_normal((B, N + M, 3, 4, 32)) * 1.3)
    out = ops.attention(qkv.to(DEV), N, M, False, topk=k).cpu().double()
    for lo, hi in ((0, N), (N, N + M)):
        q = qkv[:, lo:hi, 0].permute(0, 3, 2, 1)
        kk = qkv[:, lo:hi, 1].permute(0, 3, 2, 1)
        v = qkv[:, lo:hi, 2].permute(0, 3, 2, 1)
        ref, _ = O.dynamic_attention(q, kk, v, k)
        err = (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs()                      # [B, n, 128]
        # A row whose k-th and (k+1)-th largest logits are closer than fp32 can resolve may legitimately select
        # the other key (the reference runs in fp64): such rows are excluded, and there must be almost none.
        logits = torch.einsum('bdhn,bdhm->bhnm', q, kk) / 32 ** 0.5
        n_keys = logits.shape[-1]
        if k < n_keys:
            top = logits.topk(k + 1, dim=3).values
            near_tie = (top[..., k - 1] - top[..., k]) < 5e-6                   # [B, H, n]
        else:
            near_tie = torch.zeros(logits.shape[:3], dtype=torch.bool)
        assert near_tie.double().mean() < 2e-3
        ok = ~near_tie.permute(0, 2, 1)                                         # [B, n, H]
        err_h = err.reshape(err.shape[0], err.shape[1], 4, 32).amax(3)          # per (row, head)
        assert err_h[ok].max() < 1e-5, (lo, err_h[ok].max())


def test_attention_topk_golden_and_errors(golden_dir):
    g = _g(golden_dir, 'op_vectors')
    q, k, v = (torch.from_numpy(g[x]) for x in ('att_q', 'att_k', 'att_v'))   # q: 40 queries, k/v: 56 keys
    # library layout needs both frames: frame 0 = the 40 queries (with dummy k/v), frame 1 = the 56 keys
    B, N, M = q.shape[0], q.shape[3], k.shape[3]
    qkv = torch.zeros(B, N + M, 3, 4, 32, dtype=torch.float64)
    qkv[:, :N, 0] = q.permute(0, 3, 2, 1)
    qkv[:, N:, 1] = k.permute(0, 3, 2, 1)
    qkv[:, N:, 2] = v.permute(0, 3, 2, 1)
    full = ops.attention(qkv.to(DEV), N, M, True).cpu().double()[:, :N]
    assert (full - _ref_msg_to_lib(torch.from_numpy(g['att_full']))).abs().max() < 1e-5
    for kk in (1, 8):
        dyn = ops.attention(qkv.to(DEV), N, M, True, topk=kk).cpu().double()[:, :N]
        assert (dyn - _ref_msg_to_lib(torch.from_numpy(g[f'att_dyn{kk}']))).abs().max() < 1e-5
    with pytest.raises(RuntimeError, match='exceeds the number of keys'):
        ops.attention(qkv.to(DEV), N, M, True, topk=57)       # torch.topk raises in the reference
    with pytest.raises(RuntimeError, match='2048'):
        ops.attention(torch.zeros(1, 4200, 3, 4, 32, device=DEV), 2100, 2100, False, topk=8)   # > 2048 keys: dynamic unsupported


@pytest.mark.parametrize('N,k,dups', [(64, 16, 2), (512, 64, 3), (512, 128, 2), (256, 128, 5), (1024, 128, 4), (2048, 64, 3), (100, 30, 2)])
def test_attention_topk_with_ties(N, k, dups):
    """Duplicated keys give exactly equal logits.  Where such a group straddles the k-th place `torch.topk` keeps an
    unspecified subset of it; the kernel keeps EXACTLY k keys too and breaks the tie towards the lowest key indices.
    Checked through the selection tap: every row holds k keys, the kept members of a tied group are its first ones,
    and with that selection forced the oracle agrees to 1e-5 on every row (rows whose selection the tie does not touch
    are the oracle's own)."""
    rs = np.random.RandomState(5 + N + k)
    M = N
    qkv = torch.from_numpy(rs.standard_normal((1, N + M, 3, 4, 32)))
    group = [10 + 7 * i for i in range(dups)]                 # keys 10, 17, 24, ... of frame 0 identical
    for g in group[1:]:
        qkv[:, g, 1:] = qkv[:, group[0], 1:]
    out, (sel0, sel1) = ops.attention(qkv.to(DEV), N, M, False, topk=k, return_selection=True)
    # the kernels without the tap find the surplus after the softmax pass and take the dropped key out again (or redo
    # the pass): the same result as the tap build up to the rounding of that correction
    assert (out - ops.attention(qkv.to(DEV), N, M, False, topk=k)).abs().max() < 1e-6
    out = out.cpu().double()
    sel0, sel1 = sel0.cpu(), sel1.cpu()
    assert torch.isfinite(out).all()
    assert (sel1.sum(-1) == k).all() and (sel0.sum(-1) == k).all()      # exactly k everywhere
    kept = sel0[..., group].long()                                       # [1, 4, N, dups]
    assert (kept[..., :-1] >= kept[..., 1:]).all()                       # a kept member never follows a dropped one
    partial = (kept.sum(-1) > 0) & (kept.sum(-1) < dups)
    assert partial.any()                                                 # the straddling case is exercised
    q = qkv[:, :N, 0].permute(0, 3, 2, 1)
    kk = qkv[:, :N, 1].permute(0, 3, 2, 1)
    v = qkv[:, :N, 2].permute(0, 3, 2, 1)
    rep = []
    ref, _ = O.dynamic_attention(q, kk, v, k, forced=sel0, report=rep)
    assert (out[:, :N] - _ref_msg_to_lib(ref)).abs().max() < 1e-5
    assert rep[0]['bad_count'] == 0 and rep[0]['max_gap'] < 5e-6
    ref_own, _ = O.dynamic_attention(q, kk, v, k)
    err = (out[:, :N] - _ref_msg_to_lib(ref_own)).abs().reshape(1, N, 4, 32).amax(3)       # [1, N, H]
    assert err[(~partial).permute(0, 2, 1)].max() < 1e-5


def test_attention_topk_all_keys_equal():
    """Every logit of a row equal (identical keys): k or more ties AT THE MAXIMUM - the first k keys stay."""
    rs = np.random.RandomState(2)
    for N, k in ((64, 16), (512, 64), (1024, 128)):
        qkv = torch.from_numpy(rs.standard_normal((1, 2 * N, 3, 4, 32)))
        qkv[:, :, 1] = qkv[:, :1, 1]
        out, (sel0, sel1) = ops.attention(qkv.to(DEV), N, N, False, topk=k, return_selection=True)
        assert (out - ops.attention(qkv.to(DEV), N, N, False, topk=k)).abs().max() < 1e-6
        for sel in (sel0.cpu(), sel1.cpu()):
            assert (sel.sum(-1) == k).all()
            assert sel[..., :k].all() and not sel[..., k:].any()
        v = qkv[:, :N, 2].permute(0, 3, 2, 1)
        ref = v[..., :k].mean(-1, keepdim=True).expand(-1, -1, -1, N)          # uniform softmax over the first k values
        assert (out[:, :N].cpu().double() - _ref_msg_to_lib(ref)).abs().max() < 1e-5


@pytest.mark.parametrize('n,k', [(512, 64), (512, 128), (2048, 64), (1024, 128), (256, 128), (100, 30)])
@pytest.mark.parametrize('dist', ['bimodal', 'outliers', 'lognormal', 'cauchy', 'two_clusters_at_k'])
def test_attention_topk_count_on_hard_distributions(n, k, dist):
    """The threshold search must end with EXACTLY k kept keys whatever the shape of a logit row (no ties here): sharply
    bimodal rows, a few huge outliers, heavy tails.  Its density / interpolation steps converge only linearly on such
    rows; the interleaved ordinal bisection bounds the search (without it the probe cap was hit and nearly
    all keys were kept).  Checked through the selection tap; the message is compared with the oracle as well."""
    rs = np.random.RandomState(n + k + len(dist))
    B = 1
    if dist == 'bimodal':            # 10 % of the keys near 8 +- 1, the rest near -2 +- 0.05
        x = np.where(rs.uniform(size=(B, 2 * n)) < 0.1, 8 + rs.standard_normal((B, 2 * n)), -2 + 0.05 * rs.standard_normal((B, 2 * n)))
    elif dist == 'outliers':
        x = 0.01 * rs.standard_normal((B, 2 * n))
        x[:, rs.choice(2 * n, 12, replace=False)] = rs.uniform(-300, 300, 12)
    elif dist == 'lognormal':
        x = np.exp(1.5 * rs.standard_normal((B, 2 * n)))
    elif dist == 'cauchy':
        x = np.clip(rs.standard_cauchy((B, 2 * n)), -500, 500)
    else:                            # the k-th place falls inside a tight cluster far from the mean
        x = np.where(np.arange(2 * n)[None] % n < k - 3, 20 + rs.standard_normal((B, 2 * n)), 1e-3 * rs.standard_normal((B, 2 * n)))
    qkv = np.zeros((B, 2 * n, 3, 4, 32))
    qkv[:, :, 2] = rs.standard_normal((B, 2 * n, 4, 32))
    qkv[:, :, 1, :, 0] = x[:, :, None]                               # key component = the prescribed distribution
    qkv[:, :, 1, :, 1] = 0.3 * rs.standard_normal((B, 2 * n, 4))      # + a second component so that rows differ
    qkv[:, :, 0, :, 0] = np.sqrt(32) * rs.uniform(0.5, 2.0, (B, 2 * n, 4)) * np.where(rs.uniform(size=(B, 2 * n, 4)) < 0.2, -1, 1)
    qkv[:, :, 0, :, 1] = np.sqrt(32) * rs.uniform(-1, 1, (B, 2 * n, 4))
    qkv = torch.from_numpy(qkv)
    out, (sel0, sel1) = ops.attention(qkv.to(DEV), n, n, False, topk=k, return_selection=True)
    cnt = torch.cat([sel0.sum(-1), sel1.sum(-1)], dim=2).cpu()
    assert (cnt == k).all(), (dist, int(cnt.min()), int(cnt.max()))
    out = out.cpu().double()
    for lo, hi, sel in ((0, n, sel0), (n, 2 * n, sel1)):
        q = qkv[:, lo:hi, 0].permute(0, 3, 2, 1)
        kk = qkv[:, lo:hi, 1].permute(0, 3, 2, 1)
        v = qkv[:, lo:hi, 2].permute(0, 3, 2, 1)
        rep = []
        ref, _ = O.dynamic_attention(q, kk, v, k, forced=sel.cpu(), report=rep)
        scale = max(1.0, float(np.abs(x).max()) * 2.5 / 10.0)               # logits up to ~1000 here: fp32 resolution 6e-5
        assert (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max() < 2e-5 * scale
        assert rep[0]['max_gap'] < 5e-6 * scale and rep[0]['rows'] <= 4, rep[0]


@pytest.mark.parametrize('n,k', [(512, 128), (512, 64), (256, 128), (128, 64), (1024, 128)])
@pytest.mark.parametrize('case', ['zero_keys', 'all_negative', 'tiny'])
def test_attention_topk_threshold_at_zero_and_below(n, k, case):
    """The kernels test "logit >= threshold" as clamp((s - t') 2^100) with t' the float below t (attention.hip, ge_ind).
    Edge cases of that form: the k-th place falls into a block of logits that are EXACTLY zero (zero key vectors:
    threshold 0, its predecessor is a denormal - the guard keeps t' 2^-90 away), every logit negative (predecessor of a
    negative float is one step further from zero), and logits of magnitude 1e-30 (differences far below one but far
    above 2^-100).  Exactly k keys per row, ties towards the lowest indices, message equal to the oracle's with that
    selection forced."""
    rs = np.random.RandomState(n + k + len(case))
    qkv = torch.from_numpy(rs.standard_normal((1, 2 * n, 3, 4, 32)))
    if case == 'zero_keys':
        # fewer than k keys have positive logits for most rows: all but k // 2 keys of each frame are zero vectors
        keep = k // 2
        for base in (0, n):
            zero = np.setdiff1d(np.arange(n), rs.choice(n, keep, replace=False))
            qkv[:, base + zero, 1] = 0.0
    elif case == 'all_negative':
        qkv[:, :, 0, :, 0] = 40.0                              # q . k dominated by -40 * |k_0|
        qkv[:, :, 1, :, 0] = -qkv[:, :, 1, :, 0].abs() - 0.5
    else:
        qkv[:, :, 1] *= 1e-30
    out, (sel0, sel1) = ops.attention(qkv.to(DEV), n, n, False, topk=k, return_selection=True)
    # the kernels without the tap keep every tied key in their pass and take the surplus out of the written rows again, one
    # key at a time: with HUNDREDS of zero keys tied at the k-th place that correction is applied hundreds of times per row
    # and its rounding adds up (measured 1.7e-5; a handful of ties, the realistic case: < 1e-6, test_attention_topk_with_ties)
    assert (out - ops.attention(qkv.to(DEV), n, n, False, topk=k)).abs().max() < (5e-5 if case == 'zero_keys' else 1e-6)
    out = out.cpu().double()
    for lo, hi, sel in ((0, n, sel0.cpu()), (n, 2 * n, sel1.cpu())):
        assert (sel.sum(-1) == k).all(), (case, int(sel.sum(-1).min()), int(sel.sum(-1).max()))
        q, kk, v = (qkv[:, lo:hi, i].permute(0, 3, 2, 1) for i in range(3))
        rep = []
        ref, _ = O.dynamic_attention(q, kk, v, k, forced=sel, report=rep)
        assert (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max() < 1e-5
        assert rep[0]['bad_count'] == 0 and rep[0]['max_gap'] < 5e-6, rep[0]
        if case == 'zero_keys':
            # rows that need zero-logit keys to reach k take the lowest-indexed ones: the kept zero keys of a row form a
            # prefix of the zero keys (in index order)
            zmask = (qkv[0, lo:hi, 1].abs().sum((1, 2)) == 0)                       # [n]
            kept_zero = sel[0][:, :, zmask]                                         # [H, n, zeros]
            assert (kept_zero[..., :-1].long() >= kept_zero[..., 1:].long()).all()
            assert kept_zero.any()


@pytest.mark.parametrize('n', [256, 512, 1024])
def test_attention_topk_unmeasured_bracket_ends(n):
    # Shapes whose rows fill whole 32-key blocks seed the threshold search from sub-sampled statistics and an
    # assumed count at the maximum; these rows force the search to measure both ends after all:
    #  (a) k-th value far below mean - 8 sd (three outlier keys, k drops only two of them),
    #  (b) more than k keys tied at the maximum (identical keys AND values: any k of them give the same message).
    rs = np.random.RandomState(n)
    qkv = torch.from_numpy(rs.standard_normal((2, 2 * n, 3, 4, 32)))
    qkv[:, :, 0, :, 0] = 5.0
    a = qkv.clone()
    a[:, [3, 77, n - 1], 1, :, 0] = -200.0
    a[:, [n + 5, n + 64, 2 * n - 2], 1, :, 0] = -200.0
    b = qkv.clone()
    for base in (0, n):
        b[:, base + 20:base + 30, 1:] = b[:, base + 20:base + 21, 1:]
        b[:, base + 20:base + 30, 1, :, 0] = 200.0
    for x, k in ((a, n - 2), (b, 8)):
        out = ops.attention(x.to(DEV), n, n, False, topk=k).cpu().double()
        for lo, hi in ((0, n), (n, 2 * n)):
            q, kk, v = (x[:, lo:hi, i].permute(0, 3, 2, 1) for i in range(3))
            ref, _ = O.dynamic_attention(q, kk, v, k)
            assert (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max() < 1e-5


@pytest.mark.parametrize('tag', ['sk_7x5', 'sk_64x64', 'sk_48x64'])
def test_sinkhorn_golden(golden_dir, tag):
    g = _g(golden_dir, 'op_vectors')
    iters, alpha = g[tag + '_meta']
    Z = ops.sinkhorn(torch.from_numpy(g[tag + '_scores']).to(DEV), float(alpha), int(iters)).cpu().double().numpy()
    assert np.abs(Z - g[tag + '_Z']).max() < 1e-4


def test_sinkhorn_512_golden(golden_dir):
    g = _g(golden_dir, 'op_vectors')
    iters, alpha = g['sk_512x512_meta']
    s = torch.from_numpy(np.random.RandomState(int(g['sk_512x512_seed'][0])).standard_normal((1, 512, 512)) * 3.0)
    Z = ops.sinkhorn(s.to(DEV), float(alpha), int(iters)).cpu().double().numpy()
    assert np.abs(Z[:, ::8, ::8] - g['sk_512x512_Z_sub']).max() < 1e-4
    assert np.abs(Z[:, -1, :] - g['sk_512x512_Z_lastrow']).max() < 1e-4
    assert np.abs(Z[:, :, -1] - g['sk_512x512_Z_lastcol']).max() < 1e-4


@pytest.mark.parametrize('N,M,iters', [(1, 1, 3), (3, 200, 10), (130, 65, 25), (256, 256, 20), (300, 513, 30), (700, 900, 10),
                                       (64, 64, 0)])
def test_sinkhorn_vs_oracle(N, M, iters):
    rs = np.random.RandomState(N * 3 + M)
    s = torch.from_numpy(rs.standard_normal((3, N, M)) * 4.0)
    ref = O.log_optimal_transport(s, 0.7, iters)
    for streaming in (False, True):     # cluster kernel (N, M <= 512) and streaming kernel
        Z = ops.sinkhorn(s.to(DEV), 0.7, iters, streaming=streaming).cpu().double()
        assert (Z - ref).abs().max() < 1e-4, streaming


@pytest.mark.parametrize('N,M,iters,scale', [(512, 128, 7, 30.0), (512, 512, 3, 40.0), (300, 200, 20, 25.0), (1024, 512, 5, 30.0)])
def test_sinkhorn_wide_dynamic_range(N, M, iters, scale):
    """Scores spanning ~100 units and few iterations: entries of K = exp2(s - row maximum) fall INTO THE DENORMALS (and the
    scalings fold them further down); the epilogue recovers the scores from log2 K, and v_log_f32 of a denormal is -inf
    (found by tools/fuzz_forward.py: 7 entries of Z at -inf where the reference has -60).  Z must be finite and within
    1e-4 of the oracle everywhere, for the cluster and the streaming kernel."""
    rs = np.random.RandomState(N + M + iters)
    s = torch.from_numpy(rs.standard_normal((2, N, M)) * scale + 50.0)
    ref = O.log_optimal_transport(s, 0.37, iters)
    for streaming in (False, True):
        Z = ops.sinkhorn(s.to(DEV), 0.37, iters, streaming=streaming).cpu().double()
        assert torch.isfinite(Z).all(), streaming
        assert (Z - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()) / 100), (streaming, float((Z - ref).abs().max()))


@pytest.mark.parametrize('B,N,M,wide', [(5, 512, 512, (1,)), (9, 300, 400, (0, 7)), (3, 1024, 700, (2,)), (66, 512, 512, (13, 40))])
def test_sinkhorn_range_fallback_is_per_pair(B, N, M, wide):
    """A pair whose scores are beyond the range of the scaling form is handed to the log-domain kernel - THAT pair, not the
    launch it happens to share with others: every pair of a mixed batch gets, bit for bit, what it gets alone (what a pair
    returns must not depend on its batch: tools/fuzz_forward.py found slicings of one batch differing by 1e-5 on Z when the
    whole launch was redone), wide pairs equal the streaming kernel's result, and everything is within the bar of the oracle."""
    g = torch.Generator(DEV).manual_seed(B + N + M)
    s = torch.randn(B, N, M, device=DEV, generator=g) * 3
    for w in wide:
        s[w] = s[w] * 14 + 50.0                    # spread over ~250 units: far beyond 100 octaves below the row maximum
    Z = ops.sinkhorn(s, 0.8, 20)
    Zs = ops.sinkhorn(s, 0.8, 20, streaming=True)
    torch.cuda.synchronize()
    assert torch.isfinite(Z).all()
    for b in sorted(set(wide) | {0, B - 1, B // 2}):
        alone = ops.sinkhorn(s[b:b + 1].contiguous(), 0.8, 20)
        assert torch.equal(alone[0], Z[b]), b
        if b in wide:
            assert torch.equal(Z[b], Zs[b]), b      # redone by the streaming kernel
        else:
            assert not torch.equal(Z[b], Zs[b]) and (Z[b] - Zs[b]).abs().max() < 1e-4, b    # the cluster kernel's own result
    some = sorted(set(wide) | {B - 1})
    ref = O.log_optimal_transport(s[some].cpu().double(), 0.8, 20)
    got = Z[some].cpu().double()
    assert (got - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()) / 100)


@pytest.mark.parametrize('B,N,M', [(3, 512, 512), (70, 300, 512), (2, 1024, 700), (1, 2048, 2048)])
def test_sinkhorn_fallback_after_a_lost_partner(B, N, M, monkeypatch):
    """The cluster kernel's workgroups wait for their partners with bounded spins; a workgroup that gives up raises the
    launch's error word and the gated streaming kernel launched behind every cluster launch redoes the launch.  The hook
    MDGAT_SK_FORCE_FALLBACK=1 sets that word up front, so the gated kernel and - through mdgat_forward - the extraction
    from its Z run for real: same Z as the streaming kernel on its own (bit for bit: it IS that kernel), 1e-4 from the
    oracle, and the cluster path again once the hook is off."""
    s = torch.randn(B, N, M, device=DEV, generator=torch.Generator(DEV).manual_seed(N + M)) * 3
    Zs = ops.sinkhorn(s, 0.8, 25, streaming=True)
    Zc = ops.sinkhorn(s, 0.8, 25)
    monkeypatch.setenv('MDGAT_SK_FORCE_FALLBACK', '1')
    Zf = ops.sinkhorn(s, 0.8, 25)
    torch.cuda.synchronize()
    monkeypatch.delenv('MDGAT_SK_FORCE_FALLBACK')
    assert torch.equal(Zf, Zs)
    assert not torch.equal(Zc, Zs) and (Zc - Zs).abs().max() < 1e-4
    assert torch.equal(ops.sinkhorn(s, 0.8, 25), Zc)
    if N <= 1024:
        ref = O.log_optimal_transport(s.cpu().double(), 0.8, 25)
        assert (Zf.cpu().double() - ref).abs().max() < 1e-4


@pytest.mark.parametrize('B,N,M', [(1, 512, 512), (70, 512, 512), (5, 200, 300), (9, 500, 37), (3, 128, 512), (130, 129, 64),
                                   (2, 512, 512), (7, 512, 512), (12, 512, 512), (13, 300, 512), (60, 512, 100), (3, 1000, 400),
                                   (5, 2048, 300), (2, 2048, 2048)])
def test_sinkhorn_cluster_matches_streaming(B, N, M):
    """The two kernels implement the same iteration: they must agree to fp32 round-off for any batch size
    (more pairs than resident groups: the persistent loop; ragged shapes: the masks; batches that are not multiples of 8:
    the grid padded to 8 groups for the XCD placement, surplus workgroups leaving at once; 2048 x 2048: no placement)."""
    s = torch.randn(B, N, M, device=DEV, generator=torch.Generator(DEV).manual_seed(B + N)) * 3
    Zc = ops.sinkhorn(s, 1.0, 30)
    Zs = ops.sinkhorn(s, 1.0, 30, streaming=True)
    assert torch.isfinite(Zc).all()
    assert (Zc - Zs).abs().max() < 2e-5


def test_sinkhorn_marginals_full_size():
    # size-independent property at the bench shape: after the last v-update every column of exp(Z) sums to
    # its marginal exactly (1 for keypoint columns, N for the dustbin column) and rows do approximately
    B, N, M = 8, 512, 512
    s = torch.randn(B, N, M, device=DEV, generator=torch.Generator(DEV).manual_seed(1)) * 3
    Z = ops.sinkhorn(s, 1.0, 100).double()
    col = torch.logsumexp(Z, dim=1)
    assert col[:, :M].abs().max() < 1e-4
    assert (col[:, M] - np.log(N)).abs().max() < 1e-4
    row = torch.logsumexp(Z, dim=2)
    assert row[:, :N].abs().max() < 5e-2


@pytest.mark.parametrize('name', ['fwd_n64_L4_S20', 'fwd_n64_L5_S20', 'fwd_n48m64_L4_S20'])
def test_extract_golden(golden_dir, name):
    g = _g(golden_dir, name)
    Z32 = torch.from_numpy(g['Z']).float()
    for tag, mode in (('default', _lib.EXTRACT_DUSTBIN), ('mutual', _lib.EXTRACT_DUSTBIN_MUTUAL),
                      ('sg', _lib.EXTRACT_THRESHOLD), ('sgmutual', _lib.EXTRACT_THRESHOLD_MUTUAL)):
        if f'{tag}_matches0' not in g:
            continue
        m0, m1, s0, s1 = ops.extract(Z32.to(DEV), mode, 0.2)
        # the oracle on the SAME fp32-rounded Z must agree bit-exactly on indices
        lm, mc = {'default': ('triplet_loss', False), 'mutual': ('triplet_loss', True), 'sg': ('superglue', False),
                  'sgmutual': ('superglue', True)}[tag]
        r0, r1, rs0, rs1 = O.extract_matches(Z32.double(), lm, mc, 0.2)
        assert torch.equal(m0.cpu(), r0) and torch.equal(m1.cpu(), r1), tag
        assert (s0.cpu().double() - rs0).abs().max() < 1e-6 and (s1.cpu().double() - rs1).abs().max() < 1e-6, tag
        # and with the reference's own outputs
        np.testing.assert_array_equal(m0.cpu().numpy(), g[f'{tag}_matches0'])
        np.testing.assert_array_equal(m1.cpu().numpy(), g[f'{tag}_matches1'])
        assert np.abs(s0.cpu().numpy() - g[f'{tag}_mscores0']).max() < 1e-5


def test_extract_all_dustbin_and_ties(golden_dir):
    g = _g(golden_dir, 'edge_cases')
    Z32 = torch.from_numpy(g['alldust_Z']).float()
    m0, m1, s0, s1 = ops.extract(Z32.to(DEV), _lib.EXTRACT_DUSTBIN, 0.2)
    np.testing.assert_array_equal(m0.cpu().numpy(), g['alldust_matches0'])
    np.testing.assert_array_equal(m1.cpu().numpy(), g['alldust_matches1'])
    assert (s0 == 0).all() and (s1 == 0).all()          # mdgat.py:465-467: all-zero scores
    # first-index tie rule of torch.max: constant rows pick column 0, constant columns pick row 0
    Zc = torch.zeros(1, 9, 7)
    m0, m1, s0, s1 = ops.extract(Zc.to(DEV), _lib.EXTRACT_DUSTBIN, 0.2)
    assert (m0 == 0).all() and (m1 == 0).all()
    r0, r1, _, _ = O.extract_matches(Zc.double())
    assert torch.equal(m0.cpu(), r0) and torch.equal(m1.cpu(), r1)


def test_knn_golden(golden_dir):
    g = _g(golden_dir, 'op_vectors')
    for Cc in (3, 128):
        x, s = torch.from_numpy(g[f'knn{Cc}_x']), torch.from_numpy(g[f'knn{Cc}_s'])
        idx, adj = ops.knn(x.to(DEV), s.to(DEV), 9, adjacency=True)
        np.testing.assert_array_equal(idx.cpu().numpy(), g[f'knn{Cc}_idx'])
        np.testing.assert_array_equal(adj.cpu().numpy(), g[f'knn{Cc}_adj'])
    with pytest.raises(RuntimeError):
        ops.knn(x.to(DEV), s.to(DEV), 71)


def _knn_check(x, s, k, eps, **kw):
    """GPU kNN against the fp64 oracle: index lists identical on every row whose k + 1 nearest neighbours are separated
    by more than eps in the oracle's fp64 distances (closer than fp32-class arithmetic resolves: either order is right)."""
    idx, adj = ops.knn(x.to(DEV), s.to(DEV), k, adjacency=True, **kw)
    ref = O.knn(x, s, k)
    inner = -2.0 * torch.matmul(x.transpose(2, 1), s)
    nd = -(x ** 2).sum(1, keepdim=True).transpose(2, 1) - inner - (s ** 2).sum(1, keepdim=True)
    top = nd.topk(min(k + 1, nd.shape[-1]), dim=-1).values
    # (fp32 resolution grows with the magnitude of the distances: eps is meant at |d^2| <= 100)
    tol = eps * torch.clamp(top[..., 1:].abs() / 100.0, min=1.0)
    clear = ((top[..., :-1] - top[..., 1:]) > tol).all(-1) if top.shape[-1] > 1 else torch.ones(top.shape[:2], dtype=torch.bool)
    frac = 1.0 - clear.double().mean().item()
    assert torch.equal(idx.cpu()[clear], ref[clear]), 'neighbour lists differ on rows without near-ties'
    # every row: the same SET up to near-ties is implied above; the adjacency is the scatter of the returned indices
    assert torch.equal(adj.cpu(), O.knn_adjacency(x, s, k, idx=idx.cpu()))
    assert (adj.sum(-1) == k).all()
    return frac


@pytest.mark.parametrize('C,N,M,k,eps', [(3, 2048, 2048, 9, 1e-4), (3, 2048, 2048, 64, 1e-4), (128, 2048, 2048, 9, 2e-4),
                                         (128, 2048, 2048, 64, 2e-4), (3, 300, 5000, 16, 1e-4), (128, 130, 9000, 40, 2e-4),
                                         (3, 100, 4097, 1, 1e-4), (7, 257, 600, 600, 1e-4), (3, 64, 1500, 1024, 1e-4),
                                         (128, 512, 512, 128, 2e-4)])
def test_knn_large(C, N, M, k, eps):
    """BASELINE configs[4] size (N = M = 2048; coordinates C = 3 and feature space C = 128 on the matrix cores), rows
    longer than one LDS segment (M > 4096: merged best-k lists), k = M, k = 1, the largest k."""
    rs = np.random.RandomState(C + N + M + k)
    scale = 20.0 if C == 3 else 1.0
    x = torch.from_numpy(scale * rs.standard_normal((2 if N <= 512 else 1, C, N)))
    s = torch.from_numpy(scale * rs.standard_normal((x.shape[0], C, M)))
    frac = _knn_check(x, s, k, eps)
    print(f'knn C={C} N={N} M={M} k={k}: rows excluded as near-ties {frac:.3%}')
    assert frac <= min(1.0, 0.05 + 0.004 * k)           # (k adjacent gaps per row can be a near-tie)
    if C == 128:        # the same rows through the path without the matrix-core workspace
        _knn_check(x[:, :, :64], s, k, eps, mfma=False)


def test_knn_exact_ties_lowest_index_first():
    """Duplicated source points: equal distances - neighbours in the order of their indices, k exactly."""
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.standard_normal((1, 3, 40)))
    s = torch.from_numpy(rs.standard_normal((1, 3, 200)))
    s[:, :, 150:] = s[:, :, :50]                       # points 150.. duplicate points 0..49
    idx = ops.knn(x.to(DEV), s.to(DEV), 7).cpu()
    srcf = s.float()
    d = ((x.float()[:, :, :, None] - srcf[:, :, None, :]) ** 2).sum(1)            # [1, 40, 200], the kernel's fp32 form
    got = d.gather(2, idx)
    assert (got[..., 1:] >= got[..., :-1]).all()                                   # nearest first
    same = got[..., 1:] == got[..., :-1]
    assert same.any() and (idx[..., 1:][same] > idx[..., :-1][same]).all()          # ties: ascending index
    kth = got[..., -1:]
    assert ((d < kth).sum(-1) <= 7).all() and ((d <= kth).sum(-1) >= 7).all()


def test_mfma_sustained_probe():
    """bench.py's ceiling measurement (mdgat_mfma_probe): a positive rate below the dense peak at 2.4 GHz, a plausible clock,
    and about 16 cycles per 16x16x32 MFMA and SIMD (two waves take turns on the pipe)."""
    r = ops.mfma_sustained(DEV, reps=500)
    assert 200.0 < r['tflops'] < 2600.0, r
    assert 0.5 < r['clock_ghz'] < 2.6, r
    assert 15.5 < r['ticks_per_mfma_per_simd'] < 24.0, r

