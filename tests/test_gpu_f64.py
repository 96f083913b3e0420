"""GPU parity of the reference-exact mode (csrc/f64.hip) through the C ABI - what a float64 module runs (net.double(), as the
reference's callers do before every forward; config['arithmetic'] = 'fp64' pins it whatever the dtype).

The point of the mode is the LITERAL north-star bar: Z within 1e-4 of the reference's fp64 output on every pair, with no
attribution of top-k flips - because no selection flips: the encoders and the layers through the last dynamic one compute in
fp64 from fp64 inputs and weights, so ``logits.topk(k)`` (mdgat.py:202) selects what the reference selects.  The tests compare
with the REFERENCE's own outputs (tests/golden/cfg_*.npz, made by tools/make_goldens.py from the imported reference) at the
literal tolerance, and with the oracle (pinned to the reference by tests/test_oracle_golden.py) op by op."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mdgat_matcher_amd import MDGAT, _lib, ops, synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402

DEV = 'cuda:0'
Z_TOL = 1e-4            # north star, literal
F64_TOL = 1e-11         # the fp64 kernels against torch fp64 on the same operands (summation order only)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('rows,cout,K,relu,res', [(100, 32, 4, True, False), (130, 64, 33, True, False), (257, 128, 64, True, False),
                                                  (1024, 384, 128, False, False), (300, 256, 256, True, False),
                                                  (777, 128, 256, False, True), (64, 128, 128, False, False), (1, 5, 3, False, True)])
def test_pointwise_f64(rows, cout, K, relu, res):
    rs = np.random.RandomState(rows + cout + K)
    a = torch.from_numpy(rs.standard_normal((rows, K)))
    w = torch.from_numpy(rs.standard_normal((cout, K)) / np.sqrt(K))
    b = torch.from_numpy(rs.standard_normal(cout))
    r = torch.from_numpy(rs.standard_normal((rows, cout))) if res else None
    out = ops.pointwise_f64(a.to(DEV), w.to(DEV), b.to(DEV), relu, r.to(DEV) if res else None).cpu()
    ref = a @ w.T + b
    if relu:
        ref = torch.relu(ref)
    if res:
        ref = ref + r
    assert (out - ref).abs().max() < F64_TOL


def _ref_msg_to_lib(msg):
    b, dh, h, n = msg.shape
    return msg.permute(0, 3, 2, 1).reshape(b, n, h * dh)


def _sides(qkv, N, M, cross):
    """(query rows, source rows) per frame, in the oracle's [B, dh, H, n] layout"""
    fr = ((0, N), (N, N + M))
    for side in range(2):
        lo, hi = fr[side]
        slo, shi = fr[1 - side] if cross else fr[side]
        yield (lo, hi), qkv[:, lo:hi, 0].permute(0, 3, 2, 1), qkv[:, slo:shi, 1].permute(0, 3, 2, 1), qkv[:, slo:shi, 2].permute(0, 3, 2, 1)


@pytest.mark.parametrize('N,M', [(64, 64), (40, 56), (512, 512), (257, 130), (33, 31), (2048, 1300), (16, 16), (5, 3)])
@pytest.mark.parametrize('cross', [False, True])
@pytest.mark.parametrize('form', [0, 1])
def test_attention_f64_full(N, M, cross, form):
    """Both launch forms of the full-attention kernel (mdgat_set_f64_attention_form: 0 = the keys of a query tile split over the four
    waves, what launches of this size get by default; 1 = one wave per 32 queries, what batches of 32 pairs of 512 keypoints get)."""
    from mdgat_matcher_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(N + 7 * M + cross)
    B = 2 if N <= 512 else 1
    qkv = torch.from_numpy(rs.standard_normal((B, N + M, 3, 4, 32)) * 1.3)
    lib.mdgat_set_f64_attention_form(form)
    try:
        out = ops.attention_f64(qkv.to(DEV), N, M, cross).cpu()
    finally:
        lib.mdgat_set_f64_attention_form(-2)
    for (lo, hi), q, k, v in _sides(qkv, N, M, cross):
        ref, _ = O.attention(q, k, v)
        assert (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max() < F64_TOL


@pytest.mark.parametrize('N,M,k,cross', [(64, 64, 16, False), (64, 64, 1, False), (64, 64, 63, True), (40, 56, 8, True), (512, 512, 128, False),
                                         (512, 512, 64, True), (256, 256, 128, False), (100, 70, 70, False), (48, 64, 16, True),
                                         (300, 500, 64, True), (1024, 1024, 128, False), (2048, 2048, 64, False), (700, 600, 100, False),
                                         (1500, 520, 64, True), (513, 513, 512, False), (17, 17, 16, False)])
def test_attention_f64_topk_selects_like_fp64(N, M, k, cross):
    """No near-tie rows excluded (tests/test_gpu_ops.py::test_attention_topk has to): the kept keys are torch.topk's on fp64
    logits, row by row, and the message agrees to fp64 rounding."""
    rs = np.random.RandomState(N + 13 * M + k)
    B = 2 if N <= 512 else 1
    qkv = torch.from_numpy(rs.standard_normal((B, N + M, 3, 4, 32)) * 1.3)
    out, masks = ops.attention_f64(qkv.to(DEV), N, M, cross, topk=k, return_selection=True)
    out = out.cpu()
    for side, ((lo, hi), q, kk, v) in enumerate(_sides(qkv, N, M, cross)):
        rep = []
        ref, _ = O.dynamic_attention(q, kk, v, k, report=rep)
        own = rep[0]['own']                                              # [B, H, n, m]
        assert torch.equal(masks[side].cpu(), own), int((masks[side].cpu() ^ own).any(-1).sum())
        assert (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max() < F64_TOL


def test_attention_f64_topk_fp32_ties_are_ranked_in_fp64():
    """Logits whose fp32 roundings coincide at the k-th place: the kernel's search runs on the roundings, the tied candidates are
    then ranked by their fp64 values (f64.hip: the list path) - or, when more than 32 tie, by key order (exact duplicates)."""
    rs = np.random.RandomState(5)
    N = M = 128
    k = 16
    qkv = torch.from_numpy(rs.standard_normal((1, N + M, 3, 4, 32)) * 1.3)
    # per frame and head: make the (k+1)-th key of query row r a copy of the k-th key, moved by ~1e-10 relative - below the
    # fp32 resolution of the logit (6e-8 relative), far above fp64's
    planted = 0
    for lo in (0, N):
        for h in range(4):
            q = qkv[0, lo:lo + N, 0, h]
            kk = qkv[0, lo:lo + N, 1, h]
            logits = q @ kk.T / 32 ** 0.5
            used = set()                # a key is copied / overwritten at most once: two copies of one key would tie in fp64 too
            for r in range(0, N, 9):
                order = logits[r].argsort(descending=True)
                a, b = int(order[k - 1]), int(order[k])
                if a in used or b in used:
                    continue
                used.update((a, b))
                sign = 1.0 if (planted & 1) else -1.0
                kk[b] = kk[a] * (1.0 + sign * 1e-10)
                qkv[0, lo + b, 2, h] = torch.from_numpy(rs.standard_normal(32))          # a different value row: a wrong pick moves the message
                logits = q @ kk.T / 32 ** 0.5
                planted += 1
    assert planted > 12
    out, masks = ops.attention_f64(qkv.to(DEV), N, M, False, topk=k, return_selection=True)
    out = out.cpu()
    tied_rows = 0
    for side, ((lo, hi), q, kk, v) in enumerate(_sides(qkv, N, M, False)):
        rep = []
        ref, _ = O.dynamic_attention(q, kk, v, k, report=rep)
        logits = torch.einsum('bdhn,bdhm->bhnm', q, kk) / 32 ** 0.5
        top = logits.topk(k + 1, dim=3).values
        tied_rows += int((top[..., k - 1].float() == top[..., k].float()).sum())
        assert torch.equal(masks[side].cpu(), rep[0]['own'])
        assert (out[:, lo:hi] - _ref_msg_to_lib(ref)).abs().max() < F64_TOL
    assert tied_rows >= 8, tied_rows          # the construction did produce fp32 ties at the k-th place
    # masses of EXACT duplicates (40 copies of one key straddling the k-th place): exactly k keys, lowest indices among the ties
    qkv2 = torch.from_numpy(rs.standard_normal((1, 2 * 96, 3, 4, 32)))
    qkv2[0, 10:50, 1] = qkv2[0, 10, 1]
    qkv2[0, 10:50, 2] = qkv2[0, 10, 2]
    out2, masks2 = ops.attention_f64(qkv2.to(DEV), 96, 96, False, topk=24, return_selection=True)
    assert bool((masks2[0].sum(-1) == 24).all()) and bool((masks2[1].sum(-1) == 24).all())
    q, kk, v = (qkv2[:, :96, i].permute(0, 3, 2, 1) for i in range(3))
    ref, _ = O.dynamic_attention(q, kk, v, 24)
    assert (out2.cpu()[:, :96] - _ref_msg_to_lib(ref)).abs().max() < 1e-9       # (equal logits, equal values: any choice among them)
    m = masks2[0].cpu()[0]                                                      # [H, n, keys]
    dup = m[:, :, 10:50]
    first_gap = (~dup).float().argmax(-1)                                       # kept duplicates are a prefix of the group
    assert bool(((dup.sum(-1) == first_gap) | dup.all(-1)).all())


def _build(g, **cfg_over):
    B, n, m, L, S, seed, first_pair = [int(x) for x in g['meta']]
    k = [None if x < 0 else int(x) for x in g['k']]
    bin_score = float(g['bin_score']) if 'bin_score' in g else 1.0
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, **cfg_over)
    sd = synth.make_state_dict(L=L, seed=seed, bin_score=bin_score)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to(DEV)
    data = synth.make_batch(B, n, m, first_pair=first_pair)
    return net, cfg, sd, data, (B, n, m, L)


@pytest.mark.parametrize('B,N,M,iters,alpha,scale', [(1, 7, 5, 20, 1.0, 3.0), (2, 64, 64, 100, 0.37, 3.0), (2, 48, 64, 50, 1.0, 3.0), (3, 512, 512, 100, 1.0, 3.0),
                                                     (2, 400, 512, 100, 1.0, 6.0), (1, 100, 31, 10, 2.0, 1.0), (9, 256, 256, 20, 1.0, 3.0), (1, 1, 1, 5, 1.0, 1.0),
                                                     (2, 575, 33, 30, 1.0, 2.0), (1, 33, 575, 30, 0.5, 10.0),
                                                     # beyond 575 keypoints: the streaming form (K in memory, one launch per iteration)
                                                     (2, 600, 700, 100, 1.0, 3.0), (1, 577, 30, 50, 1.0, 3.0), (2, 30, 1200, 50, 1.0, 3.0),
                                                     (1, 2048, 2048, 200, 1.0, 3.0), (1, 1500, 2000, 100, 0.5, 6.0), (2, 2175, 2175, 3, 1.0, 3.0)])
def test_sinkhorn_f64(B, N, M, iters, alpha, scale):
    """csrc/sinkhorn_f64.hip (log_optimal_transport in the reference's own arithmetic, mdgat.py:279-308 run in float64) against the
    oracle: Z to 1e-12, and the match extraction of all four branches - every arg-max decided on the fp64 Z inside the kernel - equal to
    the extraction of the oracle's Z."""
    _sinkhorn_f64_case(B, N, M, iters, alpha, scale)


@pytest.mark.parametrize('B,N,M,iters,alpha,scale', [(1, 7, 5, 20, 1.0, 3.0), (2, 64, 64, 100, 0.37, 3.0), (2, 48, 64, 50, 1.0, 3.0), (3, 512, 512, 100, 1.0, 3.0),
                                                     (2, 400, 512, 100, 1.0, 6.0), (2, 65, 129, 0, 1.0, 3.0), (1, 1, 1, 5, 1.0, 1.0), (1, 33, 575, 30, 0.5, 10.0)])
def test_sinkhorn_f64_streaming_form_on_small_frames(B, N, M, iters, alpha, scale):
    """mdgat_set_f64_sinkhorn_form(1): the streaming form on frames the register-resident kernel holds - against the oracle like the
    default, and against the resident kernel: equal to rounding (other summation orders), the extraction identical."""
    lib = _lib.load()
    rs = np.random.RandomState(N + 3 * M + iters)
    s = torch.from_numpy(rs.standard_normal((B, N, M)) * scale).to(DEV)
    res = ops.sinkhorn_f64(s, alpha, iters)
    res_m = ops.sinkhorn_f64_extract(s, alpha, iters)
    prev = lib.mdgat_set_f64_sinkhorn_form(1)
    try:
        _sinkhorn_f64_case(B, N, M, iters, alpha, scale)
        wide = ops.sinkhorn_f64(s, alpha, iters)
        wide_m = ops.sinkhorn_f64_extract(s, alpha, iters)
    finally:
        lib.mdgat_set_f64_sinkhorn_form(prev)
    assert (wide - res).abs().max().item() < 1e-12
    assert all(torch.equal(a, b) for a, b in zip(wide_m[:2], res_m[:2])) and all((a - b).abs().max().item() < 1e-6 for a, b in zip(wide_m[2:], res_m[2:]))


def test_sinkhorn_f64_refuses_frames_beyond_its_kernels():
    with pytest.raises(RuntimeError, match='supported'):
        ops.sinkhorn_f64(torch.zeros(1, 8, 2176, dtype=torch.float64, device=DEV), 1.0, 2)
    with pytest.raises(RuntimeError, match='supported'):
        ops.sinkhorn_f64(torch.zeros(1, 2176, 8, dtype=torch.float64, device=DEV), 1.0, 2)


def _sinkhorn_f64_case(B, N, M, iters, alpha, scale):
    rs = np.random.RandomState(N + 3 * M + iters)
    s = torch.from_numpy(rs.standard_normal((B, N, M)) * scale)
    Zr = O.log_optimal_transport(s, torch.tensor(alpha, dtype=torch.float64), iters)
    Z = ops.sinkhorn_f64(s.to(DEV), alpha, iters).cpu()
    assert (Z - Zr).abs().max().item() < 1e-12
    for mode in range(4):
        m0, m1, s0, s1, Z32 = ops.sinkhorn_f64_extract(s.to(DEV), alpha, iters, mode=mode, match_threshold=0.01, want_Z=True)
        r0, r1, rs0, rs1 = ops.extract(Zr.float().to(DEV), mode=mode, match_threshold=0.01)
        assert torch.equal(m0, r0) and torch.equal(m1, r1), mode
        assert (s0 - rs0).abs().max().item() < 1e-6 and (s1 - rs1).abs().max().item() < 1e-6
        assert (Z32.cpu().double() - Zr).abs().max().item() < 4e-6 * max(1.0, Zr.abs().max().item() / 32)      # the fp32 rounding of the fp64 Z


def test_sinkhorn_f64_launches_on_concurrent_streams():
    """Launches of the fp64 Sinkhorn in flight on four streams, some of more workgroups than the device has CUs: the workgroups of a
    pair wait for each other while holding their CUs, so unadmitted concurrent launches can starve each other (three partly resident
    pairs fill an XCD; measured: every spin ran into its bound).  The launcher admits one launch at a time per device; every output
    must be bit-equal to the serial run's."""
    torch.manual_seed(0)
    cases = [(torch.randn(B, N, M, dtype=torch.float64, device=DEV) * 3, it) for (B, N, M, it) in
             [(3, 512, 512, 40), (9, 256, 300, 20), (1, 575, 575, 30), (20, 512, 512, 15), (40, 400, 400, 10), (2, 33, 17, 50)]]
    ref = [ops.sinkhorn_f64_extract(s, 1.0, it, want_Z=True) for s, it in cases]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV) for _ in range(4)]
    for rnd in range(3):
        outs = []
        for i, (s, it) in enumerate(cases * 2):
            with torch.cuda.stream(streams[(i + rnd) % 4]):
                outs.append((i % len(cases), ops.sinkhorn_f64_extract(s, 1.0, it, want_Z=True)))
        torch.cuda.synchronize()
        for ci, o in outs:
            for a, b in zip(o, ref[ci]):
                assert torch.equal(a, b), (rnd, ci)


def test_exact_mode_small_forwards_on_concurrent_streams():
    """One pair per call in the exact mode from four streams at once, two modules (two handles) on the device: every layer tail is a
    clustered launch (four workgroups per 16-row block that wait for each other), every Sinkhorn a launch of waiting row slabs - all
    admitted one at a time per device, a forward's launches as one group (csrc/coop_chain.hpp).  Every output bit-equal to the serial
    run's; no spin runs into its bound (check() would raise)."""
    L = 3
    nets = []
    for seed in (5, 6):
        net = MDGAT(synth.default_config(L=L, k=[64, None, 32, None], sinkhorn_iterations=15)).double()
        net.load_state_dict(synth.make_state_dict(L=L, seed=seed))
        nets.append(net.eval().to(DEV))
    keys = ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1')
    cases = [synth.make_batch(B, n, m, first_pair=fp, device=DEV, dtype=torch.float64) for (B, n, m, fp) in
             [(1, 512, 512, 0), (1, 256, 300, 3), (2, 200, 180, 5), (1, 512, 400, 9), (3, 100, 120, 11)]]
    with torch.no_grad():
        ref = [[[t.clone() for t in net._run(*(d[k] for k in keys), want_Z=True)] for d in cases] for net in nets]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(DEV) for _ in range(4)]
        for rnd in range(3):
            outs = []
            for i in range(4 * len(cases)):
                ni, ci = i % 2, (i // 2 + rnd) % len(cases)
                with torch.cuda.stream(streams[(i + rnd) % 4]):
                    outs.append((ni, ci, [t.clone() for t in nets[ni]._run(*(cases[ci][k] for k in keys), want_Z=True)]))
            torch.cuda.synchronize()
            for ni, ci, o in outs:
                for a, b in zip(o, ref[ni][ci]):
                    assert torch.equal(a, b), (rnd, ni, ci)
    for net in nets:
        net.check(DEV)


def test_sinkhorn_f64_decides_near_ties_like_fp64():
    """Two rows that are the same but for 1e-9 in one column: their potentials agree, so the two candidates of that column are 8e-10
    apart in the fp64 Z - equal as fp32 numbers.  The kernel's arg-max (superglue branch: over the inner block, mdgat.py:444) follows
    the larger one, whichever row holds it; an exact tie goes to the first index, as torch.max."""
    rs = np.random.RandomState(5)
    N = M = 96
    s = torch.from_numpy(rs.standard_normal((2, N, M)) * 2.0)
    s[:, 20, 7] = 6.0
    s[:, 60, :] = s[:, 20, :]
    s[0, 60, 7] += 1e-9                                     # pair 0: row 60 wins column 7
    s[1, 20, 7] += 1e-9                                     # pair 1: row 20
    Zr = O.log_optimal_transport(s, torch.tensor(1.0, dtype=torch.float64), 50)
    ref1 = Zr[:, :-1, :-1].max(1).indices
    assert int(ref1[0, 7]) == 60 and int(ref1[1, 7]) == 20 and 0 < float(Zr[0, 60, 7] - Zr[0, 20, 7]) < 2e-9
    assert float(Zr.float()[0, 60, 7]) == float(Zr.float()[0, 20, 7])             # (what an fp32 Z can see)
    m0, m1, s0, s1 = ops.sinkhorn_f64_extract(s.to(DEV), 1.0, 50, mode=2, match_threshold=0.01)
    assert int(m1[0, 7]) == 60 and int(m1[1, 7]) == 20
    assert torch.equal(m1.cpu(), ref1)                       # (every column clears the threshold here)
    s[1, 20, 7] = s[1, 60, 7]                               # an exact tie: the lower row
    m0, m1, s0, s1 = ops.sinkhorn_f64_extract(s.to(DEV), 1.0, 50, mode=2, match_threshold=0.01)
    assert int(m1[1, 7]) == 20


@pytest.mark.parametrize('name', ['fwd_n64_L1_S1', 'fwd_n64_L4_S20', 'fwd_n64_L5_S20', 'fwd_n48m64_L4_S20'])
def test_f64_stages_vs_reference(golden_dir, name):
    """Every stage tensor of the fp64 layers against the reference's (forward hooks, tools/make_goldens.py): fp32 taps of fp64
    values, so 1e-6 here; Z, matches and scores at the bar."""
    g = _g(golden_dir, name)
    net, cfg, sd, data, (B, n, m, L) = _build(g, arithmetic='fp64', f64_layers=2 * int(g['meta'][3]))
    P = n + m
    dev = {k: v.to(DEV) for k, v in data.items()}
    taps = {'x_enc': torch.empty(B, P, 128, device=DEV), 'x_layers': torch.empty(2 * L, B, P, 128, device=DEV),
            'mdesc': torch.empty(B, P, 128, device=DEV), 'scores': torch.empty(B, n, m, device=DEV)}
    m0, m1, s0, s1, Z = net._run(dev['keypoints0'], dev['scores0'], dev['descriptors0'], dev['keypoints1'], dev['scores1'],
                                 dev['descriptors1'], want_Z=True, taps=taps)
    torch.cuda.synchronize()
    net.check(DEV)
    t = lambda x: np.transpose(x, (0, 2, 1))
    enc = taps['x_enc'].cpu().double().numpy()
    assert max(np.abs(enc[:, :n] - t(g['enc0'])).max(), np.abs(enc[:, n:] - t(g['enc1'])).max()) < 2e-6
    xl = taps['x_layers'].cpu().double().numpy()
    for i in range(2 * L):
        e = max(np.abs(xl[i][:, :n] - t(g[f'layer{i}_desc0'])).max(), np.abs(xl[i][:, n:] - t(g[f'layer{i}_desc1'])).max())
        assert e < 5e-6, (i, e)
    assert np.abs(Z.cpu().double().numpy() - g['Z']).max() < 2e-5
    np.testing.assert_array_equal(m0.cpu().numpy(), g['default_matches0'])
    np.testing.assert_array_equal(m1.cpu().numpy(), g['default_matches1'])
    assert np.abs(s0.cpu().double().numpy() - g['default_mscores0']).max() < 2e-5


@pytest.mark.parametrize('name', ['cfg_n256_L4_S20', 'cfg_n512_L9_S100', 'cfg_n512_L9_S100_seed7', 'cfg_n2048_L9_S200', 'cfg_n2048_L9_S200_b'])
def test_literal_bar_on_every_reference_held_pair(golden_dir, name):
    """THE bar, literally: against the reference's own fp64 outputs, every pair - matches bit-identical, |dZ| < 1e-4 on every held
    entry, matching scores < 1e-4 - and not one top-k row selected differently from the fp64 oracle on the same trajectory."""
    from parity_util import hip_forward_with_selection
    g = _g(golden_dir, name)
    net, cfg, sd, data, (B, n, m, L) = _build(g)            # no 'arithmetic' key: the float64 module is the request (test.py:193)
    assert net.exact()
    dev = {k: v.to(DEV) for k, v in data.items()}
    (m0, m1, s0, s1, Z), forced = hip_forward_with_selection(net, dev)
    net.check(DEV)
    Zc = Z.cpu().double().numpy()
    sub = int(g['sub']) if 'sub' in g else 8
    mine = np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    err = np.abs(mine - ref_Z).reshape(B, -1).max(1)
    es = max(np.abs(s0.cpu().double().numpy() - g['default_mscores0']).max(), np.abs(s1.cpu().double().numpy() - g['default_mscores1']).max())
    print(f'[parity-f64] {name}: per-pair max|dZ| vs the reference {np.array2string(err, precision=2)}; mscores {es:.2e}; '
          f'pairs within the literal 1e-4: {int((err < Z_TOL).sum())}/{B}')
    np.testing.assert_array_equal(m0.cpu().numpy(), g['default_matches0'])
    np.testing.assert_array_equal(m1.cpu().numpy(), g['default_matches1'])
    assert (err < Z_TOL).all(), err
    assert es < Z_TOL
    assert np.abs(torch.logsumexp(Z.double(), 1).cpu().numpy() - g['Z_col_lse']).max() < Z_TOL
    # the selections are the fp64 oracle's own on the same trajectory: zero rows differ (every shape: at 2048 keypoints the oracle
    # takes a minute on the box's 16 cores)
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data, cap, forced_topk=forced)
    rows = sum(r['rows'] for reps in cap.get('topk_report', {}).values() for r in reps)
    bad = sum(r['bad_count'] for reps in cap.get('topk_report', {}).values() for r in reps)
    print(f'[parity-f64] {name}: top-k rows differing from the fp64 selection: {rows}; full Z max|d| {np.abs(Zc - cap["Z"].numpy()).max():.2e}')
    assert rows == 0 and bad == 0
    # the tail is fp64 at every size (beyond 575 keypoints the streaming fp64 Sinkhorn): Z is the fp32 rounding of an fp64 Z
    assert np.abs(Zc - cap['Z'].numpy()).max() < 1.2e-7 * max(1.0, np.abs(cap['Z'].numpy()).max())
    assert torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1'])
    # the untapped kernels (what ships) give the same bits
    plain = net._run(dev['keypoints0'], dev['scores0'], dev['descriptors0'], dev['keypoints1'], dev['scores1'], dev['descriptors1'], want_Z=True)
    assert torch.equal(plain[0], m0) and torch.equal(plain[1], m1) and torch.equal(plain[4], Z)


@pytest.mark.parametrize('tail', ['auto', 'fp32'])
def test_literal_bar_on_a_reference_held_batch_that_runs_in_slices(golden_dir, tail):
    """tests/golden/cfg_n512_L9_S100_b40.npz: 40 pairs of the headline shape run through the REFERENCE as one batch.  Here they are
    more than 32 768 keypoints: the library cuts them into slices of 32 + 8 pairs on two lanes, and the 32-pair slice gets the
    kernels big launches get (full attention with one wave per 32 queries, 32-keypoint layer tiles).  Through the dict API of a
    float64 module, against the reference's own outputs, on all 40 pairs: matching scores and every held entry of Z within the
    literal 1e-4, and
      * by default (sinkhorn_arithmetic 'auto': the tail - every layer, final_proj, scores, Sinkhorn, the extraction's arg-maxes - in
        fp64 too, csrc/sinkhorn_f64.hip): ALL 40 960 matches bit-identical, Z to the rounding of its fp32 output (1e-6);
      * with the fp32-class tail of rounds 5 / 6 ('fp32': Z good to 6e-6): identical except arg-max NEAR TIES below that accuracy - the
        batch holds exactly one: pair 18, column 71, whose two best rows the reference's fp64 Z separates by 1.3e-6.  Every mismatch
        is checked to be such a tie by this path's own Z (2e-5)."""
    from parity_util import near_tie_mismatches
    g = _g(golden_dir, 'cfg_n512_L9_S100_b40')
    net, cfg, sd, data, (B, n, m, L) = _build(g, **({} if tail == 'auto' else {'sinkhorn_arithmetic': tail}))
    assert net.exact() and 'arithmetic' not in cfg and B == 40
    dev = {k: v.to(DEV) for k, v in data.items()}
    with torch.no_grad():
        out = net(dev)
        Z = net.match(dev['keypoints0'], dev['descriptors0'], dev['keypoints1'], dev['descriptors1'], dev['scores0'], dev['scores1'],
                      return_scores=True)[4]
    torch.cuda.synchronize()
    net.check(DEV)
    ties = near_tie_mismatches(Z, out['matches0'], out['matches1'], g['default_matches0'], g['default_matches1'], 2e-5)
    print(f'[parity-f64] cfg_n512_L9_S100_b40, tail {tail}: arg-max near ties decided the other way: {ties}')
    assert len(ties) == (0 if tail == 'auto' else 1), ties
    es = max(np.abs(out['matching_scores0'].cpu().numpy() - g['default_mscores0']).max(),
             np.abs(out['matching_scores1'].cpu().numpy() - g['default_mscores1']).max())
    Zc = Z.cpu().double().numpy()
    sub = int(g['sub'])
    mine = np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    err = np.abs(mine - ref_Z).reshape(B, -1).max(1)
    print(f'[parity-f64] cfg_n512_L9_S100_b40, tail {tail}: worst pair max|dZ| vs the reference {err.max():.2e}, mscores {es:.2e}; '
          f'pairs within the literal 1e-4: {int((err < Z_TOL).sum())}/{B}')
    assert (err < (2e-6 if tail == 'auto' else Z_TOL)).all(), err
    assert es < (1e-6 if tail == 'auto' else Z_TOL)
    assert np.abs(torch.logsumexp(Z.double(), 1).cpu().numpy() - g['Z_col_lse']).max() < Z_TOL


@pytest.mark.parametrize('name,pairs', [('cfg_n400_L9_S100_b24', 24), ('cfg_n256_L4_S20_b64', 64)])
def test_literal_bar_on_a_second_reference_held_batch(golden_dir, name, pairs):
    """tests/golden/cfg_n400_L9_S100_b24.npz: 24 pairs of 400 keypoints (no multiple of the 32-row slabs of the fp64 Sinkhorn, of the
    64-key register blocks, of the 128-query workgroups) the REFERENCE ran as one batch, with other weights (seed 7) and bin score
    0.37; cfg_n256_L4_S20_b64.npz: 64 pairs of BASELINE configs[0]'s shape (L = 4: every self layer dynamic; weights seed 3).  A float64
    module, no extra key: every match identical, Z to the rounding of its fp32 output, matching scores to 1e-6."""
    g = _g(golden_dir, name)
    net, cfg, sd, data, (B, n, m, L) = _build(g)
    assert net.exact() and B == pairs
    dev = {k: v.to(DEV) for k, v in data.items()}
    with torch.no_grad():
        out = net(dev)
        Z = net.match(dev['keypoints0'], dev['descriptors0'], dev['keypoints1'], dev['descriptors1'], dev['scores0'], dev['scores1'],
                      return_scores=True)[4]
    torch.cuda.synchronize()
    net.check(DEV)
    np.testing.assert_array_equal(out['matches0'].cpu().numpy(), g['default_matches0'])
    np.testing.assert_array_equal(out['matches1'].cpu().numpy(), g['default_matches1'])
    es = max(np.abs(out['matching_scores0'].cpu().numpy() - g['default_mscores0']).max(),
             np.abs(out['matching_scores1'].cpu().numpy() - g['default_mscores1']).max())
    Zc = Z.cpu().double().numpy()
    sub = int(g['sub'])
    mine = np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    err = np.abs(mine - ref_Z).max()
    print(f'[parity-f64] {name}: max|dZ| vs the reference {err:.2e}, mscores {es:.2e}, every match identical')
    assert err < 2e-6 and es < 1e-6


@pytest.mark.parametrize('B', [3, 40])
def test_f64_tail_applies_the_batch_wide_dustbin_rule(B):
    """mdgat.py:465-467 through the fp64 tail (the extraction runs from the arg-maxes the fp64 Sinkhorn decided): a bin score of 60 sends
    every keypoint to the dustbin - all matches -1 and, the reference's quirk, INTEGER zero scores - for one launch and for a batch that
    runs in slices on two lanes (the rule is batch-wide: applied after the slices)."""
    cfg = synth.default_config(L=2, k=[64, None, 32, None], sinkhorn_iterations=10)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5, bin_score=60.0))
    net = net.eval().to(DEV)
    assert net.exact()
    out = net(synth.make_batch(B, 512, 512, device=DEV))
    assert (out['matches0'] == -1).all() and (out['matches1'] == -1).all()
    assert out['matching_scores0'].dtype == torch.int64 and (out['matching_scores0'] == 0).all() and (out['matching_scores1'] == 0).all()


@pytest.mark.parametrize('B,n', [(3, 300), (2, 600), (40, 512)])
def test_exact_mode_forward_is_capturable_as_a_hip_graph(B, n):
    """The exact mode under stream capture (torch.cuda.CUDAGraph = hipGraph): the register-resident fp64 Sinkhorn (300 keypoints; its
    launcher's one-launch-at-a-time chaining stands aside for a capturing stream) and the streaming form (600 keypoints: 2 S + 3
    launches, nothing waits inside a launch); 40 pairs of 512: two lanes, a resident Sinkhorn launch on each, side by side in the graph.
    The replay gives the bits of the eager call, also after the inputs changed in place."""
    cfg = synth.default_config(L=2, k=[64, None, 32, None], sinkhorn_iterations=12)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5))
    net = net.eval().to(DEV)
    assert net.exact()
    keys = ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1')
    d = synth.make_batch(B, n, n - 7, device=DEV, dtype=torch.float64)
    other = synth.make_batch(B, n, n - 7, first_pair=300, device=DEV, dtype=torch.float64)
    inputs = tuple(d[k] for k in keys)
    with torch.no_grad():
        eager = [t.clone() for t in net._run(*inputs, want_Z=True)]
        eager_other = [t.clone() for t in net._run(*(other[k] for k in keys), want_Z=True)]
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            net._run(*inputs, want_Z=True)
        torch.cuda.current_stream(DEV).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = net._run(*inputs, want_Z=True)
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(eager, out))
        for k, t in zip(keys, inputs):
            t.copy_(other[k])
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(eager_other, out))
        assert not all(torch.equal(a, b) for a, b in zip(eager, eager_other))
    net.check(DEV)


def test_sinkhorn_arithmetic_key():
    """config['sinkhorn_arithmetic']: 'auto' and 'fp64' are the fp64 tail (frames beyond 575 keypoints: the streaming form of the fp64
    Sinkhorn), 'fp32' the fp32-class one; the three agree on the matches of an ordinary pair, the first two bit for bit."""
    L = 1
    sd = synth.make_state_dict(L=L, seed=4)
    outs = {}
    for tail in ('auto', 'fp64', 'fp32'):
        net = MDGAT(synth.default_config(L=L, k=[16, None], sinkhorn_iterations=10, sinkhorn_arithmetic=tail)).double()
        net.load_state_dict(sd)
        net = net.eval().to(DEV)
        d = synth.make_batch(2, 80, 70, device=DEV)
        outs[tail] = net.match(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'], return_scores=True)
        net.check(DEV)
        big = synth.make_batch(1, 600, 600, device=DEV)
        args = (big['keypoints0'], big['descriptors0'], big['keypoints1'], big['descriptors1'], big['scores0'], big['scores1'])
        outs[tail + '_big'] = net.match(*args, return_scores=True)
        net.check(DEV)
    for sfx in ('', '_big'):
        assert torch.equal(outs['auto' + sfx][0], outs['fp64' + sfx][0]) and torch.equal(outs['auto' + sfx][4], outs['fp64' + sfx][4])
        assert (outs['auto' + sfx][4] - outs['fp32' + sfx][4]).abs().max().item() < 2e-5
    assert torch.equal(outs['auto'][0], outs['fp32'][0])
    with pytest.raises(ValueError):
        MDGAT(synth.default_config(L=1, sinkhorn_arithmetic='bf16'))


@pytest.mark.parametrize('name', ['cfg_n512_L9_S100', 'cfg_n256_L4_S20', 'cfg_n2048_L9_S200'])
def test_literal_bar_in_the_big_batch_attention_form(golden_dir, name):
    """The same bar with full attention forced into the form batches of 32 pairs and more get (one wave per 32 queries over all keys,
    mdgat_set_f64_attention_form(1)): the reference-held pairs are batches of 8 / 8 / 1, so by default they exercise the split-key form
    only.  Matches bit-identical to the reference's, Z within the literal 1e-4 on every held entry - and matches and Z (to the rounding
    of the fp32 hand-over) the same as in the default form."""
    from mdgat_matcher_amd import _lib
    lib = _lib.load()
    g = _g(golden_dir, name)
    net, cfg, sd, data, (B, n, m, L) = _build(g)
    assert net.exact()
    dev = {k: v.to(DEV) for k, v in data.items()}
    args = (dev['keypoints0'], dev['scores0'], dev['descriptors0'], dev['keypoints1'], dev['scores1'], dev['descriptors1'])
    base = net._run(*args, want_Z=True)
    assert lib.mdgat_set_f64_attention_form(1) == -1
    try:
        m0, m1, s0, s1, Z = net._run(*args, want_Z=True)
        torch.cuda.synchronize()
        net.check(DEV)
    finally:
        lib.mdgat_set_f64_attention_form(-2)
    Zc = Z.cpu().double().numpy()
    sub = int(g['sub']) if 'sub' in g else 8
    mine = np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    err = np.abs(mine - ref_Z).reshape(B, -1).max(1)
    print(f'[parity-f64] {name}, one wave per 32 queries: per-pair max|dZ| vs the reference {np.array2string(err, precision=2)}; '
          f'against the split-key form {(Z - base[4]).abs().max().item():.2e}')
    np.testing.assert_array_equal(m0.cpu().numpy(), g['default_matches0'])
    np.testing.assert_array_equal(m1.cpu().numpy(), g['default_matches1'])
    assert (err < Z_TOL).all(), err
    assert torch.equal(m0, base[0]) and torch.equal(m1, base[1]) and (Z - base[4]).abs().max().item() < 1e-5


@pytest.mark.parametrize('name', ['var_n256_L4_S20', 'var_n512_L9_S100', 'var_n400m512_L9_S100'])
def test_literal_bar_on_the_reference_held_variants(golden_dir, name):
    """The other reference-held pairs at config scale (tests/golden/var_*.npz): all four extraction branches of mdgat.py:441-483 and
    a ragged 400 x 512 pair, through the dict API of a float64 module with no extra config key.  Against the REFERENCE's own
    outputs: matches bit-identical and matching scores within the literal 1e-4 in every branch, Z (the same for every branch)
    within 1e-4 on every held entry, zero top-k rows selected differently from the fp64 oracle."""
    from parity_util import hip_forward_with_selection
    g = _g(golden_dir, name)
    checked = 0
    for tag, (loss_method, mutual) in {'default': ('triplet_loss', False), 'mutual': ('triplet_loss', True),
                                       'sg': ('superglue', False), 'sgmutual': ('superglue', True)}.items():
        if f'{tag}_matches0' not in g:
            continue
        net, cfg, sd, data, (B, n, m, L) = _build(g, loss_method=loss_method, mutual_check=mutual)
        assert net.exact() and 'arithmetic' not in cfg
        dev = {k: v.to(DEV) for k, v in data.items()}
        with torch.no_grad():
            out = net(dev)
        np.testing.assert_array_equal(out['matches0'].cpu().numpy(), g[f'{tag}_matches0'], err_msg=tag)
        np.testing.assert_array_equal(out['matches1'].cpu().numpy(), g[f'{tag}_matches1'], err_msg=tag)
        es = max(np.abs(out['matching_scores0'].cpu().numpy() - g[f'{tag}_mscores0']).max(),
                 np.abs(out['matching_scores1'].cpu().numpy() - g[f'{tag}_mscores1']).max())
        assert es < Z_TOL, (tag, es)
        checked += 1
        if tag != 'default':
            continue
        (m0, m1, s0, s1, Z), forced = hip_forward_with_selection(net, dev)
        net.check(DEV)
        Zc = Z.cpu().double().numpy()
        mine = np.concatenate([Zc[:, ::8, ::8].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
        ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
        err = np.abs(mine - ref_Z).max()
        cap = {}
        O.mdgat_forward(sd, cfg, data, cap, forced_topk=forced)
        rows = sum(r['rows'] for reps in cap.get('topk_report', {}).values() for r in reps)
        print(f'[parity-f64] {name}: max|dZ| vs the reference {err:.2e}, mscores {es:.2e}, top-k rows differing {rows}')
        assert err < Z_TOL and rows == 0
        assert np.abs(Zc - cap['Z'].numpy()).max() < Z_TOL
    assert checked >= 2


@pytest.mark.parametrize('B,n,m,L,k', [(1, 512, 512, 3, [128, None, 64, None]), (3, 100, 77, 2, [16, None, 8, None]), (20, 512, 512, 2, [64, None]),
                                       (2, 1000, 1030, 2, [64, 32, None, 16]), (5, 33, 47, 4, [None, 8, 4, None]), (3, 70, 64, 1, [])])
def test_f64_fused_layer_tail_equals_three_launches(B, n, m, L, k):
    """csrc/layer_f64.hip (mlp.0 -> ReLU -> mlp.3 -> + x -> next q | k | v in one launch, the hidden activation in LDS, the weights in
    fragment order straight from L2) against the three gemm_f64_kernel launches it replaces - and the fused encoder launch (both
    encoders, their sum, layer 0's q | k | v) against the seven it replaces: every accumulator walks k in the same order, so EVERY
    output bit is the same - matches, scores, Z, and the fp32 taps of the encoder output and of every layer's descriptors - for each
    tile height (16 / 32 / 64 keypoints per workgroup), ragged keypoint counts, one pair, a batch large enough to run in slices, and
    a network without a dynamic layer (encoders only in fp64: the encoder launch hands over)."""
    from mdgat_matcher_amd import _lib
    lib = _lib.load()
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=10)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=21))
    net = net.eval().to(DEV)
    assert net.exact()
    d = synth.make_batch(B, n, m, device=DEV, first_pair=40)
    P = n + m

    def run(mode, taps):
        prev = lib.mdgat_set_f64_layer_fusion(mode)
        try:
            t = {'x_layers': torch.zeros(2 * L, B, P, 128, device=DEV), 'x_enc': torch.zeros(B, P, 128, device=DEV)} if taps else None
            out = net._run(d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'], want_Z=True, taps=t)
            torch.cuda.synchronize()
            net.check(DEV)
            return out, t
        finally:
            lib.mdgat_set_f64_layer_fusion(-1)
            assert prev in (0, 1, 2, 16, 32, 64)
    for taps in (False, True):          # (taps run the whole batch unsliced)
        ref, rt = run(0, taps)
        for mode in (1, 2, 16, 32, 64):     # (1: launches of few 16-row blocks take the clustered kernel - four workgroups per block; 2: never)
            out, ot = run(mode, taps)
            for a, b, what in zip(ref, out, ('matches0', 'matches1', 'mscores0', 'mscores1', 'Z')):
                assert torch.equal(a, b), (mode, taps, what)
            if taps:
                assert torch.equal(rt['x_layers'], ot['x_layers']) and torch.equal(rt['x_enc'], ot['x_enc']), mode


def test_f64_dict_api_slices_and_errors():
    """forward(dict) in the exact mode: a batch large enough to run in slices on two lanes equals the pairs run alone; fp32
    entry points refuse an fp64 handle."""
    L = 2
    cfg = synth.default_config(L=L, k=[16, None, 8, None], sinkhorn_iterations=10, arithmetic='fp64')
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=3))
    net = net.eval().to(DEV)
    os.environ.pop('MDGAT_FORWARD_SLICE_POINTS', None)
    data = synth.make_batch(6, 96, 80, device=DEV)
    out = net(data)
    assert out['matches0'].dtype == torch.int64 and out['matching_scores0'].dtype == torch.float64
    args = (data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'], data['scores0'], data['scores1'])
    m0, m1, s0, s1, Z = net.match(*args, return_scores=True)
    assert torch.equal(m0, out['matches0'])
    one = net.match(*[a[4:5] for a in args], return_scores=True)
    assert torch.equal(one[0][0], m0[4]) and torch.equal(one[4][0], Z[4])
    sd = synth.make_state_dict(L=L, seed=3)
    cap = {}
    ref = O.mdgat_forward(sd, cfg, {k: v.cpu() for k, v in data.items()}, cap)
    assert torch.equal(m0.cpu(), ref['matches0']) and (Z.cpu().double() - cap['Z']).abs().max() < 2e-5
    # records in (mdgat_forward_frames on the fp64 handle): the same path
    rec0 = torch.cat([data['keypoints0'], data['scores0'][..., None], data['descriptors0']], -1).float()
    rec1 = torch.cat([data['keypoints1'], data['scores1'][..., None], data['descriptors1']], -1).float()
    mf = net.match_frames(rec0, rec1, normalize=False)
    assert (mf[0] == m0).float().mean() > 0.95


def test_f64_records_in_equal_the_reference_loaders_arrays(golden_dir):
    """Exact mode from the raw keypoint records (mdgat_forward_frames on an fp64 handle): the library takes the float32 records
    through the loader's own sequence (load_data.py:146-165, 290-295: FPFH normalised in float32 as numpy does it, then widened),
    so matching the records equals matching the arrays the REFERENCE loader made of them (tests/golden/aux_loader.npz) - not
    approximately: every output bit for bit.  (The fp32-class path normalises with its own summation order: 1e-5.)"""
    g = np.load(os.path.join(golden_dir, 'aux_loader.npz'))
    L = 2
    cfg = synth.default_config(L=L, k=[32, None, 16, None], sinkhorn_iterations=20, arithmetic='fp64')
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=4))
    net = net.eval().to(DEV)
    for j in range(int(g['n_items'])):
        r0 = torch.from_numpy(g[f'item{j}_rec0']).to(DEV)
        r1 = torch.from_numpy(g[f'item{j}_rec1']).to(DEV)
        a = net.match_frames(r0, r1, return_scores=True)
        t = {k: torch.from_numpy(g[f'item{j}_{k}']).to(DEV) for k in ('keypoints0', 'keypoints1', 'descriptors0', 'descriptors1',
                                                                         'scores0', 'scores1')}
        b = net.match(t['keypoints0'], t['descriptors0'], t['keypoints1'], t['descriptors1'], t['scores0'], t['scores1'],
                      return_scores=True)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # a batch of records in slices on two lanes, normalisation off: the arrays widened on the host
    data = synth.make_batch(5, 200, 168, device=DEV)
    rec0 = torch.cat([data['keypoints0'], data['scores0'][..., None], data['descriptors0']], -1).float()
    rec1 = torch.cat([data['keypoints1'], data['scores1'][..., None], data['descriptors1']], -1).float()
    os.environ['MDGAT_FORWARD_SLICE_POINTS'] = '900'
    try:
        a = net.match_frames(rec0, rec1, normalize=False, return_scores=True)
        b = net.match(rec0[..., :3].double(), rec0[..., 4:].double(), rec1[..., :3].double(), rec1[..., 4:].double(),
                      rec0[..., 3].double(), rec1[..., 3].double(), return_scores=True)
    finally:
        os.environ.pop('MDGAT_FORWARD_SLICE_POINTS', None)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_f64_refuses_non_finite_inputs():
    """The reference lets a NaN / inf input run through to NaN outputs.  In the exact mode a ReLU would turn the NaN into 0 and the
    call would return plausible numbers: the inputs are tested by their bits (csrc/f64.hip is compiled with -fno-honor-nans) and the
    call is refused like any out-of-range activation - arrays and raw records, and a record whose FPFH row is all zero
    (1 / 0 in the loader's normalisation: NaN descriptors in the reference)."""
    L = 2
    cfg = synth.default_config(L=L, k=[16, None, 8, None], sinkhorn_iterations=10, arithmetic='fp64')
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=3))
    net = net.eval().to(DEV)
    clean = synth.make_batch(2, 96, 80, device=DEV)
    net(clean)                                            # a clean call passes
    for key, idx in (('descriptors0', (1, 5, 7)), ('keypoints1', (0, 3, 1)), ('scores0', (1, 9))):
        for bad in (float('nan'), float('inf'), -float('inf')):
            data = {k: v.clone() for k, v in clean.items()}
            data[key][idx] = bad
            with pytest.raises(RuntimeError):
                net(data)
    rec0 = torch.cat([clean['keypoints0'], clean['scores0'][..., None], clean['descriptors0']], -1).float()
    rec1 = torch.cat([clean['keypoints1'], clean['scores1'][..., None], clean['descriptors1']], -1).float()
    net.match_frames(rec0, rec1)
    net.check(DEV)
    for poke in ('nan', 'zero_row'):
        r0 = rec0.clone()
        if poke == 'nan':
            r0[1, 4, 20] = float('nan')
        else:
            r0[0, 7, 4:] = 0.0
        net.match_frames(r0, rec1)
        with pytest.raises(RuntimeError):
            net.check(DEV)
    net(clean)                                            # and the module keeps working afterwards


@pytest.mark.parametrize('plant', ['residual_overflow', 'qk_beyond_range', 'hidden_nan'])
def test_f64_refuses_non_finite_values_mid_stack(plant):
    """Non-finite values that arise INSIDE the fp64 layers - not at the inputs, not at the hand-over to the fp32 kernels.  The
    reference propagates them to NaN outputs (mdgat.py:192-193); csrc/f64.hip is compiled with -fno-honor-nans and its ReLU /
    row maximum / clamped exponential would swallow a NaN, so every fp64 product tests its outputs by their exponent bits in its
    epilogue (f64_out_of_range: not finite or beyond 2^500, before the ReLU and after the residual) and the call is refused.
    The plants go into the fp64 blob only (load_packed: the fp32 blob's own range check at load time would refuse them first):
      residual_overflow  mlp.3 of layer 0 scaled by 1e300: the residual stream overflows to +-inf in layer 0's tail;
      qk_beyond_range    the q and k rows of layer 1 scaled by 1e160: q.k would overflow and the online softmax would meet
                         inf - inf - refused where q and k are produced, so no logit can be anything but finite;
      hidden_nan         one mlp.0 weight of layer 1 is NaN: the hidden layer's ReLU would turn the NaN into 0."""
    from mdgat_matcher_amd import pack
    L = 2
    cfg = synth.default_config(L=L, k=[16, None, 8, None], sinkhorn_iterations=10)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=3))
    net = net.eval().to(DEV)
    assert net.exact()
    data = synth.make_batch(2, 96, 80, device=DEV)
    good = net(data)
    lay = pack.blob_layout(L)
    blob = net.packed_weights()
    blob64 = net.packed_weights(np.float64).copy()

    def sl(layer, name, n):
        o = lay['layer0'] + layer * lay['layer_stride'] + lay[name]
        return slice(o, o + n)
    if plant == 'residual_overflow':
        blob64[sl(0, 'mlp2_w', 128 * 256)] *= 1e300
    elif plant == 'qk_beyond_range':
        blob64[sl(1, 'qkv_w', 256 * 128)] *= 1e160
        blob64[sl(1, 'qkv_b', 256)] *= 1e160
    else:
        blob64[sl(1, 'mlp1_w', 256 * 256)][1234] = float('nan')
    net.load_packed(torch.from_numpy(blob).to(DEV), torch.from_numpy(blob64).to(DEV))
    with pytest.raises(RuntimeError, match='range|finite'):
        net(data)
    # the clean blobs again: the module works and gives the same bits as before
    net.load_packed(torch.from_numpy(blob).to(DEV), torch.from_numpy(net.packed_weights(np.float64)).to(DEV))
    again = net(data)
    assert torch.equal(again['matches0'], good['matches0']) and torch.equal(again['matching_scores0'], good['matching_scores0'])


def test_f64_large_batch_runs_in_slices_on_two_lanes():
    """40 pairs of 512 keypoints are more than 32 768 keypoints: the library cuts the batch into two slices on two lanes (csrc/api.hip:
    forward_batched) - in the exact mode too, each lane with its own fp64 workspace.  The lanes setting changes no bit.  What a pair
    returns does not depend on the batch it travels in: with full attention in the split-key form at every launch size
    (mdgat_set_f64_attention_form(0)) bit for bit; by default the 32-pair slice runs the one-wave-per-query-block form, which sums a
    row in another order than the form a single pair gets - the same matches, Z equal to the rounding of the fp32 hand-over."""
    from mdgat_matcher_amd import _lib
    lib = _lib.load()
    L = 2
    cfg = synth.default_config(L=L, k=[64, None, 32, None], sinkhorn_iterations=20, arithmetic='fp64')
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=9))
    net = net.eval().to(DEV)
    data = synth.make_batch(40, 512, 512, device=DEV, first_pair=500)
    args = (data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'], data['scores0'], data['scores1'])
    for form in (0, -1):
        prev = lib.mdgat_set_f64_attention_form(form)
        try:
            m0, m1, s0, s1, Z = net.match(*args, return_scores=True)
            torch.cuda.synchronize()
            net.check(DEV)
            for b in (0, 19, 20, 39):
                one = net.match(*[a[b:b + 1] for a in args], return_scores=True)
                assert torch.equal(one[0][0], m0[b]) and torch.equal(one[1][0], m1[b]), (form, b)
                if form == 0:
                    assert torch.equal(one[4][0], Z[b]), b
                else:
                    assert (one[4][0] - Z[b]).abs().max().item() < 1e-5, (b, (one[4][0] - Z[b]).abs().max().item())
            net.set_lanes(1)
            again = net.match(*args, return_scores=True)
            assert torch.equal(again[0], m0) and torch.equal(again[4], Z)
            net.set_lanes(2)
        finally:
            lib.mdgat_set_f64_attention_form(-2)
            assert prev == -1


def test_fuzz_forward_f64_short():
    """20 s of tools/fuzz_forward_f64.py: random shapes, depths, top-k schedules, extraction modes, bin scores through the exact
    mode against the UNFORCED oracle - literal 1e-4 on Z, no top-k row selected differently, matches consistent."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('fuzz_forward_f64', os.path.join(root, 'tools', 'fuzz_forward_f64.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    cases, fails, worst = fz.run(20.0, seed=3, verbose=True)
    assert cases >= 5 and fails == 0 and worst < Z_TOL


def test_fuzz_f64_topk_short():
    """15 s of tools/fuzz_f64_topk.py: the fp64 dynamic attention's kept keys = torch.topk on fp64 logits, random shapes, k, logit
    scales and duplicated keypoints (it found the radix select's one bug)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('fuzz_f64_topk', os.path.join(root, 'tools', 'fuzz_f64_topk.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    cases, rows, bad = fz.run(15.0, seed=7, verbose=True)
    assert cases >= 20 and rows > 10000 and not bad
