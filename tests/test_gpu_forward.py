"""End-to-end GPU parity of the drop-in MDGAT module (HIP path through the C ABI) against the golden
vectors captured from the real reference, plus size-independent properties at the bench shape."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mdgat_matcher_amd import MDGAT, synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402

from parity_util import assert_attributed, assert_plain, assert_plain_vs_oracle, attributed_parity  # noqa: E402

DEV = 'cuda:0'
Z_TOL = 1e-4        # north star: soft-assignment matrix within 1e-4 (fp32-class kernels vs fp64 reference)
# Dynamic (top-k) layers are discontinuous in their logits.  Configurations that have them are held to the bar in the
# two-statement form of tests/parity_util.py: Z within 1e-4 everywhere and identical matches against the fp64 oracle
# run with the HIP path's own top-k selections, and every selection that differs from the oracle's is a near-tie
# below GAP_EPS.  On top of that, ALWAYS (assert_plain): matches bit-identical to the unforced fp64 result (the
# reference's golden output where one is held), plain max|dZ| < 2e-3, at most 1e-3 of the entries beyond 1e-4, and the
# literal 1e-4 on every pair in which no selection flipped.


@pytest.fixture(autouse=True)
def _throughput_path(monkeypatch):
    """This file tests the fp32-class THROUGHPUT path.  A float64 module (net.double(), which most tests here call because the
    reference's callers do) runs the reference-exact mode by default - tests/test_gpu_f64.py and tests/test_gpu_dropin.py cover
    that - so the documented environment default pins the path under test for modules whose config carries no 'arithmetic'."""
    monkeypatch.setenv('MDGAT_ARITHMETIC', 'fp32')


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _build(g, **cfg_over):
    B, n, m, L, S, seed, first_pair = [int(x) for x in g['meta']]
    k = [None if x < 0 else int(x) for x in g['k']]
    bin_score = float(g['bin_score']) if 'bin_score' in g else 1.0
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, **cfg_over)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=seed, bin_score=bin_score))
    net = net.double().eval().to(DEV)
    data = synth.make_batch(B, n, m, first_pair=first_pair, device=DEV)
    return net, data, (B, n, m, L)


def _to_lib(x):
    """reference [B, 128, N] -> [B, N, 128]"""
    return np.transpose(x, (0, 2, 1))


@pytest.mark.parametrize('name', ['fwd_n64_L1_S1', 'fwd_n64_L4_S20', 'fwd_n64_L5_S20', 'fwd_n48m64_L4_S20'])
def test_forward_stages_golden(golden_dir, name):
    g = _g(golden_dir, name)
    net, data, (B, n, m, L) = _build(g)
    P = n + m
    taps = {
        'x_enc': torch.empty(B, P, 128, device=DEV),
        'x_layers': torch.empty(2 * L, B, P, 128, device=DEV),
        'mdesc': torch.empty(B, P, 128, device=DEV),
        'scores': torch.empty(B, n, m, device=DEV),
    }
    m0, m1, s0, s1, Z = net._run(data['keypoints0'], data['scores0'], data['descriptors0'],
                                 data['keypoints1'], data['scores1'], data['descriptors1'], want_Z=True, taps=taps)
    torch.cuda.synchronize()
    enc = taps['x_enc'].cpu().double().numpy()
    assert np.abs(enc[:, :n] - _to_lib(g['enc0'])).max() < 1e-5
    assert np.abs(enc[:, n:] - _to_lib(g['enc1'])).max() < 1e-5
    xl = taps['x_layers'].cpu().double().numpy()
    for i in range(2 * L):
        e0 = np.abs(xl[i][:, :n] - _to_lib(g[f'layer{i}_desc0'])).max()
        e1 = np.abs(xl[i][:, n:] - _to_lib(g[f'layer{i}_desc1'])).max()
        assert max(e0, e1) < 5e-5, (i, e0, e1)
    md = taps['mdesc'].cpu().double().numpy()
    assert np.abs(md[:, :n] - _to_lib(g['mdesc0'])).max() < 5e-5
    assert np.abs(taps['scores'].cpu().double().numpy() - g['scores']).max() < Z_TOL
    assert np.abs(Z.cpu().double().numpy() - g['Z']).max() < Z_TOL
    np.testing.assert_array_equal(m0.cpu().numpy(), g['default_matches0'])
    np.testing.assert_array_equal(m1.cpu().numpy(), g['default_matches1'])
    assert np.abs(s0.cpu().double().numpy() - g['default_mscores0']).max() < Z_TOL
    assert np.abs(s1.cpu().double().numpy() - g['default_mscores1']).max() < Z_TOL


@pytest.mark.parametrize('name', ['fwd_n64_L4_S20', 'fwd_n48m64_L4_S20'])
def test_forward_dict_contract_and_variants(golden_dir, name):
    g = _g(golden_dir, name)
    for tag, (loss_method, mutual) in {'default': ('triplet_loss', False), 'mutual': ('triplet_loss', True),
                                       'sg': ('superglue', False), 'sgmutual': ('superglue', True)}.items():
        if f'{tag}_matches0' not in g:
            continue
        net, data, _ = _build(g, loss_method=loss_method, mutual_check=mutual)
        with torch.no_grad():
            out = net(data)
        assert set(out) >= {'matches0', 'matches1', 'matching_scores0', 'matching_scores1', 'loss'}
        assert out['matches0'].dtype == torch.int64 and out['matching_scores0'].dtype == torch.float64
        np.testing.assert_array_equal(out['matches0'].cpu().numpy(), g[f'{tag}_matches0'], err_msg=tag)
        np.testing.assert_array_equal(out['matches1'].cpu().numpy(), g[f'{tag}_matches1'], err_msg=tag)
        assert np.abs(out['matching_scores0'].cpu().numpy() - g[f'{tag}_mscores0']).max() < Z_TOL, tag
        assert np.abs(out['matching_scores1'].cpu().numpy() - g[f'{tag}_mscores1']).max() < Z_TOL, tag


@pytest.mark.parametrize('name', ['cfg_n256_L4_S20', 'cfg_n512_L9_S100', 'cfg_n2048_L9_S200', 'cfg_n2048_L9_S200_b', 'cfg_n512_L9_S100_seed7'])
def test_config_shapes_golden(golden_dir, name):
    """BASELINE configs[0] / configs[1] shapes (8 pairs each) and configs[4] (N = 2048, L = 9, 200 iterations: 1 + 3 pairs) with the default dynamic schedule, against the REFERENCE's
    own fp64 output (tests/golden/cfg_*.npz) - unconditionally: matches bit-identical, plain |dZ| bounded, the literal
    1e-4 on every pair without a flipped selection - and against the oracle with the HIP selections forced."""
    g = _g(golden_dir, name)
    net, _, (B, n, m, L) = _build(g)
    seed, first_pair = int(g['meta'][5]), int(g['meta'][6])
    k = [None if x < 0 else int(x) for x in g['k']]
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=int(g['meta'][4]))
    sd = synth.make_state_dict(L=L, seed=seed, bin_score=float(g['bin_score']) if 'bin_score' in g else 1.0)
    res = attributed_parity(net, cfg, sd, synth.make_batch(B, n, m, first_pair=first_pair), DEV)
    assert_attributed(res, name)
    Z = res['out'][4]
    # column marginals are exact by construction and independent of the top-k selections
    assert np.abs(torch.logsumexp(Z.double(), 1).cpu().numpy() - g['Z_col_lse']).max() < Z_TOL
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    sub = int(g['sub']) if 'sub' in g else 8
    assert_plain(res, ref_Z, g['default_matches0'], g['default_matches1'], g['default_mscores0'], g['default_mscores1'],
                 name + ' (reference golden)',
                 z_index=lambda Zc: np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1))


def test_reference_held_batch_that_runs_in_slices(golden_dir):
    """tests/golden/cfg_n512_L9_S100_b40.npz: 40 pairs of the headline shape the REFERENCE ran as one batch; here 32 + 8 pairs on two
    lanes.  The throughput path as it ships (no taps: the tapped kernels resolve exactly tied logits in another order, which 40 pairs
    are enough to meet) against the reference's outputs: Z and the matching scores bounded as everywhere on this path (a flipped
    top-k near tie moves Z by a few 1e-4 around one keypoint), and every match the same except arg-max NEAR TIES by this path's own Z -
    the batch holds one whose two candidates the reference's fp64 Z separates by 1.3e-6 (pair 18, column 71)."""
    from parity_util import near_tie_mismatches, PLAIN_MAX, PLAIN_FRAC
    g = _g(golden_dir, 'cfg_n512_L9_S100_b40')
    net, _, (B, n, m, L) = _build(g)
    seed, first_pair = int(g['meta'][5]), int(g['meta'][6])
    data = synth.make_batch(B, n, m, first_pair=first_pair)
    dev = {k: v.to(DEV) for k, v in data.items()}
    with torch.no_grad():
        m0, m1, s0, s1, Z = net._run(dev['keypoints0'], dev['scores0'], dev['descriptors0'], dev['keypoints1'], dev['scores1'], dev['descriptors1'], want_Z=True)
    torch.cuda.synchronize()
    net.check(DEV)
    ties = near_tie_mismatches(Z, m0, m1, g['default_matches0'], g['default_matches1'], 1e-3)
    Zc = Z.cpu().double().numpy()
    sub = int(g['sub'])
    mine = np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    err = np.abs(mine - ref_Z).reshape(B, -1)
    es = max(np.abs(s0.cpu().double().numpy() - g['default_mscores0']).max(), np.abs(s1.cpu().double().numpy() - g['default_mscores1']).max())
    print(f'[parity] cfg_n512_L9_S100_b40 (throughput path): arg-max near ties decided the other way {ties}; max|dZ| {err.max():.2e}, '
          f'entries beyond 1e-4 {(err > Z_TOL).mean():.2e}, mscores {es:.2e}; pairs within the literal 1e-4: {int((err.max(1) < Z_TOL).sum())}/{B}')
    assert len(ties) <= 4, ties
    assert err.max() < PLAIN_MAX and (err > Z_TOL).mean() < PLAIN_FRAC and es < PLAIN_MAX


@pytest.mark.parametrize('name', ['var_n256_L4_S20', 'var_n512_L9_S100', 'var_n400m512_L9_S100'])
def test_config_shape_variants_golden(golden_dir, name):
    """All four extraction branches of mdgat.py:441-483 at the BASELINE shapes and a ragged 400 x 512 pair, against the
    REFERENCE's own outputs (tests/golden/var_*.npz): matches bit-identical, matching scores to 1e-4 except where a flipped
    top-k near-tie moved them (bounded like Z in parity_util.assert_plain)."""
    from parity_util import PLAIN_MAX
    g = _g(golden_dir, name)
    for tag, (loss_method, mutual) in {'default': ('triplet_loss', False), 'mutual': ('triplet_loss', True),
                                       'sg': ('superglue', False), 'sgmutual': ('superglue', True)}.items():
        if f'{tag}_matches0' not in g:
            continue
        net, data, (B, n, m, L) = _build(g, loss_method=loss_method, mutual_check=mutual)
        with torch.no_grad():
            out = net(data)
        np.testing.assert_array_equal(out['matches0'].cpu().numpy(), g[f'{tag}_matches0'], err_msg=tag)
        np.testing.assert_array_equal(out['matches1'].cpu().numpy(), g[f'{tag}_matches1'], err_msg=tag)
        e0 = np.abs(out['matching_scores0'].cpu().numpy() - g[f'{tag}_mscores0'])
        e1 = np.abs(out['matching_scores1'].cpu().numpy() - g[f'{tag}_mscores1'])
        print(f'[parity] {name} {tag}: matches identical to the reference, max|d mscores| {max(e0.max(), e1.max()):.2e}')
        assert max(e0.max(), e1.max()) < PLAIN_MAX and (np.concatenate([e0.ravel(), e1.ravel()]) > Z_TOL).mean() < 1e-2, tag


def test_dataparallel_dropin_like_test_py(golden_dir):
    """The call sequence of test.py:156-201: DataParallel wrapper, 'module.'-prefixed checkpoint,
    net.double().eval() before every forward, dict in -> dict out."""
    g = _g(golden_dir, 'fwd_n64_L4_S20')
    B, n, m, L, S, seed, first_pair = [int(x) for x in g['meta']]
    k = [None if x < 0 else int(x) for x in g['k']]
    net = MDGAT(synth.default_config(L=L, k=k, sinkhorn_iterations=S))
    net = torch.nn.DataParallel(net)
    net.load_state_dict({'module.' + kk: v for kk, v in synth.make_state_dict(L=L, seed=seed).items()})
    net.to(DEV)
    with torch.no_grad():
        for _ in range(2):
            net.double().eval()
            pred = synth.make_batch(B, n, m, first_pair=first_pair)
            pred = {kk: v.cuda() for kk, v in pred.items()}
            data = net(pred)
            np.testing.assert_array_equal(data['matches0'][0].cpu().detach().numpy(), g['default_matches0'][0])
            np.testing.assert_array_equal(data['matches1'][0].cpu().detach().numpy(), g['default_matches1'][0])


def test_empty_keypoints_and_errors(golden_dir):
    g = _g(golden_dir, 'edge_cases')
    net = MDGAT(synth.default_config(L=1, k=[], sinkhorn_iterations=5)).double()
    net.load_state_dict(synth.make_state_dict(L=1, seed=3))
    net = net.double().eval().to(DEV)
    data = synth.make_batch(1, 8, 8, device=DEV)
    data['keypoints0'] = data['keypoints0'][:, :0]
    out = net(data)
    assert out['skip_train'] is True
    for a, b in (('matches0', 'empty_matches0'), ('matches1', 'empty_matches1'),
                 ('matching_scores0', 'empty_mscores0'), ('matching_scores1', 'empty_mscores1')):
        assert tuple(out[a].shape) == g[b].shape
        np.testing.assert_array_equal(out[a].cpu().numpy(), g[b])
    # k larger than the number of keypoints raises like torch.topk does in the reference
    net2 = MDGAT(synth.default_config(L=1, k=[16, None], sinkhorn_iterations=5)).double()
    net2.load_state_dict(synth.make_state_dict(L=1, seed=3))
    net2 = net2.eval().to(DEV)
    with pytest.raises(RuntimeError, match='exceeds the number of keys'):
        net2(synth.make_batch(1, 8, 8, device=DEV))
    # k == M behaves as full attention
    data = synth.make_batch(1, 32, 32, device=DEV)
    net3 = MDGAT(synth.default_config(L=1, k=[32, 32], sinkhorn_iterations=10)).double()
    net3.load_state_dict(synth.make_state_dict(L=1, seed=3))
    net3 = net3.eval().to(DEV)
    Z = net3.match(data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'],
                   data['scores0'], data['scores1'], return_scores=True)[4]
    assert np.abs(Z.cpu().double().numpy() - g['keqM_Z_dyn']).max() < Z_TOL
    # all-dustbin frame
    net4 = MDGAT(synth.default_config(L=1, k=[], sinkhorn_iterations=10)).double()
    net4.load_state_dict(synth.make_state_dict(L=1, seed=3, bin_score=50.0))
    net4 = net4.eval().to(DEV)
    out = net4(data)
    np.testing.assert_array_equal(out['matches0'].cpu().numpy(), g['alldust_matches0'])
    assert (out['matching_scores0'] == 0).all() and (out['matching_scores1'] == 0).all()


def test_bench_shape_properties():
    """Full BASELINE size (B=64, N=M=512, L=9, S=100): properties that need no CPU oracle."""
    B, n, L = 64, 512, 9
    net = MDGAT(synth.default_config(L=L)).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=0))
    net = net.eval().to(DEV)
    data = synth.make_batch(B, n, n, device=DEV, dtype=torch.float32)
    args = (data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'], data['scores0'], data['scores1'])
    m0, m1, s0, s1, Z = net.match(*args, return_scores=True)
    # (1) determinism: a second run is bit-identical
    m0b, m1b, s0b, s1b, Zb = net.match(*args, return_scores=True)
    assert torch.equal(m0, m0b) and torch.equal(Z, Zb)
    # (2) batch independence: pair 5 alone gives the same bits as inside the batch
    sl = [a[5:6] for a in args]
    m0s, m1s, s0s, s1s, Zs = net.match(*sl, return_scores=True)
    assert torch.equal(m0s[0], m0[5]) and torch.equal(Zs[0], Z[5])
    # (3) column marginals of the transport plan are exact after the final v-update
    col = torch.logsumexp(Z.double(), dim=1)
    assert col[:, :n].abs().max() < 1e-4 and (col[:, n] - np.log(n)).abs().max() < 1e-4
    # (4) matches are consistent with Z: arg-max rows / dustbin rule, scores = exp(max)
    r0, r1, rs0, rs1 = O.extract_matches(Z[:4].cpu().double())
    assert torch.equal(r0, m0[:4].cpu()) and torch.equal(r1, m1[:4].cpu())
    assert (rs0 - s0[:4].cpu().double()).abs().max() < 1e-5
    # (5) permutation equivariance: permuting frame-1 keypoints permutes the assignment
    perm = torch.randperm(n, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    argsp = (args[0][:2], args[1][:2], args[2][:2, perm], args[3][:2, perm], args[4][:2], args[5][:2, perm])
    m0p, m1p, s0p, s1p, Zp = net.match(*argsp, return_scores=True)
    assert (Zp[:, :, :n] - Z[:2][:, :, perm]).abs().max() < 1e-3
    # (6) two pairs of the same batch against the fp64 oracle, north-star bar in the attributed form (parity_util.py);
    #     by (2) the result of a pair does not depend on the batch it is in
    sd = synth.make_state_dict(L=L, seed=0)
    cpu = {k: v[:2].cpu().double() for k, v in data.items()}
    res = attributed_parity(net, synth.default_config(L=L), sd, cpu, DEV)
    assert_attributed(res, 'bench shape, pairs 0-1')
    assert torch.equal(res['out'][0], m0[:2]) and (res['out'][4] - Z[:2]).abs().max() <= 5e-6       # (tapped kernels: parity_util)
    assert_plain_vs_oracle(res, synth.default_config(L=L), sd, cpu, 'bench shape, pairs 0-1')


@pytest.mark.parametrize('n,L,S', [(256, 4, 20), (512, 9, 100)])
def test_full_attention_configs_strict(n, L, S):
    """Without dynamic layers (k=[] - the SuperGlue-style configuration) nothing is discontinuous: the whole
    fp32 pipeline must stay within 1e-4 of the fp64 oracle on Z at the BASELINE shapes, matches identical."""
    cfg = synth.default_config(L=L, k=[], sinkhorn_iterations=S)
    sd = synth.make_state_dict(L=L, seed=0)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to(DEV)
    data = synth.make_batch(1, n, n, first_pair=3)
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data, cap)
    d = {k: v.to(DEV) for k, v in data.items()}
    m0, m1, s0, s1, Z = net.match(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'],
                                  d['scores0'], d['scores1'], return_scores=True)
    err = (Z.cpu().double() - cap['Z']).abs().max().item()
    print('full-attention', n, L, S, 'max|dZ|', err)
    assert err < Z_TOL
    assert torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1'])
    assert (s0.cpu().double() - ref['matching_scores0']).abs().max() < Z_TOL


def test_configs3_per_gpu_workload():
    """BASELINE configs[3]: 4096 pairs sharded 8-way = 512 pairs per GPU (N=M=512, L=9, S=100, fp32).  One rank's
    workload runs as 8 slices of 64 pairs inside mdgat_forward (api.hip: forward_sliced): deterministic, every pair
    independent of its batch and of the slice it lands in (pairs either side of a slice boundary and the last one,
    alone vs in the batch: bit-identical), and sampled pairs against the fp64 oracle in both forms of parity_util.py."""
    B, n, L, S = 512, 512, 9, 100
    cfg = synth.default_config(L=L, sinkhorn_iterations=S)
    sd = synth.make_state_dict(L=L, seed=0)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.eval().to(DEV)
    data = synth.make_batch(B, n, n, device=DEV, dtype=torch.float32)
    args = (data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'], data['scores0'], data['scores1'])
    m0, m1, s0, s1, Z = net.match(*args, return_scores=True)
    again = net.match(*args, return_scores=True)
    for a, b in zip((m0, m1, s0, s1, Z), again):
        assert torch.equal(a, b)                                    # determinism
    for p in (0, 63, 64, 300, 511):                                 # 63 | 64: the first slice boundary
        alone = net.match(*[a[p:p + 1] for a in args], return_scores=True)
        for a, b in zip((m0, m1, s0, s1, Z), alone):
            assert torch.equal(a[p], b[0]), p
    col = torch.logsumexp(Z.double(), dim=1)                        # exact column marginals, all 512 pairs
    assert col[:, :n].abs().max() < 1e-4 and (col[:, n] - np.log(n)).abs().max() < 1e-4
    assert (m0 >= 0).any(dim=1).all()                               # every pair found matches (half of its keypoints are shared)
    idx = [63, 64, 300, 511]
    cpu = {k: v[idx].cpu().double() for k, v in data.items()}
    res = attributed_parity(net, cfg, sd, cpu, DEV)
    assert_attributed(res, 'configs[3], pairs 63 / 64 / 300 / 511')
    from parity_util import TAP_EPS
    assert torch.equal(res['out'][0], m0[idx]) and (res['out'][4] - Z[idx]).abs().max() <= TAP_EPS     # (tapped kernels: parity_util)
    assert_plain_vs_oracle(res, cfg, sd, cpu, 'configs[3], pairs 63 / 64 / 300 / 511')


def test_f16_attention_mode_configs2():
    """BASELINE configs[2]: q, k, v and the probabilities as single f16 values in the attention products (fp32
    accumulation, softmax statistics and Sinkhorn).  A throughput mode OUTSIDE the 1e-4 bar; the test pins what it is on
    8 pairs: engaged (results differ from the fp32 path), close (max|dZ| against the fp64 oracle bounded), and how far the
    ASSIGNMENT is from the reference's: SURVEY section 7 expected the argmax to survive reduced-precision attention - it
    does on most pairs but not bit-exactly (a handful of keypoints per 1024 change their match, near-ties between an
    inner column and the dustbin), so the bound is the measured identity rate with margin, not equality."""
    L, S, n, B = 9, 100, 512, 8
    cfg = synth.default_config(L=L, sinkhorn_iterations=S)
    sd = synth.make_state_dict(L=L, seed=0)
    data = synth.make_batch(B, n, n, first_pair=5)
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data, cap)
    d = {k: v.to(DEV) for k, v in data.items()}
    out = {}
    for dt in ('fp32', 'f16'):
        net = MDGAT(dict(cfg, attention_dtype=dt)).double()
        net.load_state_dict(sd)
        net = net.double().eval().to(DEV)
        out[dt] = net.match(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'],
                            d['scores0'], d['scores1'], return_scores=True)
    e32 = (out['fp32'][4].cpu().double() - cap['Z']).abs().max().item()
    e16 = (out['f16'][4].cpu().double() - cap['Z']).abs().amax(dim=(1, 2))
    same0 = (out['f16'][0].cpu() == ref['matches0']).double().mean(dim=1)
    same1 = (out['f16'][1].cpu() == ref['matches1']).double().mean(dim=1)
    agree = torch.minimum(same0, same1)
    exact = int(((same0 == 1) & (same1 == 1)).sum())
    print('attention_dtype f16, 8 pairs: max|dZ| per pair', [f'{x:.2e}' for x in e16.tolist()], '(fp32 path', f'{e32:.2e})',
          'matches identical per pair', [f'{x:.4f}' for x in agree.tolist()], 'pairs with bit-identical matches', exact, 'of', B)
    assert not torch.equal(out['f16'][4], out['fp32'][4])
    assert e16.max() < 2e-2
    assert agree.min() >= 0.995 and agree.mean() >= 0.998       # measured: 6 of 8 pairs bit-identical, the other two 0.998
    # the fp32 path on the same pairs: matches identical to the reference (the argmax bar is the fp32 path's)
    assert torch.equal(out['fp32'][0].cpu(), ref['matches0']) and torch.equal(out['fp32'][1].cpu(), ref['matches1'])
    with pytest.raises(ValueError):
        MDGAT(dict(cfg, attention_dtype='int8'))
    # a configs[2]-sized batch (512 pairs, sliced) in f16 mode: deterministic, batch-independent across a slice boundary
    big = synth.make_batch(512, n, n, device=DEV, dtype=torch.float32)
    bargs = (big['keypoints0'], big['descriptors0'], big['keypoints1'], big['descriptors1'], big['scores0'], big['scores1'])
    r1 = net.match(*bargs)
    r2 = net.match(*bargs)
    assert all(torch.equal(a, b) for a, b in zip(r1, r2))
    for p in (63, 64, 511):
        alone = net.match(*[a[p:p + 1] for a in bargs])
        assert all(torch.equal(a[p], b[0]) for a, b in zip(r1, alone)), p
    # shapes no f16 kernel covers ignore the flag: bit-identical to the default path
    small = synth.make_batch(2, 100, 75, first_pair=9)
    ds = {k: v.to(DEV) for k, v in small.items()}
    zs = []
    for dt in ('fp32', 'f16'):
        net = MDGAT(dict(synth.default_config(L=2, k=[16, None], sinkhorn_iterations=10), attention_dtype=dt))
        net.load_state_dict(synth.make_state_dict(L=2, seed=3))
        net = net.double().eval().to(DEV)
        zs.append(net.match(ds['keypoints0'], ds['descriptors0'], ds['keypoints1'], ds['descriptors1'],
                            ds['scores0'], ds['scores1'], return_scores=True)[4])
    assert torch.equal(zs[0], zs[1])


@pytest.mark.parametrize('n,m,L,S,k', [(2048, 2048, 9, 200, None), (1024, 700, 2, 50, [128, None, 64, None]),
                                       (600, 1300, 2, 30, [])])
def test_large_frames(n, m, L, S, k):
    """More than 512 keypoints per frame (BASELINE.json configs[4]: N=2048, L=9, 200 iterations): windowed full
    attention, the wide dynamic-attention kernel and the tiled Sinkhorn.  Against the fp64 oracle on the same
    inputs: 1e-4 on Z and identical matches, for dynamic configurations in the attributed form of parity_util.py."""
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S)
    sd = synth.make_state_dict(L=L, seed=0)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to(DEV)
    data = synth.make_batch(1, n, m, first_pair=1)
    if k == []:
        cap = {}
        ref = O.mdgat_forward(sd, cfg, data, cap)
        d = {kk: v.to(DEV) for kk, v in data.items()}
        m0, m1, s0, s1, Z = net.match(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'],
                                      d['scores0'], d['scores1'], return_scores=True)
        err = (Z.cpu().double() - cap['Z']).abs()
        print('large', n, m, L, S, 'max|dZ|', err.max().item(), 'median', err.median().item())
        assert err.max() < Z_TOL
        assert torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1'])
    else:
        res = attributed_parity(net, cfg, sd, data, DEV)
        assert_attributed(res, f'large {n}x{m} L={L} S={S}')
        assert_plain_vs_oracle(res, cfg, sd, data, f'large {n}x{m} L={L} S={S}')


@pytest.mark.parametrize('B,n,m,k', [(3, 37, 53, []), (2, 130, 75, []), (5, 20, 44, [8, None]), (1, 1, 9, []), (2, 128, 256, [])])
def test_ragged_shapes_vs_oracle(B, n, m, k):
    """Keypoint counts that are not multiples of anything (tiles of 128 / waves of 16 keypoints straddle frames and
    pairs, the last tile is partial, V^T goes through the element-wise store path): full-attention configurations
    must match the fp64 oracle to 1e-4 on Z with identical matches, dynamic ones up to near-tie flips."""
    L = 2
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=15)
    sd = synth.make_state_dict(L=L, seed=3)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to(DEV)
    data = synth.make_batch(B, n, m, first_pair=2)
    if k != []:
        res = attributed_parity(net, cfg, sd, data, DEV)
        assert_attributed(res, f'ragged {B}x{n}x{m}')
        assert_plain_vs_oracle(res, cfg, sd, data, f'ragged {B}x{n}x{m}')
        return
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data, cap)
    d = {kk: v.to(DEV) for kk, v in data.items()}
    m0, m1, s0, s1, Z = net.match(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'],
                                  d['scores0'], d['scores1'], return_scores=True)
    err = (Z.cpu().double() - cap['Z']).abs()
    assert err.max() < Z_TOL, err.max()
    assert torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1'])


def test_forward_falls_back_when_sinkhorn_loses_a_partner(monkeypatch):
    """Whole forward with the Sinkhorn fallback forced (MDGAT_SK_FORCE_FALLBACK, see test_gpu_ops.py): the matches come
    from the streaming kernel's Z inside the same call - with and without a Z requested, sliced batches included - are
    identical to the normal path's, the handle reports the fallback as information and stays usable."""
    cfg = synth.default_config(L=2, k=[128, None, 64, None], sinkhorn_iterations=40)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5))
    net = net.eval().to(DEV)
    d = synth.make_batch(100, 512, 512, device=DEV, dtype=torch.float32)          # 100 pairs: two slices of 50
    args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
    ref = net.match(*args, return_scores=True)
    assert net.check(DEV) == {'sinkhorn_fallback': False}
    monkeypatch.setenv('MDGAT_SK_FORCE_FALLBACK', '1')
    with_z = net.match(*args, return_scores=True)
    without_z = net.match(*args)
    status = net.check(DEV)
    monkeypatch.delenv('MDGAT_SK_FORCE_FALLBACK')
    assert status == {'sinkhorn_fallback': True}
    assert torch.isfinite(with_z[4]).all() and (with_z[4] - ref[4]).abs().max() < 1e-4
    # the streaming kernel is another fp32 evaluation of the same iteration (Z agrees to ~1e-5): an arg-max that is a
    # near-tie between two entries of Z may fall the other way - a handful of the 102 400 keypoints at most, each one a tie
    # within 1e-4 in the normal path's Z
    r0, r1, rs0, rs1 = O.extract_matches(with_z[4].cpu().double())
    assert torch.equal(r0, with_z[0].cpu()) and torch.equal(r1, with_z[1].cpu())      # the matches ARE the arg-maxes of the fallback's Z
    for a, b, c in zip(ref[:2], with_z[:2], without_z[:2]):
        assert torch.equal(b, c)                                                      # with or without Z requested: same bits
        assert int((a != b).sum()) <= 4, int((a != b).sum())
    for a, b, c in zip(ref[2:4], with_z[2:4], without_z[2:4]):
        assert torch.equal(b, c)
        assert ((a - b).abs() > 1e-4).sum() <= 4
    again = net.match(*args, return_scores=True)
    assert all(torch.equal(a, b) for a, b in zip(ref, again)) and net.check(DEV) == {'sinkhorn_fallback': False}


def test_two_streams_run_forwards_concurrently():
    """Two B=64 forwards of BASELINE configs[1] in flight at once on two streams of one device (each stream has its own
    workspace; the Sinkhorn cluster kernels of both are on the device together, every workgroup of which waits for its
    partners): both must return what they return alone, bit for bit - finite, oracle-consistent (the serial results are
    held to the oracle by the tests above) - whichever path the Sinkhorn launches took."""
    L = 9
    net = MDGAT(synth.default_config(L=L)).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=0))
    net = net.eval().to(DEV)
    batches = [synth.make_batch(64, 512, 512, first_pair=64 * i, device=DEV, dtype=torch.float32) for i in range(2)]
    args = [(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1']) for d in batches]
    serial = [net.match(*a, return_scores=True) for a in args]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV) for _ in range(2)]
    for rounds in range(3):
        outs = [None, None]
        for _ in range(4):                                   # several forwards per stream back to back: the launches interleave
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i] = net.match(*args[i], return_scores=True)
        torch.cuda.synchronize()
        status = net.check(DEV)
        for i in range(2):
            assert torch.isfinite(outs[i][4]).all()
            if status['sinkhorn_fallback']:                  # a launch was redone by the streaming kernel: Z to round-off, near-tie arg-maxes apart
                assert int((outs[i][0] != serial[i][0]).sum()) <= 4 and int((outs[i][1] != serial[i][1]).sum()) <= 4
                assert (outs[i][4] - serial[i][4]).abs().max() < 1e-4
            else:
                for a, b in zip(outs[i], serial[i]):
                    assert torch.equal(a, b)
    print('two-stream forwards: sinkhorn fallback taken in the last round:', status['sinkhorn_fallback'])


def test_f16_operand_range_is_guarded():
    """Every operand of the matrix products is an f16 head + f16 residual: |activation| must stay
    below 65504.  Synthetic weights keep activations at O(1-10); a checkpoint that does not is NOT silently turned into
    inf / NaN: the kernels flag values >= 6e4 or non-finite ones where the activations pass (layer inputs, scores) and
    the failing call raises in forward(); match() - asynchronous - reports through check() or the handle's next call."""
    L = 2
    cfg = synth.default_config(L=L, k=[16, None, 16, None], sinkhorn_iterations=10)
    data = synth.make_batch(2, 64, 64, device=DEV)
    args = (data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'], data['scores0'], data['scores1'])
    # (weights below the range check of mdgat_load_weights - test_weights_beyond_the_f16_range_are_refused -, activations beyond)
    for key, factor in (('kenc.encoder.9.weight', 3e4), ('gnn.layers.1.mlp.3.weight', 1e5), ('final_proj.weight', 3e4)):      # (final_proj: caught by the score kernel's operand guard)
        sd = synth.make_state_dict(L=L, seed=1)
        sd[key] = sd[key] * factor
        net = MDGAT(cfg).double()
        net.load_state_dict(sd)
        net = net.double().eval().to(DEV)
        with pytest.raises(RuntimeError, match='f16 operand range'):
            with torch.no_grad():
                net(data)
        net.match(*args)                                     # asynchronous: no error yet ...
        with pytest.raises(RuntimeError, match='f16 operand range'):
            net.check(DEV)                                   # ... until the caller asks
        net.match(*args)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match='f16 operand range'):
            net.match(*args)                                 # ... or the next call on the handle finds it
        net.check(DEV)                                       # (reported once: the status is clear again)
    # a gain of 10^5 on q alone is no overflow any more: the packer balances q against k (pack.py, gauge fixing), the logits
    # grow a 10^5-fold and the rows become one-hot - finite, and the same matches as the fp64 oracle's
    sd = synth.make_state_dict(L=L, seed=1)
    sd['gnn.layers.0.attn.proj.0.weight'] = sd['gnn.layers.0.attn.proj.0.weight'] * 1e5
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.eval().to(DEV)
    with torch.no_grad():
        out = net(data)
    assert torch.isfinite(out['matching_scores0']).all() and net.check(DEV) == {'sinkhorn_fallback': False}
    sd = synth.make_state_dict(L=L, seed=1)                  # the unscaled weights pass
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to(DEV)
    with torch.no_grad():
        out = net(data)
    assert torch.isfinite(out['matching_scores0']).all() and net.check(DEV) == {'sinkhorn_fallback': False}


def test_deepcopy_of_a_running_module():
    """copy.deepcopy of a module that has already run: the copy builds its own library handle on first use and gives the
    same bits; the original keeps working and both can be freed independently."""
    import copy
    net = MDGAT(synth.default_config(L=2, k=[16, None, 8, None], sinkhorn_iterations=10)).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=1))
    net = net.eval().to(DEV)
    d = synth.make_batch(2, 64, 48, device=DEV, dtype=torch.float32)
    args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
    ref = net.match(*args, return_scores=True)
    clone = copy.deepcopy(net)
    assert clone._states == {} and len(net._states) == 1
    out = clone.match(*args, return_scores=True)
    assert all(torch.equal(a, b) for a, b in zip(ref, out))
    assert clone._states[0].handle.value != net._states[0].handle.value
    del clone
    assert all(torch.equal(a, b) for a, b in zip(ref, net.match(*args, return_scores=True)))


def test_fuzz_short():
    """15 s of tools/fuzz_forward.py: random B, N, M (1 ... 700, not multiples of anything), L, Sinkhorn iterations, top-k
    schedules, all four extraction modes, bin scores - Z within 1e-4 of the oracle with the HIP selections forced, matches
    and scores exactly what the extraction rules of mdgat.py:441-483 make of that Z, k keys per dynamic row, clean status.
    (2 250 cases of it ran clean in round 3.)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('fuzz_forward', os.path.join(root, 'tools', 'fuzz_forward.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    cases, fails, worst = fz.run(15.0, seed=3, verbose=True)
    assert cases >= 20 and fails == 0 and worst <= 1e-4


def test_repeatable_bitwise():
    """Same inputs, same outputs, bit for bit: the cross-workgroup sums of the Sinkhorn kernel are taken in a fixed
    order and nothing else in the path depends on scheduling."""
    cfg = synth.default_config(L=3, k=[64, None, 32, None, None, None], sinkhorn_iterations=50)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=3, seed=5))
    net = net.eval().to(DEV)
    d = synth.make_batch(9, 512, 512, device=DEV, dtype=torch.float32)
    args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
    ref = [t.clone() for t in net.match(*args, return_scores=True)]
    for _ in range(3):
        out = net.match(*args, return_scores=True)
        for a, b in zip(ref, out):
            assert torch.equal(a, b)


@pytest.mark.parametrize('bin_score', [1.0, 60.0])
def test_large_batch_runs_in_slices(bin_score):
    """A large batch runs as balanced slices inside mdgat_forward (api.hip: forward_batched).  Pairs are
    independent, so the result must be bit-identical to running the same pairs in two separate calls - including the one
    batch-wide rule of the reference (mdgat.py:465-467: no frame-0 keypoint matched anywhere -> all scores zero; bin_score 60
    sends every keypoint to the dustbin)."""
    B, n = 130, 512
    net = MDGAT(synth.default_config(L=2, k=[128, None, 64, None], sinkhorn_iterations=10)).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5, bin_score=bin_score))
    net = net.to(DEV).eval()
    data = synth.make_batch(B, n, n, device=DEV)
    with torch.no_grad():
        whole = net(data)                                     # 130 pairs: six slices of 22 on two lanes
        halves = [net({k: (v[i:i + 65] if torch.is_tensor(v) else v) for k, v in data.items()}) for i in (0, 65)]   # separate calls
    for key in ('matches0', 'matches1', 'matching_scores0', 'matching_scores1'):
        assert torch.equal(whole[key], torch.cat([h[key] for h in halves])), key
    if bin_score > 10:
        assert (whole['matches0'] == -1).all() and (whole['matches1'] == -1).all()
        assert (whole['matching_scores0'] == 0).all() and (whole['matching_scores1'] == 0).all()
    else:
        assert (whole['matches0'] >= 0).any()


@pytest.mark.parametrize('B,n,m,bin_score', [(64, 512, 512, 1.0), (100, 200, 256, 1.0), (131, 512, 480, 1.0), (70, 512, 512, 60.0),
                                            (33, 512, 512, 1.0)])
def test_two_lanes_equal_one_lane(B, n, m, bin_score):
    """A batch of more than 32 768 keypoints runs as an even number of slices alternating between the caller's stream and the
    handle's second stream (api.hip: forward_batched).  Bit-identical to the same batch on one stream - matches, scores and Z,
    ragged slice sizes, the batch-wide all-dustbin rule (mdgat.py:465-467; bin score 60) - and the call stays ordered on the
    caller's stream: the outputs are read right behind it, and a second call reuses both workspaces."""
    cfg = synth.default_config(L=2, k=[128, None, 64, None], sinkhorn_iterations=10)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5, bin_score=bin_score))
    net = net.to(DEV).eval()
    data = synth.make_batch(B, n, m, device=DEV, dtype=torch.float32)
    args = (data['keypoints0'], data['descriptors0'], data['keypoints1'], data['descriptors1'], data['scores0'], data['scores1'])
    with torch.no_grad():
        net.set_lanes(1)
        one = net.match(*args, return_scores=True)
        torch.cuda.synchronize()
        net.set_lanes(2)
        two = [net.match(*args, return_scores=True) for _ in range(3)]      # (back to back: the lanes' workspaces are reused)
        side = torch.cuda.Stream(DEV)                                       # and from a stream of the caller's own
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            two.append(net.match(*args, return_scores=True))
        torch.cuda.synchronize()
    for out in two:
        for a, b in zip(one, out):
            assert torch.equal(a, b)
    if bin_score > 10:
        assert (one[0] == -1).all() and (one[2] == 0).all() and (one[3] == 0).all()
    else:
        assert (one[0] >= 0).any()
    assert not net.check(DEV)['sinkhorn_fallback']
    # the Python wrapper's forward(): the batch-wide rule and the status words behave the same with two lanes
    with torch.no_grad():
        out = net(data)
    assert torch.equal(out['matches0'], one[0]) and (out['matching_scores0'].dtype == torch.int64) == (bin_score > 10)


def test_forward_is_capturable_as_a_hip_graph():
    """The library allocates nothing and launches only on the stream it is given - plus, for a batch on two lanes, on its own
    second stream, which is forked from and joined to the caller's stream with events: a stream capture of the caller's
    stream therefore takes in both lanes.  torch.cuda.CUDAGraph (= hipGraph on ROCm) of a B = 64 forward replays to the
    bits of the eager call, also after the inputs have changed in place."""
    cfg = synth.default_config(L=2, k=[128, None, 64, None], sinkhorn_iterations=10)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5))
    net = net.to(DEV).eval()
    d = synth.make_batch(64, 512, 512, device=DEV, dtype=torch.float32)
    inputs = tuple(d[k] for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1'))
    other = synth.make_batch(64, 512, 512, first_pair=500, device=DEV, dtype=torch.float32)
    with torch.no_grad():
        eager = [t.clone() for t in net._run(*inputs, want_Z=True)]
        eager_other = [t.clone() for t in net._run(*(other[k] for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1',
                                                                          'scores1', 'descriptors1')), want_Z=True)]
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):                                  # (warm-up on the capture side, as torch asks)
            net._run(*inputs, want_Z=True)
        torch.cuda.current_stream(DEV).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = net._run(*inputs, want_Z=True)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(eager, out):
            assert torch.equal(a, b)
        for k, t in zip(('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1'), inputs):
            t.copy_(other[k])
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(eager_other, out):
            assert torch.equal(a, b)
    assert not net.check(DEV)['sinkhorn_fallback']


def test_match_frames_raw_records():
    """Raw 37-float keypoint records (load_data.py:152-165) straight into the encoder kernel, FPFH normalisation
    (load_data.py:290-292) fused: same result as decoding with the oracle's restatement of the loader."""
    L, S, B, n, m = 2, 20, 2, 200, 168
    cfg = synth.default_config(L=L, k=[64, None, 32, None], sinkhorn_iterations=S)
    sd = synth.make_state_dict(L=L, seed=2)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to(DEV)
    rs = np.random.RandomState(11)
    def frames(count):
        rec = np.zeros((B, count, 37), dtype=np.float32)
        rec[..., :3] = 20.0 * rs.standard_normal((B, count, 3))
        rec[..., 3] = rs.uniform(0, 1, (B, count))
        rec[..., 4:] = rs.uniform(0, 200, (B, count, 33))        # un-normalised histogram-like FPFH
        return rec
    r0, r1 = frames(n), frames(m)
    k0, s0, d0 = O.decode_frames(r0)
    k1, s1, d1 = O.decode_frames(r1)
    data = {'keypoints0': k0, 'scores0': s0, 'descriptors0': d0, 'keypoints1': k1, 'scores1': s1, 'descriptors1': d1}
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data, cap)
    m0, m1, c0, c1, Z = net.match_frames(torch.from_numpy(r0).to(DEV), torch.from_numpy(r1).to(DEV), return_scores=True)
    assert (Z.cpu().double() - cap['Z']).abs().max() < Z_TOL
    assert torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1'])
    # and the array entry point agrees with the record entry point on pre-decoded inputs
    m0b = net.match(k0.to(DEV), d0.to(DEV), k1.to(DEV), d1.to(DEV), s0.to(DEV), s1.to(DEV))[0]
    assert torch.equal(m0b, m0)


def test_match_frames_vs_reference_loader_outputs(golden_dir):
    """The record decode fused into the encoder kernel against the REFERENCE loader's outputs (tests/golden/aux_loader.npz:
    SparseDataset.__getitem__, load_data.py:146-169 and 290-295): matching the raw 37-float records must give the same
    result as matching the keypoints / saliency / normalised descriptors the reference's loader made of them."""
    g = np.load(os.path.join(golden_dir, 'aux_loader.npz'))
    L = 2
    cfg = synth.default_config(L=L, k=[32, None, 16, None], sinkhorn_iterations=20)
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=4))
    net = net.double().eval().to(DEV)
    for j in range(int(g['n_items'])):
        r0 = torch.from_numpy(g[f'item{j}_rec0']).to(DEV)
        r1 = torch.from_numpy(g[f'item{j}_rec1']).to(DEV)
        a = net.match_frames(r0, r1, return_scores=True)
        t = {k: torch.from_numpy(g[f'item{j}_{k}']).to(DEV) for k in ('keypoints0', 'keypoints1', 'descriptors0', 'descriptors1',
                                                                         'scores0', 'scores1')}
        b = net.match(t['keypoints0'], t['descriptors0'], t['keypoints1'], t['descriptors1'], t['scores0'], t['scores1'],
                      return_scores=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert (a[4] - b[4]).abs().max() < 1e-5      # (the kernel normalises in fp32 with its own summation order)


def test_replicas_made_by_torch_replicate(golden_dir):
    """torch.nn.parallel.replicate() - what DataParallel.forward runs whenever it has more than one device (test.py:158,
    train.py:192-196) - strips the parameters off the replicas; the replica must run on the blob its owner packed."""
    g = _g(golden_dir, 'fwd_n64_L4_S20')
    B, n, m, L, S, seed, first_pair = [int(x) for x in g['meta']]
    k = [None if x < 0 else int(x) for x in g['k']]
    net = MDGAT(synth.default_config(L=L, k=k, sinkhorn_iterations=S)).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=seed))
    net = net.double().eval().to(DEV)
    data = synth.make_batch(B, n, m, first_pair=first_pair, device=DEV)
    for _ in range(2):                                  # DataParallel replicates before every forward
        replica = torch.nn.parallel.replicate(net, [0])[0]
        assert 'bin_score' not in replica._parameters
        with torch.no_grad():
            out = replica(data)
        np.testing.assert_array_equal(out['matches0'].cpu().numpy(), g['default_matches0'])
        np.testing.assert_array_equal(out['matches1'].cpu().numpy(), g['default_matches1'])
    assert len(net._states) == 1                        # one handle per device, shared by owner and replicas


def test_load_packed_blob_on_device():
    """The multi-rank weight path (shard.broadcast_weights -> load_packed -> mdgat_load_weights(on_device=1)): a rank
    whose own parameters are random init runs the broadcast blob, bit-identically to the rank that packed it, also
    after the net.double().eval() of test.py:193."""
    L = 3
    cfg = synth.default_config(L=L, k=[64, None, 32, None], sinkhorn_iterations=30)
    src = MDGAT(cfg).double()
    src.load_state_dict(synth.make_state_dict(L=L, seed=7))
    src = src.eval().to(DEV)
    d = synth.make_batch(3, 200, 256, device=DEV, dtype=torch.float32)
    args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
    ref = src.match(*args, return_scores=True)
    blob = torch.from_numpy(src.packed_weights()).to(DEV)
    other = MDGAT(cfg).eval().to(DEV)                   # random init: NOT the weights that must run
    other.load_packed(blob)
    for cast in (lambda x: x, lambda x: x.double().eval(), lambda x: x.float(), lambda x: x.to(DEV)):
        other = cast(other)
        out = other.match(*args, return_scores=True)
        for a, b in zip(ref, out):
            assert torch.equal(a, b)
    other.repack()                                      # back to its own (random) parameters
    assert not torch.equal(other.match(*args, return_scores=True)[4], ref[4])


def test_bench_two_ranks_one_gpu():
    """bench.py's N > 1 control flow on a single-GPU box, in the PLAIN form `python bench.py --gpus 2` (no launcher: bench.py
    starts its own two ranks under torch.distributed.run - VERDICT r4 #2), both on device 0 with gloo collectives
    (MDGAT_SHARE_DEVICE: RCCL refuses two ranks on one device).  Rank 1 never loads a checkpoint: it runs on the broadcast blob.
    One JSON line from rank 0, whole-job value = pairs of both ranks over the slowest rank's time."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MDGAT_SHARE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--batch', '8', '--no-breakdown',
           '--no-exact-mode']
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['scaling'] == 'weak'
    assert d['config']['pairs_per_gpu'] == 8 and '2-way' in d['config']['parallelism'] and d['config']['rccl_world_size'] == 2
    assert abs(d['value'] - 2 * 8 * 5 / (d['ms_per_step'] * 5e-3)) < 1e-6 * d['value']
    assert 'cpu_baseline' not in d                      # rank 0 at N = 1 only


def test_fuzz_checkpoint_short():
    """15 s of tools/fuzz_checkpoint.py - the forward-level robustness sweep standing in for the checkpoint the reference tree
    does not hold (pre-trained/best_model.pth is a missing blob): function-preserving rescalings of q against k, v against
    merge and of every hidden layer by 10^-3 ... 10^3 (the packer's gauge fixing must neutralise them), real gains on logits /
    messages / updates / scores, bin scores -5 ... 5, sparse FPFH rows, duplicated keypoints.  Every case ends EXACT (Z within
    max(1e-4, 8 x the error of a plain fp32 PyTorch run) of the fp64 oracle, k keys per dynamic row, near-tie selections only,
    matches = the extraction rules applied to the HIP path's own Z) or GUARDED (the f16 range guard raises) - never in
    silent garbage."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('fuzz_checkpoint', os.path.join(root, 'tools', 'fuzz_checkpoint.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    cases, guarded, exact, fails, worst = fz.run(15.0, seed=5, verbose=True)
    assert cases >= 30 and fails == 0 and exact >= cases // 2


def test_weights_beyond_the_f16_range_are_refused():
    """mdgat_load_weights checks the packed weights: a value beyond the f16 operand range (or non-finite) would become an
    infinite f16 head whose damage a ReLU can turn back into finite garbage - the load fails instead.  Rescalings the packer
    can undo (gauge fixing, pack.py) do not trip it."""
    L = 1
    cfg = synth.default_config(L=L, k=[], sinkhorn_iterations=5)
    d = synth.make_batch(1, 32, 32, device=DEV)
    args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
    sd = synth.make_state_dict(L=L, seed=2)
    ref_net = MDGAT(cfg).double()
    ref_net.load_state_dict(sd)
    ref = ref_net.eval().to(DEV).match(*args, return_scores=True)[4]
    # a hidden layer blown up by 10^6 against the convolution that reads it: the same network, and the packer says so
    sd2 = {k: v.clone() for k, v in sd.items()}
    for t in ('weight', 'bias'):
        sd2[f'gnn.layers.0.mlp.1.{t}'] = sd2[f'gnn.layers.0.mlp.1.{t}'] * 1e6
    sd2['gnn.layers.0.mlp.3.weight'] = sd2['gnn.layers.0.mlp.3.weight'] / 1e6
    net = MDGAT(cfg).double()
    net.load_state_dict(sd2)
    Z = net.eval().to(DEV).match(*args, return_scores=True)[4]
    assert (Z - ref).abs().max().item() < 1e-4
    # a weight that no rescaling can tame / a non-finite one: refused at load
    for bad in (1e9, float('nan')):
        sd3 = {k: v.clone() for k, v in sd.items()}
        sd3['final_proj.weight'][3, 5, 0] = bad
        net = MDGAT(cfg).double()
        net.load_state_dict(sd3)
        with pytest.raises(RuntimeError, match='f16 operand range'):
            net.eval().to(DEV).match(*args)


@pytest.mark.parametrize('B,n,m,L,k', [(1, 256, 256, 4, None), (1, 512, 512, 2, [128, None, 64, None]), (3, 100, 77, 2, [16, None, 8, None]),
                                       (2, 37, 53, 1, []), (1, 1, 9, 1, []), (5, 130, 75, 2, []), (16, 512, 512, 1, [])])
def test_layer_split_equals_layer_kernel(B, n, m, L, k):
    """csrc/layer_split.hip (32-keypoint workgroups, output channels split over the waves: launches of a few tiles) against
    csrc/layer.hip (a wave owns 16 keypoints): every stage tensor and every output of the forward bit for bit - the two kernels
    run the same split-f16 products in the same order (mdgat.py:227-232, 246-248, 274, 397)."""
    from mdgat_matcher_amd import _lib
    lib = _lib.load()
    cfg = synth.default_config(L=L, sinkhorn_iterations=10) if k is None else synth.default_config(L=L, k=k, sinkhorn_iterations=10)
    net = MDGAT(cfg)
    net.load_state_dict(synth.make_state_dict(L=L, seed=5))
    net = net.eval().to(DEV)
    data = synth.make_batch(B, n, m, first_pair=11, device=DEV)
    P = n + m

    def run():
        taps = {'x_enc': torch.empty(B, P, 128, device=DEV), 'x_layers': torch.empty(2 * L, B, P, 128, device=DEV),
                'mdesc': torch.empty(B, P, 128, device=DEV), 'scores': torch.empty(B, n, m, device=DEV)}
        out = net._run(data['keypoints0'], data['scores0'], data['descriptors0'], data['keypoints1'], data['scores1'],
                       data['descriptors1'], want_Z=True, taps=taps)
        torch.cuda.synchronize()
        return [t.clone() for t in out] + [taps[key].clone() for key in ('x_layers', 'mdesc', 'scores')]

    prev = lib.mdgat_set_layer_split_tiles(1 << 20)      # every launch on the channel-split kernel
    try:
        split = run()
        lib.mdgat_set_layer_split_tiles(0)               # never
        whole = run()
    finally:
        lib.mdgat_set_layer_split_tiles(prev)
    net.check()                                         # raises on a range violation
    for a, b in zip(split, whole):
        assert torch.equal(a, b)
    assert torch.isfinite(split[4]).all()
