"""Pin the CPU oracle (oracle/mdgat_oracle.py) to outputs of the real reference.

The fixtures under tests/golden/ were produced by tools/make_goldens.py, which imports
/root/reference/models/mdgat.py (fp64, CPU).  The reference ships no tests of its own
(SURVEY.md section 4), so these are the only pins.  Everything here is fp64 vs fp64: the
agreement must be at round-off level, and integer outputs must be identical."""
import os

import numpy as np
import pytest
import torch

from mdgat_matcher_amd import synth
from oracle import mdgat_oracle as O

FWD = ['fwd_n64_L1_S1', 'fwd_n64_L4_S20', 'fwd_n64_L5_S20', 'fwd_n48m64_L4_S20']
TOL = 1e-9


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _setup(g):
    B, n, m, L, S, seed, first_pair = [int(x) for x in g['meta']]
    k = [None if x < 0 else int(x) for x in g['k']]
    bin_score = float(g['bin_score']) if 'bin_score' in g else 1.0
    sd = synth.make_state_dict(L=L, seed=seed, bin_score=bin_score)
    data = synth.make_batch(B, n, m, first_pair=first_pair)
    return sd, data, k, L, S, n, m


@pytest.mark.parametrize('name', FWD)
def test_forward_stage_tensors(golden_dir, name):
    g = _load(golden_dir, name)
    sd, data, k, L, S, n, m = _setup(g)
    cap = {}
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S)
    out = O.mdgat_forward(sd, cfg, data, cap)
    keys = ['enc0', 'enc1', 'mdesc0', 'mdesc1', 'scores', 'Z'] + \
           [f'layer{i}_desc{s}' for i in range(2 * L) for s in (0, 1)]
    for key in keys:
        err = np.abs(cap[key].numpy() - g[key]).max()
        assert err < TOL, (key, err)
    np.testing.assert_array_equal(out['matches0'].numpy(), g['default_matches0'])
    np.testing.assert_array_equal(out['matches1'].numpy(), g['default_matches1'])
    np.testing.assert_allclose(out['matching_scores0'].numpy(), g['default_mscores0'], atol=TOL)
    np.testing.assert_allclose(out['matching_scores1'].numpy(), g['default_mscores1'], atol=TOL)


@pytest.mark.parametrize('name', FWD)
def test_extraction_variants(golden_dir, name):
    g = _load(golden_dir, name)
    Z = torch.from_numpy(g['Z'])
    for tag, (loss_method, mutual) in {'default': ('triplet_loss', False), 'mutual': ('triplet_loss', True),
                                       'sg': ('superglue', False), 'sgmutual': ('superglue', True)}.items():
        if f'{tag}_matches0' not in g:
            continue
        m0, m1, s0, s1 = O.extract_matches(Z, loss_method, mutual, 0.2)
        np.testing.assert_array_equal(m0.numpy(), g[f'{tag}_matches0'], err_msg=tag)
        np.testing.assert_array_equal(m1.numpy(), g[f'{tag}_matches1'], err_msg=tag)
        np.testing.assert_allclose(s0.numpy(), g[f'{tag}_mscores0'], atol=TOL, err_msg=tag)
        np.testing.assert_allclose(s1.numpy(), g[f'{tag}_mscores1'], atol=TOL, err_msg=tag)


@pytest.mark.parametrize('name', ['cfg_n256_L4_S20', 'cfg_n512_L9_S100', 'cfg_n2048_L9_S200', 'cfg_n2048_L9_S200_b', 'cfg_n512_L9_S100_seed7',
                                  'cfg_n512_L9_S100_b40', 'cfg_n400_L9_S100_b24', 'cfg_n256_L4_S20_b64'])
def test_config_shapes(golden_dir, name):
    g = _load(golden_dir, name)
    sd, data, k, L, S, n, m = _setup(g)
    cap = {}
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S)
    out = O.mdgat_forward(sd, cfg, data, cap)
    Z = cap['Z']
    sub = int(g['sub']) if 'sub' in g else 8
    assert np.abs(Z.numpy()[:, ::sub, ::sub] - g['Z_sub']).max() < 1e-8
    assert np.abs(Z.numpy()[:, -1, :] - g['Z_lastrow']).max() < 1e-8
    assert np.abs(Z.numpy()[:, :, -1] - g['Z_lastcol']).max() < 1e-8
    assert np.abs(torch.logsumexp(Z, 2).numpy() - g['Z_row_lse']).max() < 1e-8
    assert np.abs(torch.logsumexp(Z, 1).numpy() - g['Z_col_lse']).max() < 1e-8
    np.testing.assert_array_equal(out['matches0'].numpy(), g['default_matches0'])
    np.testing.assert_array_equal(out['matches1'].numpy(), g['default_matches1'])
    np.testing.assert_allclose(out['matching_scores0'].numpy(), g['default_mscores0'], atol=1e-9)


VARIANT_MODES = {'default': ('triplet_loss', False), 'mutual': ('triplet_loss', True), 'sg': ('superglue', False), 'sgmutual': ('superglue', True)}


@pytest.mark.parametrize('name', ['var_n256_L4_S20', 'var_n512_L9_S100', 'var_n400m512_L9_S100'])
def test_config_shape_variants(golden_dir, name):
    """One pair at the BASELINE shapes (and a ragged 400 x 512 pair) through every extraction branch the reference can run."""
    g = _load(golden_dir, name)
    sd, data, k, L, S, n, m = _setup(g)
    for tag, (loss_method, mutual) in VARIANT_MODES.items():
        if f'{tag}_matches0' not in g:
            continue
        cap = {}
        cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, loss_method=loss_method, mutual_check=mutual)
        out = O.mdgat_forward(sd, cfg, data, cap)
        np.testing.assert_array_equal(out['matches0'].numpy(), g[f'{tag}_matches0'], err_msg=tag)
        np.testing.assert_array_equal(out['matches1'].numpy(), g[f'{tag}_matches1'], err_msg=tag)
        np.testing.assert_allclose(out['matching_scores0'].numpy(), g[f'{tag}_mscores0'], atol=1e-9, err_msg=tag)
        np.testing.assert_allclose(out['matching_scores1'].numpy(), g[f'{tag}_mscores1'], atol=1e-9, err_msg=tag)
        if tag == 'default':
            Z = cap['Z'].numpy()
            assert np.abs(Z[:, ::8, ::8] - g['Z_sub']).max() < 1e-8
            assert np.abs(Z[:, -1, :] - g['Z_lastrow']).max() < 1e-8 and np.abs(Z[:, :, -1] - g['Z_lastcol']).max() < 1e-8


def test_op_sinkhorn(golden_dir):
    g = _load(golden_dir, 'op_vectors')
    for tag in ('sk_7x5', 'sk_64x64', 'sk_48x64'):
        iters, alpha = g[tag + '_meta']
        Z = O.log_optimal_transport(torch.from_numpy(g[tag + '_scores']), float(alpha), int(iters))
        assert np.abs(Z.numpy() - g[tag + '_Z']).max() < 1e-10, tag
    iters, alpha = g['sk_512x512_meta']
    s = torch.from_numpy(np.random.RandomState(int(g['sk_512x512_seed'][0])).standard_normal((1, 512, 512)) * 3.0)
    Z = O.log_optimal_transport(s, float(alpha), int(iters)).numpy()
    assert np.abs(Z[:, ::8, ::8] - g['sk_512x512_Z_sub']).max() < 1e-9
    assert np.abs(Z[:, -1, :] - g['sk_512x512_Z_lastrow']).max() < 1e-9
    assert np.abs(Z[:, :, -1] - g['sk_512x512_Z_lastcol']).max() < 1e-9


def test_op_attention(golden_dir):
    g = _load(golden_dir, 'op_vectors')
    q, k, v = (torch.from_numpy(g[x]) for x in ('att_q', 'att_k', 'att_v'))
    full, _ = O.attention(q, k, v)
    assert np.abs(full.numpy() - g['att_full']).max() < 1e-12
    for kk in (1, 8, 56):
        dyn, prob = O.dynamic_attention(q, k, v, kk)
        assert np.abs(dyn.numpy() - g[f'att_dyn{kk}']).max() < 1e-12
        np.testing.assert_array_equal((prob > 0).sum(-1).numpy(), g[f'att_dyn{kk}_nnz'])
    # k == M is full attention
    assert np.abs(O.dynamic_attention(q, k, v, 56)[0].numpy() - g['att_full']).max() < 1e-12
    with pytest.raises(RuntimeError):
        O.dynamic_attention(q, k, v, 57)


def test_op_knn(golden_dir):
    g = _load(golden_dir, 'op_vectors')
    for C in (3, 128):
        x, s = torch.from_numpy(g[f'knn{C}_x']), torch.from_numpy(g[f'knn{C}_s'])
        np.testing.assert_array_equal(O.knn(x, s, 9).numpy(), g[f'knn{C}_idx'])
        np.testing.assert_array_equal(O.knn_adjacency(x, s, 9).numpy(), g[f'knn{C}_adj'])


def test_layer_schedule():
    # SURVEY.md section 3.2 [probe]: L=9, len(k)=8 -> layers 0-9 full; 10:128 11:None 12:128 13:None 14:64 15:None 16:64 17:None
    s = O.layer_topk_schedule(9, synth.DEFAULT_K)
    assert s == [None] * 10 + [128, None, 128, None, 64, None, 64, None]
    assert O.layer_topk_schedule(4, synth.DEFAULT_K) == synth.DEFAULT_K
    assert O.layer_topk_schedule(2, []) == [None] * 4


def test_edge_cases(golden_dir):
    g = _load(golden_dir, 'edge_cases')
    L = 1
    sd = synth.make_state_dict(L=L, seed=3)
    cfg = synth.default_config(L=L, k=[], sinkhorn_iterations=5)
    data = synth.make_batch(1, 8, 8)
    data['keypoints0'] = data['keypoints0'][:, :0]
    out = O.mdgat_forward(sd, cfg, data)
    assert out['skip_train'] is True and bool(g['empty_skip'])
    for a, b in (('matches0', 'empty_matches0'), ('matches1', 'empty_matches1'),
                 ('matching_scores0', 'empty_mscores0'), ('matching_scores1', 'empty_mscores1')):
        assert tuple(out[a].shape) == g[b].shape
        np.testing.assert_array_equal(out[a].numpy(), g[b])
    # k == M equals full attention end to end
    data = synth.make_batch(1, 32, 32)
    cap_full, cap_dyn = {}, {}
    O.mdgat_forward(sd, synth.default_config(L=L, k=[], sinkhorn_iterations=10), data, cap_full)
    O.mdgat_forward(sd, synth.default_config(L=L, k=[32, 32], sinkhorn_iterations=10), data, cap_dyn)
    assert np.abs(cap_full['Z'].numpy() - g['keqM_Z_full']).max() < TOL
    assert np.abs(cap_dyn['Z'].numpy() - g['keqM_Z_dyn']).max() < TOL
    assert np.abs(g['keqM_Z_full'] - g['keqM_Z_dyn']).max() < 1e-9
    # all-dustbin frame
    sd_bin = synth.make_state_dict(L=L, seed=3, bin_score=50.0)
    cap = {}
    out = O.mdgat_forward(sd_bin, synth.default_config(L=L, k=[], sinkhorn_iterations=10), data, cap)
    assert np.abs(cap['Z'].numpy() - g['alldust_Z']).max() < TOL
    np.testing.assert_array_equal(out['matches0'].numpy(), g['alldust_matches0'])
    np.testing.assert_array_equal(out['matches1'].numpy(), g['alldust_matches1'])
    assert (out['matches0'] == -1).all()
    np.testing.assert_array_equal(out['matching_scores0'].numpy(), g['alldust_mscores0'])


# ---------------------------------------------------------------- steps either side of the matcher
# fixtures: tools/make_goldens_aux.py (the reference's utils/utils_test.py and load_data.py, imported unmodified)
def test_pose_from_matches_vs_reference(golden_dir):
    g = _load(golden_dir, 'aux_pose')
    for name in g['names']:
        mk0, mk1, T_gt = g[f'{name}_mkpts0'], g[f'{name}_mkpts1'], g[f'{name}_T_gt']
        T = O.solve_icp(mk1, mk0)                                   # utils_test.py:73-110
        np.testing.assert_allclose(T, g[f'{name}_T'], atol=1e-12, err_msg=str(name))
        n = len(mk0)
        # calculate_error (utils_test.py:41-71) through the (kpts, matches) form test.py:213-216 feeds it
        Tr, cnt, inl, ratio, te, re = O.pose_from_matches(mk0, mk1, np.arange(n), T_gt, inlier_dist=1.0)
        st = g[f'{name}_stats']
        np.testing.assert_allclose(Tr, g[f'{name}_T'], atol=1e-12)
        assert cnt == st[0] and inl == st[1] and abs(ratio - st[2]) < 1e-15, name
        assert abs(te - st[3]) < 1e-9 * max(1.0, st[3]), name
        assert (np.isnan(re) and np.isnan(st[4])) or abs(re - st[4]) < 1e-9, name
    assert np.linalg.det(g['reflection_T'][:3, :3]) < 0             # the reference has no det(R) fix: mirrored


@pytest.mark.parametrize('mutual', [False, True])
def test_loader_vs_reference(golden_dir, mutual):
    g = _load(golden_dir, 'aux_loader')
    for j in range(int(g['n_items'])):
        tag = f'item{j}_' + ('mutual_' if mutual else '')
        for side in (0, 1):
            kp, score, desc = O.decode_frames(g[f'item{j}_rec{side}'][None])       # load_data.py:146-169, 290-295
            np.testing.assert_array_equal(kp[0].numpy(), g[tag + f'keypoints{side}'])
            np.testing.assert_array_equal(score[0].numpy(), g[tag + f'scores{side}'])
            np.testing.assert_array_equal(desc[0].numpy(), g[tag + f'descriptors{side}'])
        T0, T1, T_gt = O.frame_transforms(g[f'item{j}_pose0'], g[f'item{j}_pose1'], g['T_cam0_velo'])
        np.testing.assert_allclose(T_gt, g[tag + 'T_gt'], atol=1e-9)
        m0, m1, rep = O.gt_matches(g[tag + 'keypoints0'], g[tag + 'keypoints1'], T0, T1, float(g['threshold']), mutual)
        np.testing.assert_array_equal(m0, g[tag + 'gt_matches0'].astype(np.int64))         # load_data.py:257-285
        np.testing.assert_array_equal(m1, g[tag + 'gt_matches1'].astype(np.int64))
        assert rep == int(g[tag + 'rep']) and rep > 10


def test_fpfh_normalisation_order_is_numpys(golden_dir):
    """The exact mode's record entry (mdgat_forward_frames on an fp64 handle: csrc/f64.hip assemble_frames_f64_kernel) normalises
    the FPFH rows in float32 in numpy's own operation order, so that the forward sees the reference loader's inputs bit for bit.
    The order, restated step by step in the oracle, against numpy (load_data.py:290-292) and against the loader's outputs."""
    g = _load(golden_dir, 'aux_loader')
    for j in range(int(g['n_items'])):
        for side in (0, 1):
            rec = g[f'item{j}_rec{side}']
            out = O.fpfh_normalise_float32_steps(rec[:, 4:]).astype(np.float64)
            np.testing.assert_array_equal(out, g[f'item{j}_descriptors{side}'])
    rs = np.random.RandomState(5)
    d = (rs.standard_normal((50000, 33)) * 10.0 ** rs.uniform(-6, 6, (50000, 1))).astype(np.float32)
    d[rs.rand(50000, 33) < 0.3] = 0.0                                   # sparse histograms
    d[:7] = 0.0                                                         # empty rows: 0 * inf = NaN, as in the reference
    with np.errstate(divide='ignore', invalid='ignore'):
        ref = np.multiply(d, 1 / np.linalg.norm(d, axis=1).reshape(-1, 1))
    np.testing.assert_array_equal(O.fpfh_normalise_float32_steps(d), ref)


def test_committed_fixtures_are_what_the_generator_writes():
    """Fixture guard: tools/make_goldens.py --check regenerates the small fixtures from the imported reference
    into a scratch directory and compares keys and values with the committed files.  Only where /root/reference exists (the
    build container); the GPU box has no reference."""
    import subprocess
    import sys
    if not os.path.exists(os.path.join(os.environ.get('MDGAT_REFERENCE', '/root/reference'), 'models', 'mdgat.py')):
        pytest.skip('the reference is not present on this box')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'make_goldens.py'), '--check'], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert 'OK' in p.stdout.splitlines()[-1]
