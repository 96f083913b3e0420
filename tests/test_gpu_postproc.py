"""GPU parity of the steps either side of the matcher: against OUTPUTS OF THE REFERENCE
(tests/golden/aux_*.npz, produced by tools/make_goldens_aux.py from utils/utils_test.py and load_data.py) and, at more
shapes, against the oracle's restatement, which test_oracle_golden.py pins to the same fixtures.  fp64 on both sides."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mdgat_matcher_amd import ops  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402

DEV = 'cuda:0'


def _rigid(rs):
    a = rs.standard_normal(3); a /= np.linalg.norm(a)
    th = rs.uniform(0.05, 0.6)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rs.uniform(-3, 3, 3)
    return T


@pytest.mark.parametrize('N,M', [(256, 256), (512, 400), (40, 64), (2048, 2048)])
def test_pose_from_matches(N, M):
    rs = np.random.RandomState(N + M)
    B = 3
    k0 = (20 * rs.standard_normal((B, N, 3))).astype(np.float32)
    k1 = (20 * rs.standard_normal((B, M, 3))).astype(np.float32)
    m0 = -np.ones((B, N), dtype=np.int64)
    Tgt = np.stack([_rigid(rs) for _ in range(B)])
    for b in range(B):
        nm = min(N, M) // 2
        src, dst = rs.permutation(N)[:nm], rs.permutation(M)[:nm]
        Ti = np.linalg.inv(Tgt[b])
        k1[b, dst] = ((Ti[:3, :3] @ k0[b, src].T).T + Ti[:3, 3] + 0.05 * rs.standard_normal((nm, 3))).astype(np.float32)
        m0[b, src] = dst
        wrong = rs.permutation(nm)[:nm // 10]                       # some outlier matches
        m0[b, src[wrong]] = rs.randint(0, M, len(wrong))
    T, st = ops.pose_from_matches(torch.from_numpy(k0).to(DEV), torch.from_numpy(k1).to(DEV), torch.from_numpy(m0).to(DEV),
                                  T_gt=torch.from_numpy(Tgt).to(DEV))
    T, st = T.cpu().numpy(), st.cpu().numpy()
    for b in range(B):
        Tr, n, inl, ratio, te, re = O.pose_from_matches(k0[b], k1[b], m0[b], Tgt[b])
        assert np.abs(T[b] - Tr).max() < 1e-9
        assert st[b, 0] == n and st[b, 1] == inl and abs(st[b, 2] - ratio) < 1e-12
        assert abs(st[b, 3] - te) < 1e-9 and abs(st[b, 4] - re) < 1e-7


def test_pose_reflection_and_no_gt():
    # a mirrored correspondence set: solve_icp has no det(R) fix, R = U V^T is then a reflection - mirror it too
    rs = np.random.RandomState(0)
    k0 = (10 * rs.standard_normal((1, 50, 3))).astype(np.float32)
    k1 = k0.copy(); k1[..., 2] *= -1
    m0 = np.arange(50, dtype=np.int64)[None]
    T, st = ops.pose_from_matches(torch.from_numpy(k0).to(DEV), torch.from_numpy(k1).to(DEV), torch.from_numpy(m0).to(DEV))
    Tr = O.pose_from_matches(k0[0], k1[0], m0[0])[0]
    assert np.abs(T[0].cpu().numpy() - Tr).max() < 1e-9 and np.linalg.det(Tr[:3, :3]) < 0
    assert torch.isnan(st[0, 3]) and torch.isnan(st[0, 4])


@pytest.mark.parametrize('N,M,mutual', [(256, 256, False), (256, 256, True), (512, 300, False), (100, 130, True),
                                        (2048, 2048, False)])
def test_gt_matches(N, M, mutual):
    rs = np.random.RandomState(N * 3 + M + mutual)
    B = 2
    k0 = (20 * rs.standard_normal((B, N, 3))).astype(np.float32)
    k1 = (20 * rs.standard_normal((B, M, 3))).astype(np.float32)
    T0 = np.stack([_rigid(rs) for _ in range(B)])
    T1 = np.stack([_rigid(rs) for _ in range(B)])
    for b in range(B):                    # half of frame 1 re-observes frame-0 keypoints (world frame) with 0.2 m noise
        nm = min(N, M) // 2
        src, dst = rs.permutation(N)[:nm], rs.permutation(M)[:nm]
        w = (T0[b][:3, :3] @ k0[b, src].T).T + T0[b][:3, 3] + 0.2 * rs.standard_normal((nm, 3))
        Ti = np.linalg.inv(T1[b])
        k1[b, dst] = ((Ti[:3, :3] @ w.T).T + Ti[:3, 3]).astype(np.float32)
    g0, g1, rep = ops.gt_matches(torch.from_numpy(k0).to(DEV), torch.from_numpy(k1).to(DEV), torch.from_numpy(T0).to(DEV),
                                 torch.from_numpy(T1).to(DEV), threshold=0.5, mutual=mutual)
    for b in range(B):
        r0, r1, rr = O.gt_matches(k0[b], k1[b], T0[b], T1[b], 0.5, mutual)
        np.testing.assert_array_equal(g0[b].cpu().numpy(), r0)
        np.testing.assert_array_equal(g1[b].cpu().numpy(), r1)
        assert int(rep[b]) == rr and (r0 >= 0).sum() > 10
    # identity transforms
    g0, g1, rep = ops.gt_matches(torch.from_numpy(k0).to(DEV), torch.from_numpy(k1).to(DEV), threshold=5.0, mutual=mutual)
    r0, r1, rr = O.gt_matches(k0[0], k1[0], None, None, 5.0, mutual)
    np.testing.assert_array_equal(g0[0].cpu().numpy(), r0)



def _aux(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def test_pose_vs_reference_outputs(golden_dir):
    """solve_icp + calculate_error of the reference (utils_test.py:41-110) on its own outputs.  The rank-deficient sets
    ('three', 'planar': the sign of the null singular vector is the SVD routine's choice) are checked for what is
    defined there - T maps the matched points onto each other - and the others entry by entry."""
    g = _aux(golden_dir, 'aux_pose')
    for name in g['names']:
        mk0, mk1, T_gt, st = g[f'{name}_mkpts0'], g[f'{name}_mkpts1'], g[f'{name}_T_gt'], g[f'{name}_stats']
        n = len(mk0)
        T, out = ops.pose_from_matches(torch.from_numpy(mk0[None]).to(DEV), torch.from_numpy(mk1[None]).to(DEV),
                                       torch.arange(n, device=DEV)[None], T_gt=torch.from_numpy(T_gt[None]).to(DEV))
        T, out = T[0].cpu().numpy(), out[0].cpu().numpy()
        assert out[0] == st[0]
        if name in ('three', 'planar'):
            res = (T[:3, :3] @ mk1.T).T + T[:3, 3] - mk0
            ref = (g[f'{name}_T'][:3, :3] @ mk1.T).T + g[f'{name}_T'][:3, 3] - mk0
            assert np.abs(res).max() <= np.abs(ref).max() + 1e-5, name
            continue
        assert np.abs(T - g[f'{name}_T']).max() < 1e-9, name
        assert out[1] == st[1] and abs(out[2] - st[2]) < 1e-12, name
        assert abs(out[3] - st[3]) < 1e-9 and abs(out[4] - st[4]) < 1e-7, name


@pytest.mark.parametrize('mutual', [False, True])
def test_gt_matches_vs_reference_outputs(golden_dir, mutual):
    """The loader's ground-truth matcher (load_data.py:238-285) on the reference's own outputs."""
    g = _aux(golden_dir, 'aux_loader')
    for j in range(int(g['n_items'])):
        tag = f'item{j}_' + ('mutual_' if mutual else '')
        T0, T1, _ = O.frame_transforms(g[f'item{j}_pose0'], g[f'item{j}_pose1'], g['T_cam0_velo'])
        k0 = torch.from_numpy(g[tag + 'keypoints0'][None]).to(DEV)
        k1 = torch.from_numpy(g[tag + 'keypoints1'][None]).to(DEV)
        g0, g1, rep = ops.gt_matches(k0, k1, torch.from_numpy(T0[None]).to(DEV), torch.from_numpy(T1[None]).to(DEV),
                                     threshold=float(g['threshold']), mutual=mutual)
        np.testing.assert_array_equal(g0[0].cpu().numpy(), g[tag + 'gt_matches0'].astype(np.int64))
        np.testing.assert_array_equal(g1[0].cpu().numpy(), g[tag + 'gt_matches1'].astype(np.int64))
        assert int(rep[0]) == int(g[tag + 'rep'])
