"""GPU parity of the steps either side of the matcher against the oracle's restatement of
utils/utils_test.py (pose from matches) and load_data.py (ground-truth matches).  fp64 on both sides."""
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
