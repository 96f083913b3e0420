#!/usr/bin/env python3
"""Generate tests/golden/aux_*.npz by running the REAL reference code either side of the matcher
in the build container: ``utils/utils_test.py`` (``solve_icp`` 73-110, ``calculate_error`` 41-71, ``calculate_error2``
27-39) and ``load_data.py`` (``SparseDataset.__init__`` 52-112 and ``__getitem__`` 114-321: record parse, ground-truth
matcher, FPFH normalisation, T_gt).

Only runs where ``/root/reference`` exists (never on the GPU box).  Both modules are imported UNMODIFIED; they fail to
import here only because of ``import open3d`` (utils_test.py:2, load_data.py:6), which none of the functions used below
touches (the two ``o3d`` uses in load_data.py sit behind ``vis_registered_keypoints = False``), so an empty placeholder
module named ``open3d`` is put in ``sys.modules`` first.  No reference file is edited or copied; the fixtures hold
inputs (seeded here) and the reference's outputs.

The KITTI keypoint files the loader reads (``<keypoints_path>/<seq>/<idx>.bin``, N x 37 float32) are missing blobs in
the reference tree, so synthetic frames in exactly that layout are written to a temporary directory; poses, calibration
and the pair list are the reference's real files (KITTI/poses/10.txt, KITTI/calib/sequences/10/calib.txt,
KITTI/preprocess-random-full/10/groundtruths.txt)."""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('MDGAT_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')


def import_reference_aux():
    sys.modules.setdefault('open3d', types.ModuleType('open3d'))     # placeholder: never called
    sys.path.insert(0, REF)
    import load_data as LD            # noqa: E402  the reference, unmodified
    import utils.utils_test as UT     # noqa: E402
    return LD, UT


def rigid(rs, max_angle=0.6, max_t=3.0):
    a = rs.standard_normal(3)
    a /= np.linalg.norm(a)
    th = rs.uniform(0.05, max_angle)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rs.uniform(-max_t, max_t, 3)
    return T


# ------------------------------------------------------------------------------------------ pose from matches
def pose_cases(UT):
    """Seeded correspondence sets -> the reference's solve_icp / calculate_error / calculate_error2 outputs."""
    rs = np.random.RandomState(20260930)
    out = {}
    names = []
    specs = [('clean64', 64, 0.0, 0), ('noisy200', 200, 0.05, 0), ('outliers300', 300, 0.05, 30), ('three', 3, 0.0, 0),
             ('reflection', 50, 0.0, 0), ('planar', 40, 0.0, 0), ('big2048', 2048, 0.1, 200)]
    for name, n, noise, n_out in specs:
        T_gt = rigid(rs)
        mk0 = 20.0 * rs.standard_normal((n, 3))
        if name == 'planar':
            mk0[:, 2] = 0.0          # rank-2 covariance: the third singular vector's sign is the SVD's choice
        Ti = np.linalg.inv(T_gt)
        mk1 = (Ti[:3, :3] @ mk0.T).T + Ti[:3, 3] + noise * rs.standard_normal((n, 3))
        if name == 'reflection':
            mk1 = mk0.copy()
            mk1[:, 2] *= -1.0        # mirrored set: R = U V has det -1 and solve_icp does not fix it (utils_test.py:100-101)
        if n_out:
            idx = rs.permutation(n)[:n_out]
            mk1[idx] = 20.0 * rs.standard_normal((n_out, 3))
        # test.py:213-216 hands float64 numpy rows of the (double) keypoint tensors to calculate_error
        mk0 = mk0.astype(np.float32).astype(np.float64)
        mk1 = mk1.astype(np.float32).astype(np.float64)
        T_icp = UT.solve_icp(mk1, mk0)
        pred = {'T_gt': torch.tensor(T_gt[None], dtype=torch.double)}
        with np.errstate(invalid='ignore'):
            T, inlier, ratio, te, re = UT.calculate_error(mk0, mk1, pred, 0)
            T2, rte, rre = UT.calculate_error2(mk0, mk1, 0, torch.tensor(T_gt, dtype=torch.double))
        assert np.array_equal(T.numpy(), T_icp) and np.array_equal(T2.numpy(), T_icp)
        assert (np.isnan(re) and np.isnan(rre)) or (te == rte and re == rre)
        out[f'{name}_mkpts0'], out[f'{name}_mkpts1'], out[f'{name}_T_gt'] = mk0, mk1, T_gt
        out[f'{name}_T'] = T_icp
        out[f'{name}_stats'] = np.array([n, int(inlier), ratio, te, re], dtype=np.float64)
        names.append(name)
        print(f'pose {name}: n={n} det(R)={np.linalg.det(T_icp[:3, :3]):+.3f} inliers={int(inlier)} te={te:.3e} re={re:.3e}')
    out['names'] = np.array(names)
    return out


# ------------------------------------------------------------------------------------------ loader
def loader_cases(LD):
    """Synthetic keypoint frames in the KITTI layout -> SparseDataset.__getitem__ outputs (both mutual_check settings)."""
    rs = np.random.RandomState(7)
    seq = 10
    pairs = LD.load_kitti_gt_txt(os.path.join(REF, 'KITTI', 'preprocess-random-full'), seq)[:3]
    # poses / calibration exactly as SparseDataset.__init__ parses them (needed to place synthetic keypoints so that the
    # ground-truth matcher has something to find); the loader below re-reads them itself
    probe_opt = None
    out = {}
    tmp = tempfile.mkdtemp(prefix='mdgat_kpts_')
    os.makedirs(os.path.join(tmp, '%02d' % seq))

    def make_opt(mutual):
        return argparse.Namespace(train_path=os.path.join(REF, 'KITTI'), keypoints='USIP', keypoints_path=tmp,
                                  descriptor='FPFH', max_keypoints=512, threshold=0.5, ensure_kpts_num=False,
                                  mutual_check=mutual, memory_is_enough=False,
                                  txt_path=os.path.join(REF, 'KITTI', 'preprocess-random-full'))

    # first pass without files: only to get the parsed pose / calib of the real KITTI text files
    ds0 = LD.SparseDataset(make_opt(False), 'test')
    assert ds0.dataset[:3] == pairs
    sizes = [(200, 168), (256, 256), (97, 130)]
    records = {}
    for item, (n0, n1) in zip(pairs, sizes):
        i0, i1 = item['anc_idx'], item['pos_idx']
        Tcv = ds0.calib['%02d' % seq]
        W0 = ds0.pose['%02d' % seq][i0] @ Tcv       # sensor -> world of each frame (load_data.py:238-239)
        W1 = ds0.pose['%02d' % seq][i1] @ Tcv
        # frame-0 keypoints around the sensor; a subset of them re-observed in frame 1 with noise straddling the 0.5 m
        # threshold, the rest of frame 1 unrelated
        kp0 = rs.uniform(-30, 30, (n0, 3)) * np.array([1.0, 1.0, 0.1])
        n_common = min(n0, n1) // 2
        src = rs.permutation(n0)[:n_common]
        dst = rs.permutation(n1)[:n_common]
        kp1 = rs.uniform(-30, 30, (n1, 3)) * np.array([1.0, 1.0, 0.1])
        w = (W0[:3, :3] @ kp0[src].T).T + W0[:3, 3] + rs.uniform(0.0, 0.45, (n_common, 1)) * rs.standard_normal((n_common, 3))
        W1i = np.linalg.inv(W1)
        kp1[dst] = (W1i[:3, :3] @ w.T).T + W1i[:3, 3]
        for idx, kp in ((i0, kp0), (i1, kp1)):
            if idx in records:
                continue
            rec = np.zeros((len(kp), 37), dtype=np.float32)
            rec[:, :3] = kp
            rec[:, 3] = rs.uniform(0.0, 1.0, len(kp))                 # saliency
            rec[:, 4:] = rs.uniform(0.0, 200.0, (len(kp), 33))        # un-normalised FPFH histogram
            records[idx] = rec
            rec.tofile(os.path.join(tmp, '%02d' % seq, '%06d.bin' % idx))
    for mutual in (False, True):
        ds = LD.SparseDataset(make_opt(mutual), 'test')
        for j, item in enumerate(pairs):
            d = ds[j]
            tag = f'item{j}_' + ('mutual_' if mutual else '')
            assert d['sequence'] == '%02d' % seq and d['idx0'] == item['anc_idx']
            for key in ('keypoints0', 'keypoints1', 'descriptors0', 'descriptors1', 'scores0', 'scores1', 'T_gt'):
                assert d[key].dtype == torch.double
                out[tag + key] = d[key].numpy()
            out[tag + 'gt_matches0'] = np.asarray(d['gt_matches0'])
            out[tag + 'gt_matches1'] = np.asarray(d['gt_matches1'])
            out[tag + 'rep'] = np.array(d['rep'])
            print(f'loader item {j} mutual={mutual}: N={len(d["keypoints0"])} M={len(d["keypoints1"])} rep={d["rep"]} '
                  f'gt0>=0: {(np.asarray(d["gt_matches0"]) >= 0).sum()} gt1>=0: {(np.asarray(d["gt_matches1"]) >= 0).sum()}')
    for j, item in enumerate(pairs):
        out[f'item{j}_rec0'] = records[item['anc_idx']]
        out[f'item{j}_rec1'] = records[item['pos_idx']]
        out[f'item{j}_pose0'] = ds0.pose['%02d' % seq][item['anc_idx']]
        out[f'item{j}_pose1'] = ds0.pose['%02d' % seq][item['pos_idx']]
    out['T_cam0_velo'] = ds0.calib['%02d' % seq]
    out['n_items'] = np.array(len(pairs))
    out['threshold'] = np.array(0.5)
    return out


def main():
    LD, UT = import_reference_aux()
    np.savez_compressed(os.path.join(OUT, 'aux_pose.npz'), **pose_cases(UT))
    np.savez_compressed(os.path.join(OUT, 'aux_loader.npz'), **loader_cases(LD))
    for f in ('aux_pose.npz', 'aux_loader.npz'):
        print(f, os.path.getsize(os.path.join(OUT, f)), 'bytes')


if __name__ == '__main__':
    main()
