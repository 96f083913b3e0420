"""Forward-level robustness sweep standing in for the checkpoint the reference tree does not hold (GPU box):
    python tools/fuzz_checkpoint.py [seconds] [seed]
`pre-trained/best_model.pth` and the KITTI keypoint files are missing blobs (test.py:134-161, load_data.py:146-150): every
parity number of this repository is on synthetic weights whose activations are O(1-10) by construction (synth.py).  A trained
checkpoint need not be that tame, and the arithmetic here is split-f16 (|activation| < 65504, 22-bit operands).  This sweep
perturbs the synthetic network and its inputs the way a real one may differ:
  * gauge transformations (function-preserving rescalings of q against k, v against merge, hidden layers against the next
    convolution) by 10^-3 ... 10^3: the INTERNAL magnitudes the split-f16 operands see, with the network's output unchanged;
  * real gains 0.25 ... 4 on the logits / messages / residual updates of randomly chosen layers, 0.3 ... 2 on final_proj;
  * bin_score -5 ... 5;
  * FPFH rows with many EXACT zeros (real FPFH histograms are sparse), re-normalised like load_data.py:290-292;
  * duplicated keypoints (the same keypoint + descriptor several times in a frame: exactly tied logits and scores).
Every case must end in one of two ways - never in silent garbage:
  GUARDED  the f16 range guard fires (MDGAT.check() raises: activations beyond the f16 operand range or non-finite), or
  EXACT    everything is finite, Z is within max(1e-4, 8 x the error of a plain fp32 PyTorch run of the oracle) of the fp64
           oracle, both run with the HIP selections forced (the yardstick is what fp32 arithmetic makes of the same network:
           22-bit operands against 24), every dynamic row keeps exactly k
           keys, disagreeing selections are near-ties (within 1e-5 of the row's logit scale), and the matches are what
           mdgat.py:441-483 makes of the HIP path's own Z.
tests/test_gpu_forward.py::test_fuzz_checkpoint_short runs 15 s of it."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mdgat_matcher_amd import MDGAT, synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402
from parity_util import hip_forward_with_selection  # noqa: E402


def loguni(rs, lo, hi):
    return float(np.exp(rs.uniform(np.log(lo), np.log(hi))))


def perturb_state_dict(rs, sd, L):
    """Two kinds of change.  GAUGE transformations leave the network's function untouched and move its INTERNAL magnitudes -
    what the split-f16 operands see - by a factor g = 10^-3 ... 10^3: q x g with k / g; v x g with merge / g; the hidden layer
    of a propagation MLP x g (its BatchNorm's gamma and beta: ReLU is positively homogeneous) with mlp.3 / g; the same between
    the layers of the two encoders.  (A BatchNorm whose statistics absorb a gain of the convolution before it is NOT such a
    probe: eval-mode BatchNorm is folded into the convolution when the weights are packed, pack.py - the folded network is the
    same network.)  REAL changes: the gain of the logits (x 0.25 ... 4: rows from nearly uniform to nearly one-hot), of the
    messages and of the residual update (x 0.25 ... 4), of the final projection (scores x 0.1 ... 4), the bin score -5 ... 5."""
    notes = []

    def scale(key, g):
        sd[key] = sd[key] * g

    for i in range(2 * L):
        p = f'gnn.layers.{i}'
        if rs.uniform() < 0.5:
            g = loguni(rs, 1e-3, 1e3)
            for t in ('weight', 'bias'):
                scale(f'{p}.attn.proj.0.{t}', g)
                scale(f'{p}.attn.proj.1.{t}', 1.0 / g)
            notes.append(f'L{i}.q*{g:.2g}/k')
        if rs.uniform() < 0.5:
            g = loguni(rs, 1e-3, 1e3)
            for t in ('weight', 'bias'):
                scale(f'{p}.attn.proj.2.{t}', g)
            scale(f'{p}.attn.merge.weight', 1.0 / g)
            notes.append(f'L{i}.v*{g:.2g}/merge')
        if rs.uniform() < 0.5:
            g = loguni(rs, 1e-3, 1e3)
            for t in ('weight', 'bias'):
                scale(f'{p}.mlp.1.{t}', g)
            scale(f'{p}.mlp.3.weight', 1.0 / g)
            notes.append(f'L{i}.hid*{g:.2g}')
        if rs.uniform() < 0.3:
            g = loguni(rs, 0.25, 4.0)
            for t in ('weight', 'bias'):
                scale(f'{p}.attn.proj.0.{t}', g)
            notes.append(f'L{i}.logits x{g:.2g}')
        if rs.uniform() < 0.2:
            g = loguni(rs, 0.25, 4.0)
            scale(f'{p}.mlp.3.weight', g)
            notes.append(f'L{i}.update x{g:.2g}')
    for enc, bns in (('kenc.encoder', (1, 4, 7)), ('denc.encoder', (1, 4))):
        for bn in bns:
            if rs.uniform() < 0.4:
                g = loguni(rs, 1e-3, 1e3)
                for t in ('weight', 'bias'):
                    scale(f'{enc}.{bn}.{t}', g)
                scale(f'{enc}.{bn + 2}.weight', 1.0 / g)
                notes.append(f'{enc}.{bn}*{g:.2g}')
    if rs.uniform() < 0.5:
        g = loguni(rs, 0.3, 2.0)
        for t in ('weight', 'bias'):
            scale(f'final_proj.{t}', g)
        notes.append(f'final x{g:.2g}')
    sd['bin_score'] = torch.tensor(float(rs.uniform(-5.0, 5.0)), dtype=torch.float64)
    return notes


def perturb_inputs(rs, data):
    notes = []
    for side in ('0', '1'):
        d = data['descriptors' + side].clone()
        if rs.uniform() < 0.6:               # sparse FPFH histograms: exact zeros, rows re-normalised (load_data.py:290-292)
            keep = torch.from_numpy(rs.uniform(size=tuple(d.shape)) > rs.uniform(0.3, 0.85))
            keep[..., 0] = True               # (never an all-zero row: a histogram has mass somewhere)
            d = d * keep
            d = d / d.norm(dim=-1, keepdim=True)
            notes.append('sparse' + side)
        n = d.shape[1]
        if n >= 4 and rs.uniform() < 0.5:     # duplicated keypoints
            ndup = int(rs.randint(1, max(2, n // 8)))
            src = rs.randint(0, n, ndup)
            dst = rs.randint(0, n, ndup)
            for k in ('keypoints', 'scores'):
                t = data[k + side].clone()
                t[:, dst] = t[:, src]
                data[k + side] = t
            d[:, dst] = d[:, src]
            notes.append(f'dup{side}x{ndup}')
        data['descriptors' + side] = d
    return notes


def one_case(rs):
    B = int(rs.choice([1, 2, 4]))
    N, M = (int(x) for x in rs.choice([33, 64, 100, 128, 200, 256, 300, 512], 2))
    L = int(rs.choice([1, 2, 3, 4]))
    S = int(rs.choice([3, 10, 30]))
    kmax = min(N, M)
    k = [None if rs.uniform() < 0.4 else int(rs.randint(1, kmax + 1)) for _ in range(int(rs.choice([0, 2, 2 * L])))]
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, arithmetic='fp32')      # (the fp32-class path and its f16 range guard are the subject)
    sd = synth.make_state_dict(L=L, seed=int(rs.randint(100)))
    notes = perturb_state_dict(rs, sd, L)
    data = synth.make_batch(B, N, M, first_pair=int(rs.randint(1000)))
    notes += perturb_inputs(rs, data)
    tag = f'B={B} N={N} M={M} L={L} S={S} k={k} [{" ".join(notes)}]'
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.eval().to('cuda:0')
    dev = {kk: v.to('cuda:0') for kk, v in data.items()}
    try:
        (m0, m1, s0, s1, Zh), forced = hip_forward_with_selection(net, dev)
        net.check('cuda:0')
    except RuntimeError as e:
        if 'f16 operand range' in str(e):
            try:
                net.check('cuda:0')          # (clear the status for the next handle of this process)
            except RuntimeError:
                pass
            return 'guarded', tag, {}
        return 'fail', tag, {'exception': repr(e)}
    # fp64 oracle and a plain fp32 PyTorch run of it, both with the HIP selections forced: the yardstick is what fp32
    # arithmetic makes of THIS network (a wild one amplifies rounding: its scores can reach 10^5 and more)
    cap, cap32 = {}, {}
    O.mdgat_forward(sd, cfg, data, cap, forced_topk=forced)
    sd32 = {kk: (v.float() if v.is_floating_point() else v) for kk, v in sd.items()}
    d32 = {kk: (v.float() if v.is_floating_point() else v) for kk, v in data.items()}
    O.mdgat_forward(sd32, cfg, d32, cap32, forced_topk=forced)
    Z64 = cap['Z']
    finite64 = bool(torch.isfinite(Z64).all())
    errZ = float((Zh.cpu().double() - Z64).abs().max())
    err32 = float((cap32['Z'].double() - Z64).abs().max())
    finite = bool(torch.isfinite(Zh).all() and torch.isfinite(s0).all() and torch.isfinite(s1).all())
    bad = sum(r['bad_count'] for reps in cap.get('topk_report', {}).values() for r in reps)
    rel_gap = max([r['max_rel_gap'] for reps in cap.get('topk_report', {}).values() for r in reps] + [0.0])
    e0, e1, _, _ = O.extract_matches(Zh.cpu().double(), 'triplet_loss', False, cfg['match_threshold'])
    own = bool(torch.equal(m0.cpu(), e0) and torch.equal(m1.cpu(), e1))
    tol = max(1e-4, 8.0 * err32)
    ok = (finite or not finite64) and errZ <= tol and bad == 0 and rel_gap < 1e-5 and own
    info = {'errZ': errZ, 'err_fp32_torch': err32, 'tol': tol, 'zmax': float(Z64.abs().max()), 'finite': finite, 'bad_count': bad,
            'max_rel_gap': rel_gap, 'own': own}
    return ('exact' if ok else 'fail'), tag, info


def run(budget=60.0, seed=0, verbose=True):
    """Returns (cases, guarded, exact, failures, worst errZ / tolerance)."""
    rs = np.random.RandomState(seed)
    t0 = time.time()
    n = {'guarded': 0, 'exact': 0, 'fail': 0}
    worst = 0.0
    while time.time() - t0 < budget:
        kind, tag, info = one_case(rs)
        n[kind] += 1
        if kind == 'exact':
            worst = max(worst, info['errZ'] / info['tol'])
        if kind == 'fail' and verbose:
            print('FAIL', tag, info)
    if verbose:
        print(f'{sum(n.values())} cases in {time.time() - t0:.0f} s: {n["exact"]} exact (worst max|dZ| / tolerance {worst:.2f}), '
              f'{n["guarded"]} stopped by the f16 range guard, {n["fail"]} failures')
    return sum(n.values()), n['guarded'], n['exact'], n['fail'], worst


if __name__ == '__main__':
    torch.set_num_threads(synth.effective_cpu_count())
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
