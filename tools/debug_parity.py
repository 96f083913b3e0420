#!/usr/bin/env python3
"""Per-stage error report: HIP path (taps) vs the CPU oracle on the same seeded inputs (GPU box tool)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, ops, synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=256)
ap.add_argument('--m', type=int, default=None)
ap.add_argument('--L', type=int, default=4)
ap.add_argument('--S', type=int, default=20)
ap.add_argument('--B', type=int, default=1)
a = ap.parse_args()
n, m, L, S, B = a.n, a.m or a.n, a.L, a.S, a.B
dev = 'cuda:0'
torch.set_num_threads(synth.effective_cpu_count())
cfg = synth.default_config(L=L, sinkhorn_iterations=S)
sd = synth.make_state_dict(L=L, seed=0)
data = synth.make_batch(B, n, m)
cap = {}
ref = O.mdgat_forward(sd, cfg, data, cap)
net = MDGAT(cfg)
net.load_state_dict(sd)
net = net.double().eval().to(dev)
d = {k: v.to(dev) for k, v in data.items()}
P = n + m
taps = {'x_enc': torch.empty(B, P, 128, device=dev), 'x_layers': torch.empty(2 * L, B, P, 128, device=dev),
        'mdesc': torch.empty(B, P, 128, device=dev), 'scores': torch.empty(B, n, m, device=dev)}
m0, m1, s0, s1, Z = net._run(d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'],
                             d['descriptors1'], want_Z=True, taps=taps)
torch.cuda.synchronize()


def lib(x):
    return x.permute(0, 2, 1)


def rep(name, got, want):
    e = (got.cpu().double() - want).abs()
    print(f'{name:12s} max err {e.max().item():.3e}  mean {e.mean().item():.3e}  |ref|max {want.abs().max().item():.3e}')


rep('enc0', taps['x_enc'][:, :n], lib(cap['enc0']))
for i in range(2 * L):
    rep(f'layer{i}', taps['x_layers'][i], torch.cat([lib(cap[f'layer{i}_desc0']), lib(cap[f'layer{i}_desc1'])], 1))
rep('mdesc', taps['mdesc'], torch.cat([lib(cap['mdesc0']), lib(cap['mdesc1'])], 1))
rep('scores', taps['scores'], cap['scores'])
rep('Z', Z, cap['Z'])
e = (Z.cpu().double() - cap['Z']).abs()
idx = np.unravel_index(int(e.argmax()), e.shape)
print('worst Z at', idx, 'ref', cap['Z'][idx].item(), 'got', Z.cpu()[idx].item())
print('col err profile (max over rows) top5:', torch.topk(e.max(1).values.flatten(), 5))
print('row err profile (max over cols) top5:', torch.topk(e.max(2).values.flatten(), 5))
# isolate Sinkhorn: oracle OT on the HIP scores, HIP OT on oracle scores
Zo = O.log_optimal_transport(taps['scores'].cpu().double(), sd['bin_score'], S)
rep('Z|hipscores', Z, Zo)
Zh = ops.sinkhorn(cap['scores'].to(dev), float(sd['bin_score']), S)
rep('Zhip|refsc', Zh, cap['Z'])
print('matches0 equal', torch.equal(m0.cpu(), ref['matches0']), 'matches1 equal', torch.equal(m1.cpu(), ref['matches1']),
      'mismatches', int((m0.cpu() != ref['matches0']).sum()), int((m1.cpu() != ref['matches1']).sum()))
