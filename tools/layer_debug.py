"""Dev only: where does x after layer 0 differ from the golden (per 32-channel block / 32-keypoint group)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_forward as T
g = np.load('tests/golden/fwd_n64_L1_S1.npz')
net, data, (B, n, m, L) = T._build(g)
P = n + m
taps = {'x_enc': torch.empty(B, P, 128, device='cuda'), 'x_layers': torch.empty(2 * L, B, P, 128, device='cuda'),
        'mdesc': torch.empty(B, P, 128, device='cuda'), 'scores': torch.empty(B, n, m, device='cuda')}
net._run(data['keypoints0'], data['scores0'], data['descriptors0'], data['keypoints1'], data['scores1'], data['descriptors1'], want_Z=True, taps=taps)
torch.cuda.synchronize()
xl = taps['x_layers'].cpu().double().numpy()
for i in range(2 * L):
    ref = np.concatenate([T._to_lib(g[f'layer{i}_desc0']), T._to_lib(g[f'layer{i}_desc1'])], axis=1)
    err = np.abs(xl[i] - ref)          # [B, P, 128]
    print('layer', i, 'max', err.max(), 'B', B, 'P', P)
    e = err.reshape(B * P // 32, 32, 4, 32).max(axis=(1, 3))
    np.set_printoptions(linewidth=200, precision=2)
    print(e[:16])
md = taps['mdesc'].cpu().double().numpy()
ref = np.concatenate([T._to_lib(g['mdesc0']), T._to_lib(g['mdesc1'])], axis=1) if 'mdesc0' in g else None
if ref is not None: print('mdesc max', np.abs(md - ref).max())
enc = taps['x_enc'].cpu().double().numpy()
print('|x_layer0 - x_enc| max', np.abs(xl[0] - enc).max(), ' |ref0 - x_enc| max', np.abs(np.concatenate([T._to_lib(g['layer0_desc0']), T._to_lib(g['layer0_desc1'])], axis=1) - enc).max())
print('nan count', np.isnan(xl[0]).sum())
