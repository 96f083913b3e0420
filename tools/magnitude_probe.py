import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mdgat_matcher_amd import ops
from oracle import mdgat_oracle as O
DEV='cuda:0'
def lib(msg):
    b, dh, h, n = msg.shape
    return msg.permute(0, 3, 2, 1).reshape(b, n, h * dh)
for case,(sq,sk,sv) in {'small_q_large_k': (1e-3 * 32 ** 0.5, 30.0, 1.0), 'small_v': (1.3, 1.3, 1e-3), 'small_everything': (2e-2, 2e-2, 1e-3), 'large': (8.0, 8.0, 100.0), 'unit': (1.3,1.3,1.0), 'v0.05': (1.3,1.3,0.05)}.items():
    for N,topk in ((512,0),(512,128),(1024,0),(100,30)):
        rs = np.random.RandomState(N + topk + len(case))
        qkv = rs.standard_normal((2, 2 * N, 3, 4, 32)); qkv[:, :, 0] *= sq; qkv[:, :, 1] *= sk; qkv[:, :, 2] *= sv
        qkv = torch.from_numpy(qkv)
        out = ops.attention(qkv.to(DEV), N, N, False, topk=topk).cpu().double()
        q, kk, v = (qkv[:, :N, i].permute(0, 3, 2, 1) for i in range(3))
        logits = torch.einsum('bdhn,bdhm->bhnm', q, kk) / 32 ** 0.5
        if topk:
            ref,_ = O.dynamic_attention(q, kk, v, topk)
            top = logits.topk(topk + 1, dim=3).values
            ok = ((top[..., topk - 1] - top[..., topk]) >= 5e-6 * max(1.0, float(logits.abs().max()) / 10)).permute(0, 2, 1)
        else:
            ref,_ = O.attention(q, kk, v); ok = torch.ones(2, N, 4, dtype=torch.bool)
        err = (out[:, :N] - lib(ref)).abs().reshape(2, N, 4, 32).amax(3)
        print(f'{case:18s} N={N:5d} k={topk:4d} max|logit| {float(logits.abs().max()):8.2f} err {float(err[ok].max()):.2e} rel to sv {float(err[ok].max())/sv:.2e} ok-frac {float(ok.double().mean()):.4f}')
