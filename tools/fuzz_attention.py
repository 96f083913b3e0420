"""Randomised attention / dynamic-attention cases through the per-op entry point against the fp64 oracle (GPU box):
    python tools/fuzz_attention.py [seconds] [seed]
Frames of 1 ... 1100 keypoints (ragged, N != M), self / cross, full or top-k with k anywhere in 1 ... keys, operand scales
0.05 ... 6 (logits up to a few hundred).  Message rows must match the oracle to |v| (3e-6 + 2e-7 max|logit|) + 4e-8, rows
whose k-th and (k+1)-th logit are closer than the arithmetic resolves excepted (and rare), every row must keep exactly k keys."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdgat_matcher_amd import ops, synth
from oracle import mdgat_oracle as O


def lib(msg):
    b, dh, h, n = msg.shape
    return msg.permute(0, 3, 2, 1).reshape(b, n, h * dh)


def run(budget=60.0, seed=0):
    rs = np.random.RandomState(seed)
    sizes = [1, 2, 17, 31, 64, 65, 100, 128, 200, 256, 257, 320, 512, 513, 600, 1024, 1100]
    t0, cases, fails, worst = time.time(), 0, 0, 0.0
    while time.time() - t0 < budget:
        B = int(rs.choice([1, 2, 3]))
        N, M = (int(x) for x in rs.choice(sizes, 2))
        cross = bool(rs.randint(2))
        sq, sk, sv = (float(x) for x in rs.choice([0.05, 0.5, 1.3, 3.0, 6.0], 3))
        nk_min = min(N, M)
        topk = 0 if rs.uniform() < 0.35 else int(rs.randint(1, nk_min + 1))
        qkv = rs.standard_normal((B, N + M, 3, 4, 32))
        qkv[:, :, 0] *= sq; qkv[:, :, 1] *= sk; qkv[:, :, 2] *= sv
        qkv = torch.from_numpy(qkv)
        tag = dict(B=B, N=N, M=M, cross=cross, topk=topk, sq=sq, sk=sk, sv=sv)
        cases += 1
        try:
            if topk:
                out, sel = ops.attention(qkv.cuda(), N, M, cross, topk=topk, return_selection=True)
                out2 = ops.attention(qkv.cuda(), N, M, cross, topk=topk)
            else:
                out, sel = ops.attention(qkv.cuda(), N, M, cross), None
                out2 = out
            out, out2 = out.cpu().double(), out2.cpu().double()
            ok = True
            for side, (lo, hi) in enumerate(((0, N), (N, N + M))):
                slo, shi = ((N, N + M) if side == 0 else (0, N)) if cross else (lo, hi)
                q = qkv[:, lo:hi, 0].permute(0, 3, 2, 1)
                k = qkv[:, slo:shi, 1].permute(0, 3, 2, 1)
                v = qkv[:, slo:shi, 2].permute(0, 3, 2, 1)
                logits = torch.einsum('bdhn,bdhm->bhnm', q, k) / 32 ** 0.5
                Lmax = float(logits.abs().max())
                n = hi - lo
                if topk:
                    kept = sel[side].cpu()
                    if not bool((kept.sum(-1) == topk).all()):
                        ok = False; print('FAIL count', tag, int(kept.sum(-1).min()), int(kept.sum(-1).max()))
                    rep = []
                    ref, _ = O.dynamic_attention(q, k, v, topk, forced=kept, report=rep)       # the kernel's selection forced ...
                    res = 5e-7 * max(Lmax, 0.2) + 2e-7 * float(k.abs().max())
                    if rep[0]['max_gap'] > 4 * res:                                               # ... and it is the fp64 one up to near-ties
                        ok = False; print('FAIL selection', tag, rep[0]['max_gap'], res)
                else:
                    ref, _ = O.attention(q, k, v)
                tol = sv * (3e-6 + 2e-7 * Lmax) + 4e-8
                for o, name in ((out, 'tap'), (out2, 'shipped')):
                    err = float((o[:, lo:hi] - lib(ref)).abs().max())
                    # (the shipped top-k kernels take the share of surplus tied keys out of the written row: allow its rounding)
                    lim = tol * (4 if (topk and name == 'shipped') else 1)
                    worst = max(worst, err / lim)
                    if not err <= lim:
                        ok = False; print('FAIL value', name, tag, 'side', side, err, lim)
        except Exception as e:                  # noqa: BLE001
            ok = False; print('EXCEPTION', tag, repr(e))
        fails += not ok
    print(f'{cases} cases in {time.time() - t0:.0f} s, {fails} failures, worst error / tolerance {worst:.2f}')
    return cases, fails


if __name__ == '__main__':
    torch.set_num_threads(synth.effective_cpu_count())
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
