"""Build-container experiment (CPU): how much accuracy do split-half-precision MFMA schemes cost?

Runs the oracle's forward in fp32 with every contraction replaced by an emulation of a given operand
format and compares Z with the fp64 oracle.  Schemes:
  f32      plain fp32 contraction
  f16x3    a = hi + lo/2048 with hi, lo fp16; a.b ~ hi.hi + (hi.lo + lo.hi)/2048   (3 fp16 MFMAs)
  bf16x3   a = hi + lo with hi, lo bf16;      a.b ~ hi.hi + hi.lo + lo.hi          (3 bf16 MFMAs)
  f16      single fp16 operands
  f16x3u   like f16x3 with UNSCALED residual planes lo = f16(a - hi): one accumulator for all three MFMAs, no combine
           (UNSCALED=1 runs it for the q.k and p.v classes)
Per-operand ablation of f16x3 (which of the cross terms does a product need for the 1e-4 bar?): a.b = hi.hi + a_hi.b_lo + a_lo.b_hi
  f16x2a   a single f16 (drops a_lo.b_hi), b split     - for P.V: probabilities as ONE f16, V split
  f16x2ac  the same with the row sum taken over the ROUNDED a (P.V only: softmax weights still sum to one)
  f16x2b   a split, b single f16 (drops a_hi.b_lo)
(a = weights / q / probabilities, b = activations / k / v in the three product classes lin / qk / pv)
Products of two half-precision values are exact in fp32, so only the accumulation order differs from
the hardware.  Dynamic layers run with the fp64 run's top-k selections forced (arithmetic error only, no near-tie
flips).  Usage: [ABLATE=1] python tools/precision_probe.py [N] [L] [S]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402


def split_f16(a):
    hi = a.half().float()
    lo = ((a - hi) * 2048.0).half().float()
    return hi, lo


def split_bf16(a):
    hi = a.bfloat16().float()
    lo = (a - hi).bfloat16().float()
    return hi, lo


def make_mm(scheme):
    def mm(a, b):   # a [..., m, k] @ b [..., k, n]
        if scheme == 'f32':
            return a @ b
        if scheme == 'f16':
            return a.half().float() @ b.half().float()
        if scheme == 'f16x3':
            ah, al = split_f16(a)
            bh, bl = split_f16(b)
            return ah @ bh + (ah @ bl + al @ bh) * (1.0 / 2048.0)
        if scheme == 'f16x3u':       # unscaled residual planes: lo = f16(a - hi) (f16 denormals kept: the MFMA honours them)
            ah, bh = a.half().float(), b.half().float()
            al, bl = (a - ah).half().float(), (b - bh).half().float()
            return ah @ bh + ah @ bl + al @ bh
        if scheme in ('f16x2a', 'f16x2ac'):
            ah = a.half().float()
            bh, bl = split_f16(b)
            out = ah @ bh + (ah @ bl) * (1.0 / 2048.0)
            if scheme == 'f16x2ac':      # rows of a are softmax weights (sum 1 before rounding): renormalise by the rounded sum
                out = out / ah.sum(-1, keepdim=True)
            return out
        if scheme == 'f16x2b':
            ah, al = split_f16(a)
            bh = b.half().float()
            return ah @ bh + (al @ bh) * (1.0 / 2048.0)
        if scheme == 'bf16x3':
            ah, al = split_bf16(a)
            bh, bl = split_bf16(b)
            return ah @ bh + (ah @ bl + al @ bh)
        raise ValueError(scheme)
    return mm


def run(scheme_lin, scheme_qk, scheme_pv, sd32, cfg, data, forced=None):
    mm_lin, mm_qk, mm_pv = make_mm(scheme_lin), make_mm(scheme_qk), make_mm(scheme_pv)
    saved = (O._pointwise, O.attention, O.dynamic_attention, torch.einsum)

    def pointwise(w, b, x):
        return mm_lin(w[:, :, 0], x) + b[None, :, None]

    def logits_of(q, k):
        # [B, dh, H, N] x [B, dh, H, M] -> [B, H, N, M]
        return mm_qk(q.permute(0, 2, 3, 1), k.permute(0, 2, 1, 3)) / q.shape[1] ** 0.5

    def pv(prob, v):
        # [B, H, N, M] x [B, dh, H, M] -> [B, dh, H, N]
        return mm_pv(prob, v.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

    def attention(q, k, v):
        prob = torch.softmax(logits_of(q, k), dim=-1)
        return pv(prob, v), prob

    def dynamic_attention(q, k, v, topk, forced=None, report=None):
        logits = logits_of(q, k)
        if forced is not None:       # the fp64 run's selection: arithmetic error only, no near-tie flips
            prob = torch.softmax(logits.masked_fill(~forced, float('-inf')), dim=-1)
        else:
            top = logits.topk(topk, dim=3)
            prob = torch.zeros_like(logits)
            prob.scatter_(3, top.indices, torch.softmax(top.values, dim=-1))
        return pv(prob, v), prob

    def einsum(eq, *ops):
        if eq == 'bdn,bdm->bnm':
            return mm_lin(ops[0].transpose(1, 2), ops[1])
        return saved[3](eq, *ops)

    O._pointwise, O.attention, O.dynamic_attention, torch.einsum = pointwise, attention, dynamic_attention, einsum
    try:
        cap = {}
        out = O.mdgat_forward(sd32, cfg, data, cap, forced_topk=forced)
    finally:
        O._pointwise, O.attention, O.dynamic_attention, torch.einsum = saved
    return out, cap


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    torch.set_num_threads(8)
    for k in (None, []):
        cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S)
        sd = synth.make_state_dict(L=L, seed=0)
        sd32 = {kk: (v.float() if v.dtype == torch.float64 else v) for kk, v in sd.items()}
        data = synth.make_batch(1, n, n)
        d32 = {kk: (v.float() if v.dtype == torch.float64 else v) for kk, v in data.items()}
        with torch.no_grad():
            cap64 = {}
            ref = O.mdgat_forward(sd, cfg, data, cap64, forced_topk={})
            forced = {i: (r[0]['own'], r[1]['own']) for i, r in cap64.get('topk_report', {}).items()} or None
            print(f'--- N={n} L={L} S={S} k={"default" if k is None else "none"}')
            combos = (('f32', 'f32', 'f32'), ('f16x3', 'f16x3', 'f16x3'), ('bf16x3', 'bf16x3', 'bf16x3'),
                      ('f16x3', 'f16x3', 'f16'), ('f16x3', 'f32', 'f32'), ('f32', 'f16x3', 'f32'),
                      ('f32', 'f32', 'f16x3'), ('f32', 'f32', 'f16'))
            if os.environ.get('UNSCALED'):
                combos = (('f16x3', 'f16x3', 'f16x3'), ('f16x3', 'f16x3u', 'f16x3'), ('f16x3', 'f16x3u', 'f16x3u'), ('f16x3u', 'f16x3u', 'f16x3u'))
            if os.environ.get('ABLATE'):
                combos = (('f16x3', 'f16x3', 'f16x3'),
                          ('f16x3', 'f16x3', 'f16x2a'), ('f16x3', 'f16x3', 'f16x2ac'), ('f16x3', 'f16x3', 'f16x2b'),
                          ('f16x3', 'f16x2a', 'f16x3'), ('f16x3', 'f16x2b', 'f16x3'),
                          ('f16x2a', 'f16x3', 'f16x3'), ('f16x2b', 'f16x3', 'f16x3'))
            for lin, qk, pvs in combos:
                out, cap = run(lin, qk, pvs, sd32, cfg, d32, forced)
                dz = (cap['Z'].double() - cap64['Z']).abs()
                ds = (cap['scores'].double() - cap64['scores']).abs().max().item()
                mm0 = (out['matches0'] != ref['matches0']).sum().item() + (out['matches1'] != ref['matches1']).sum().item()
                print(f'lin={lin:7s} qk={qk:7s} pv={pvs:7s} max|dscores|={ds:.2e} max|dZ|={dz.max().item():.2e} '
                      f'median|dZ|={dz.median().item():.2e} frac>1e-4={(dz > 1e-4).double().mean().item():.2e} match-mismatch={mm0}')


if __name__ == '__main__':
    main()
