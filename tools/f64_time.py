#!/usr/bin/env python3
"""Times the fp64 kernels of the reference-exact mode (csrc/f64.hip) on the GPU box: the sustained v_mfma_f64 rate, the
pointwise products and the attention kernels at the BASELINE shapes, and one exact-mode forward per configuration."""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, ops, synth  # noqa: E402

DEV = 'cuda:0'


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[len(ts) // 2]


def main():
    ms, fl, tk = ops.mfma_f64_probe(DEV, 4000)
    print(f'mfma_f64 probe: {fl / ms / 1e9:.1f} TFLOP/s sustained ({ms:.3f} ms, {tk} ticks per wave-loop)')
    for B, N in ((8, 512), (32, 512), (1, 2048)):
        R = B * 2 * N
        x = torch.randn(R, 128, dtype=torch.float64, device=DEV)
        for cout, K in ((384, 128), (256, 256), (128, 256)):
            a = torch.randn(R, K, dtype=torch.float64, device=DEV)
            w = torch.randn(cout, K, dtype=torch.float64, device=DEV)
            t = timed(lambda: ops.pointwise_f64(a, w))
            print(f'B={B} N={N} gemm {R}x{cout}x{K}: {t * 1e3:.1f} us  {2.0 * R * cout * K / t / 1e9:.1f} TFLOP/s')
        qkv = torch.randn(B, 2 * N, 3, 4, 32, dtype=torch.float64, device=DEV) * 1.3
        fl_att = B * 2 * 4 * (2 * 2 * N * N * 32)
        t = timed(lambda: ops.attention_f64(qkv, N, N, False))
        print(f'B={B} N={N} attention full: {t * 1e3:.1f} us  {fl_att / t / 1e9:.1f} TFLOP/s')
        for k in (128, 64):
            t = timed(lambda: ops.attention_f64(qkv, N, N, False, topk=k))
            print(f'B={B} N={N} attention top-{k}: {t * 1e3:.1f} us  {fl_att / t / 1e9:.1f} TFLOP/s (algorithmic), {1.5 * fl_att / t / 1e9:.1f} executed')
    for (B, N, L, S) in ((8, 512, 9, 100), (32, 512, 9, 100), (1, 256, 4, 20), (2, 2048, 9, 200)):
        for arith in ('fp64', 'fp32'):
            cfg = synth.default_config(L=L, sinkhorn_iterations=S, arithmetic=arith)
            net = MDGAT(cfg).double()
            net.load_state_dict(synth.make_state_dict(L=L, seed=0))
            net = net.eval().to(DEV)
            d = synth.make_batch(B, N, N, device=DEV)
            args = (d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1'])
            t = timed(lambda: net.match(*args), n=3, warm=1)
            print(f'forward {arith}: B={B} N={N} L={L} S={S}: {t:.3f} ms = {B / t * 1e3:.0f} pairs/s')
            if arith == 'fp64':
                net.profile(DEV, True)
                net.match(*args)
                torch.cuda.synchronize()
                prof = net.profile(DEV, False)
                print('   ', {k: (round(v[0], 3), v[1]) for k, v in prof.items() if v[1]})
            del net


if __name__ == '__main__':
    main()
