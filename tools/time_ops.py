"""Ad-hoc kernel timing on the GPU box (HIP events on torch's stream): python tools/time_ops.py [B] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops  # noqa: E402


def t_ms(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    only = sys.argv[3] if len(sys.argv) > 3 else ''
    dev = 'cuda:0'
    g = torch.Generator(dev).manual_seed(0)
    qkv = torch.randn(B, 2 * n, 3, 4, 32, device=dev, generator=g) * 1.3
    fl = B * 1024.0 * n * n * 2
    for name, fn in (('attention_full', lambda: ops.attention(qkv, n, n, False, 0)),
                     ('attention_cross', lambda: ops.attention(qkv, n, n, True, 0)),
                     ('attention_top128', lambda: ops.attention(qkv, n, n, False, 128)),
                     ('attention_top64', lambda: ops.attention(qkv, n, n, False, 64))):
        if only and only not in name:
            continue
        ms = t_ms(fn)
        print(f'{name}: {ms:.4f} ms  ({fl / ms / 1e9:.1f} TFLOP/s fp32-equivalent, incl. the fp32->f16 split pre-pass)')
    if not only or 'post' in only:
        k0 = torch.randn(B, n, 3, device=dev, generator=g) * 20
        k1 = torch.randn(B, n, 3, device=dev, generator=g) * 20
        m0 = torch.randint(-1, n, (B, n), device=dev, generator=g)
        print(f'pose_from_matches: {t_ms(lambda: ops.pose_from_matches(k0, k1, m0)):.4f} ms')
        print(f'gt_matches: {t_ms(lambda: ops.gt_matches(k0, k1)):.4f} ms')
    if not only or 'knn' in only:
        # kNN graph helper (mdgat.py:8-32) at the frame size: C = 3 (coordinates) and C = 128 (feature space, matrix cores)
        for C, k in ((3, 9), (3, 64), (128, 9), (128, 64)):
            x = torch.randn(B, C, n, device=dev, generator=g) * (20 if C == 3 else 1)
            s = torch.randn(B, C, n, device=dev, generator=g) * (20 if C == 3 else 1)
            ms = t_ms(lambda: ops.knn(x, s, k))
            # algorithmic bytes: (N + M) C 4 + N k 8; the C = 128 path also writes and reads the N x M matrix once
            alg = B * ((2 * n) * C * 4 + n * k * 8)
            mat = B * 2 * n * n * 4 if C == 128 else 0
            print(f'knn C={C} k={k}: {ms:.4f} ms  (algorithmic {alg / ms / 1e6:.1f} GB/s; with the distance matrix {(alg + mat) / ms / 1e6:.1f} GB/s; '
                  f'{B * n / ms / 1e3:.2f} M queries/s)')
    if only and 'sinkhorn' not in only:
        return
    scores = torch.randn(B, n, n, device=dev, generator=g) * 3
    print(f'sinkhorn100: {t_ms(lambda: ops.sinkhorn(scores, 1.0, 100)):.4f} ms')


if __name__ == '__main__':
    main()
