"""Times the fp64 attention kernels alone (ops.attention_f64) at BASELINE configs[1]'s shape: us per launch and algorithmic TFLOP/s.
MDGAT_HIP_LIB selects an A/B build (tools/ab_build.sh f64 <variant> -D...).   python tools/f64_topk_time.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops  # noqa: E402

DEV = 'cuda:0'


def timed(fn, n=7, warm=3):
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
    return ts[len(ts) // 2] * 1e3


tag = os.path.basename(os.environ.get('MDGAT_HIP_LIB', 'shipped'))
for B in [int(x) for x in sys.argv[1:]] or [32, 64]:
    N = 512
    qkv = torch.randn(B, 2 * N, 3, 4, 32, dtype=torch.float64, device=DEV) * 1.3
    fl = B * 2 * 4 * (2 * 2 * N * N * 32)
    row = [f'{tag} B={B}:']
    for k in (0, 128, 64):
        t = timed(lambda: ops.attention_f64(qkv, N, N, False, topk=k))
        row.append(f'top-{k if k else "all"} {t:.1f} us {fl / t / 1e6:.1f} TF')
    print(' | '.join(row), flush=True)
