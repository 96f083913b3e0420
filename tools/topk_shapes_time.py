"""Dynamic attention alone (per-op entry point: includes the fp32 -> split-f16 pre-pass) at a list of shapes, GPU box:
    [MDGAT_HIP_LIB=...] python tools/topk_shapes_time.py "B:N:k,B:N:k,..."
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops
spec = sys.argv[1] if len(sys.argv) > 1 else '64:512:64,64:512:128,64:300:64,64:480:128,64:256:64,64:200:64,16:1024:128,8:2048:128,8:1500:64'
for item in spec.split(','):
    B, n, k = (int(v) for v in item.split(':'))
    qkv = torch.randn(B, 2 * n, 3, 4, 32, device='cuda:0') * 1.3
    for _ in range(3):
        ops.attention(qkv, n, n, False, k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.attention(qkv, n, n, False, k)
    b.record()
    torch.cuda.synchronize()
    print(f'B={B} N=M={n} k={k}: {a.elapsed_time(b) / 10 * 1e3:.1f} us')
