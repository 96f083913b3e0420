"""Per-iteration time of the Sinkhorn cluster kernel with and without the inter-workgroup exchange (GPU box).
N=128, M=512 is one workgroup per pair with the same 128 x 512 register tile as N=M=512 but no partners."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops

def t_ms(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

g = torch.Generator('cuda').manual_seed(0)
for B, n, m in ((64, 512, 512), (256, 128, 512), (128, 256, 512), (64, 512, 128)):
    s = torch.randn(B, n, m, device='cuda', generator=g) * 3
    ts = {it: t_ms(lambda: ops.sinkhorn(s, 1.0, it)) for it in (0, 100, 200)}
    print(f'B={B} N={n} M={m}: {ts[0]*1e3:.1f} / {ts[100]*1e3:.1f} / {ts[200]*1e3:.1f} us at 0/100/200 iterations -> {(ts[200]-ts[100])*10:.2f} us per iteration '
          f'(includes the Z write of ops.sinkhorn)')
