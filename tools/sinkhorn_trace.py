"""Phase trace of the Sinkhorn cluster kernel (GPU box; library built with -DSK_TRACE:
   tools/ab_build.sh sinkhorn sktrace -DSK_TRACE && MDGAT_HIP_LIB=$PWD/ab/lib_sktrace.so python tools/sinkhorn_trace.py [B])
s_memtime stamps of waves 0 and 7 of workgroup 0, iterations 40-47: cycles spent between consecutive points."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
scores = (torch.randn(B, N, N, generator=g) * 3).to(dev)
for _ in range(3):
    Z = ops.sinkhorn(scores, 1.0, 100)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 192)()
lib.mdgat_sk_trace_read.restype = ctypes.c_int
lib.mdgat_sk_trace_read(buf, 192)
t = np.array(buf[:], dtype=np.int64).reshape(2, 8, 12)
names = ['row FMAs', 'row sums, a, dustbin sums', 'column FMAs', 'LDS write', 'barrier 1', 'merge 8 waves + store granule',
         'poll partners', 'b, dustbin column', 'barrier 2', 'read b']
for w, wn in ((0, 'wave 0'), (1, 'wave 7')):
    d = np.diff(t[w, :, :11], axis=1).astype(float)       # [8 iterations][10 phases]
    it = (t[w, 1:, 0] - t[w, :-1, 0]).astype(float)
    print(f'{wn}: {it.mean():.0f} ticks per iteration (min {it.min():.0f}, max {it.max():.0f})')
    for k, n in enumerate(names):
        print(f'   {n:34s} {d[:, k].mean():7.0f}  (min {d[:, k].min():.0f}, max {d[:, k].max():.0f})')
    print(f'   {"fold checks + loop":34s} {(t[w, 1:, 0] - t[w, :-1, 10]).mean():7.0f}')
