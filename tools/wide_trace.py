"""Phase trace of attention_topk_wide_kernel (GPU box; tools/ab_build.sh attention widetrace -DWIDE_TRACE &&
   MDGAT_HIP_LIB=$PWD/ab/lib_widetrace.so python tools/wide_trace.py): s_memtime stamps of waves 0 and 7 of one workgroup."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops, _lib
B, n, k = 8, 2048, 128
qkv = torch.randn(B, 2 * n, 3, 4, 32, device='cuda:0') * 1.3
for _ in range(3):
    ops.attention(qkv, n, n, False, k)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 32)()
lib.mdgat_wide_trace_read.restype = ctypes.c_int
lib.mdgat_wide_trace_read(buf, 32)
t = np.array(buf[:], dtype=np.int64).reshape(4, 8)
names = ['Q K^T (K fragments from L2, 48 MFMAs)', 'row maximum (1 exchange)', 'threshold search', 'tie count (1 exchange + vote)', 'masked softmax, P V', 'write partials, barrier, merge', '-']
for w, wn in ((0, 'wave 0'), (1, 'wave 7')):
    print(wn, 'total', t[w, 7] - t[w, 0], 'ticks')
    idx = [0, 1, 2, 3, 4, 5, 7]
    for a, b, nm in zip(idx[:-1], idx[1:], names):
        print(f'   {nm:44s} {t[w, b] - t[w, a]:8d}')
