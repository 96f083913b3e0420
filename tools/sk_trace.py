"""Dev only: s_memtime stamps inside the Sinkhorn scaling kernel (library built with the trace patch)."""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops, _lib
g = torch.Generator('cuda').manual_seed(0)
scores = torch.randn(64, 512, 512, device='cuda', generator=g) * 3
for _ in range(2): ops.sinkhorn(scores, 1.0, 100)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * 128)()
lib.mdgat_debug_read(buf, 128)
t = np.array(buf[:]).reshape(2, 8, 8)
for w in range(2):
    print('wg', 8 + w)
    for i in range(7):
        r = t[w, i]; nxt = t[w, i + 1, 0]
        print('  row %5d  col %5d  barrier %5d  exchange %5d  ->bvec+barrier %5d  tail %5d   total %5d' % (r[1]-r[0], r[2]-r[1], r[3]-r[2], r[4]-r[3], r[5]-r[4], nxt-r[5], nxt-r[0]))
