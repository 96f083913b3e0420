"""Dev only: staging vs compute time of the full attention kernel (library built with the trace patch)."""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops, _lib
g = torch.Generator('cuda').manual_seed(0)
qkv = torch.randn(64, 1024, 3, 4, 32, device='cuda', generator=g) * 1.3
for _ in range(3): ops.attention(qkv, 512, 512, False, 0)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * 128)()
lib.mdgat_debug_read(buf, 128)
t = np.array(buf[:]).reshape(32, 4)
t = t[t[:, 0] != 0]
print('workgroups traced', len(t))
print('staging cycles  mean %.0f min %.0f max %.0f' % ((t[:,1]-t[:,0]).mean(), (t[:,1]-t[:,0]).min(), (t[:,1]-t[:,0]).max()))
print('compute cycles  mean %.0f min %.0f max %.0f' % ((t[:,2]-t[:,1]).mean(), (t[:,2]-t[:,1]).min(), (t[:,2]-t[:,1]).max()))
