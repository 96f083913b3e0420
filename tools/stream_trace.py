"""Dev only: s_memtime trace of attention_stream_kernel (library built with -DSTREAM_TRACE, MDGAT_HIP_LIB set)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops, _lib
B, n = 64, 512
qkv = torch.randn(B, 2 * n, 3, 4, 32, device='cuda:0') * 1.3
for _ in range(3): ops.attention(qkv, n, n, False, 0)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * 1024)()
lib.mdgat_stream_debug_read(buf, 1024)
for sel in range(4):
    ev = [(buf[sel * 256 + i] >> 48, buf[sel * 256 + i] & 0xffffffffffff) for i in range(255) if buf[sel * 256 + i]]
    if not ev: continue
    t0 = ev[0][1]
    print('== wg', 777 if sel < 2 else 1500, 'wave', 0 if sel % 2 == 0 else 3, 'total', ev[-1][1] - t0)
    prev = t0; line = []
    for slot, t in ev:
        line.append('%d:%d' % (slot, t - prev)); prev = t
        if slot in (3, 16, 21) or (slot == 15 and False): print('  ', ' '.join(line)); line = []
