"""Dev only: per-stage s_memtime trace of the layer kernel (library built with -DLAYER_TRACE)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth, _lib
dev = torch.device('cuda', 0)
cfg = synth.default_config(L=9, sinkhorn_iterations=100)
net = MDGAT(cfg).eval(); net.load_state_dict(synth.make_state_dict(L=9, seed=0, dtype=torch.float32))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.make_batch(B, 512, 512, dtype=torch.float32, device=dev)
inp = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
with torch.no_grad():
    for _ in range(3): net._run(*inp)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * 1024)()
lib.mdgat_debug_read(buf, 1024)
for sel in range(4):
    ev = [(buf[sel * 256 + i] >> 48, buf[sel * 256 + i] & 0xffffffffffff) for i in range(256) if buf[sel * 256 + i]]
    if not ev: continue
    t0 = ev[0][1]
    print('== wg', 'first' if sel < 2 else 'last', 'wave', 0 if sel % 2 == 0 else 3, 'total', ev[-1][1] - t0)
    prev = t0; line = []
    for slot, t in ev:
        line.append('%d:%d' % (slot, t - prev)); prev = t
        if slot in (2, 12, 22, 24, 31, 33, 40): print('  ', ' '.join(line)); line = []
