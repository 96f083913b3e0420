"""Coarse phases of the Sinkhorn cluster kernel inside a forward (GPU box; library built with -DSK_TRACE:
   tools/ab_build.sh sinkhorn sktrace -DSK_TRACE && MDGAT_HIP_LIB=$PWD/ab/lib_sktrace.so python tools/sinkhorn_phases.py B n L S)
s_memtime stamps of waves 0 and 7 of workgroup 0 (core clock ticks, ~2.5 per ns)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, '.')
from mdgat_matcher_amd import MDGAT, synth, _lib
dev = torch.device('cuda', 0)
B, n, L, S = [int(x) for x in sys.argv[1:5]]
net = MDGAT(synth.default_config(L=L, sinkhorn_iterations=S)).eval()
net.load_state_dict(synth.make_state_dict(L=L, seed=0, dtype=torch.float32))
d = synth.make_batch(B, n, n, dtype=torch.float32, device=dev)
inp = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
with torch.no_grad():
    for _ in range(5): net._run(*inp)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 16)()
lib.mdgat_sk_phase_read.restype = ctypes.c_int
lib.mdgat_sk_phase_read(buf, 16)
t = np.array(buf[:], dtype=np.int64).reshape(2, 8)
names = ['load scores', 'absorb row maxima (exp2), range check, setup', 'barrier + XCD handshake', f'{S} iterations', 'Z rows + row arg-maxes', 'column merge + stores']
for w, wn in ((0, 'wave 0'), (1, 'wave 7')):
    dd = np.diff(t[w, :7]).astype(float)
    print(wn, 'total ticks', t[w, 6] - t[w, 0], '(100 MHz ticks -> x 10 ns)' )
    for k, nme in enumerate(names): print(f'   {nme:34s} {dd[k]:9.0f}')
