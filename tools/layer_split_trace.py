"""Phase trace of the channel-split layer kernel at one pair (GPU box; library built with -DSPLIT_TRACE:
   tools/ab_build.sh layer_split sptrace -DSPLIT_TRACE && MDGAT_HIP_LIB=$PWD/ab/lib_sptrace.so python tools/layer_split_trace.py [n])
s_memtime stamps (core-clock ticks) of waves 0 and 7 of workgroup 0 of the last <1, 1> launch."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from mdgat_matcher_amd import MDGAT, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda', 0)
net = MDGAT(synth.default_config(L=4, sinkhorn_iterations=20)).eval()
net.load_state_dict(synth.make_state_dict(L=4, seed=0, dtype=torch.float32))
d = synth.make_batch(1, n, n, dtype=torch.float32, device=dev)
inp = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
with torch.no_grad():
    for _ in range(5): net._run(*inp)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 32)()
lib.mdgat_split_trace_read.restype = ctypes.c_int
lib.mdgat_split_trace_read(buf, 32)
t = np.array(buf[:], dtype=np.int64).reshape(2, 16)
names = ['loads issued (rows, W1, biases)', 'rows landed -> fp32 tiles', 'barrier', 'fragments built', 'phase 1 products', 'phase 1 epilogue',
         'barrier', 'phase 2 products', 'phase 2 epilogue', 'barrier', 'x rows out', 'phase 3 products', 'q / k / v epilogue + stores']
for w, wn in ((0, 'wave 0'), (1, 'wave 7')):
    dd = np.diff(t[w, :14]).astype(float)
    print(wn, 'total', t[w, 13] - t[w, 0], 'ticks')
    for k, nme in enumerate(names): print(f'   {nme:34s} {dd[k]:8.0f}')
