"""One pair per call in the exact mode (a float64 module), N = M = 512, L = 9, S = 100: 30 forwards back to back - for a kernel trace
(rocprofv3 --kernel-trace --stats -- python tools/one_pair_exact.py) of what test.py:132's batch_size = 1 costs per launch."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdgat_matcher_amd import MDGAT, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
net = MDGAT(synth.default_config(L=9, sinkhorn_iterations=100)).double()
net.load_state_dict(synth.make_state_dict(L=9, seed=0))
net = net.eval().to('cuda:0')
d = synth.make_batch(1, n, n, device='cuda:0')
args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
with torch.no_grad():
    for _ in range(5):
        net._run(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        net._run(*args)
    torch.cuda.synchronize()
print(f'one pair per call, exact mode, N={n}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms')
net.check('cuda:0')
