"""Forward time against the batch size at N=M=512, L=9, 100 Sinkhorn iterations: python tools/time_batches.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth

dev = torch.device('cuda', 0)
cfg = synth.default_config(L=9, sinkhorn_iterations=100)
net = MDGAT(cfg).eval(); net.load_state_dict(synth.make_state_dict(L=9, seed=0, dtype=torch.float32))
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    d = synth.make_batch(B, 512, 512, dtype=torch.float32, device=dev)
    inp = tuple(d[k] for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1'))
    with torch.no_grad():
        for _ in range(10): net._run(*inp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): net._run(*inp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'B={B}: {dt * 1e3:.3f} ms/batch, {B / dt:.0f} pairs/s')
