"""Forward timing of the BASELINE.json configurations on one GPU: python tools/time_configs.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth

dev = torch.device('cuda', 0)
for name, B, n, L, S in (('configs[0] B=1 N=256 L=4 S=20', 1, 256, 4, 20), ('configs[1] B=64 N=512 L=9 S=100', 64, 512, 9, 100),
                         ('B=512 N=512 L=9 S=100', 512, 512, 9, 100), ('configs[4] B=1 N=2048 L=9 S=200', 1, 2048, 9, 200),
                         ('configs[4] B=8 N=2048 L=9 S=200', 8, 2048, 9, 200)):
    cfg = synth.default_config(L=L, sinkhorn_iterations=S)
    net = MDGAT(cfg).eval(); net.load_state_dict(synth.make_state_dict(L=L, seed=0, dtype=torch.float32))
    one = synth.make_batch(1, n, n, dtype=torch.float32, device=dev)
    inp = tuple(one[k].expand(B, *one[k].shape[1:]).contiguous() for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1'))
    with torch.no_grad():
        for _ in range(10): net._run(*inp)      # (a one-off 25-80 ms stall hits one of the first steps of a process)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): net._run(*inp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'{name}: {dt * 1e3:.3f} ms/batch, {B / dt:.0f} pairs/s')
    del net
