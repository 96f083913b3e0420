"""Does a second half-batch on a second stream fill the issue slots the latency-bound Sinkhorn leaves (GPU box)?
One forward of B pairs on one stream against two forwards of B/2 pairs on two streams (two handles), steps back to back."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NSPLIT = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 60
d = synth.make_batch(B, 512, 512, dtype=torch.float32, device=dev)
keys = ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1')
full = tuple(d[k] for k in keys)
parts = [tuple(d[k][i * (B // NSPLIT):(i + 1) * (B // NSPLIT)].contiguous() for k in keys) for i in range(NSPLIT)]
sd = synth.make_state_dict(L=9, seed=0, dtype=torch.float32)
nets = []
for _ in range(NSPLIT):
    n = MDGAT(synth.default_config(L=9, sinkhorn_iterations=100)).eval()
    n.load_state_dict(sd)
    nets.append(n)
streams = [torch.cuda.Stream(dev) for _ in range(NSPLIT)]
with torch.no_grad():
    for _ in range(5):
        nets[0]._run(*full)
        for n, p in zip(nets, parts): n._run(*p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): nets[0]._run(*full)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        for n, p, s in zip(nets, parts, streams):
            with torch.cuda.stream(s): n._run(*p)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'B={B}: one stream {(t1 - t0) / steps * 1e3:.3f} ms/step ({B * steps / (t1 - t0):.0f} pairs/s); '
      f'{NSPLIT} streams x B/{NSPLIT}: {(t2 - t1) / steps * 1e3:.3f} ms/step ({B * steps / (t2 - t1):.0f} pairs/s)')
