"""Aggregate throughput with several forwards in flight on several streams (GPU box):
    python tools/overlap_streams.py ["streams:batch,streams:batch,..."]
N = M = 512, L = 9, 100 Sinkhorn iterations (the bench shape).  One Python thread enqueues round-robin; every stream has its own
inputs and workspace.  Prints pairs/s per configuration and how many Sinkhorn launches fell back to the streaming kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth

spec = sys.argv[1] if len(sys.argv) > 1 else '1:64,2:32,2:64,4:16,3:64,1:128'
dev = torch.device('cuda', 0)
cfg = synth.default_config(L=9, sinkhorn_iterations=100)
net = MDGAT(cfg).eval()
net.load_state_dict(synth.make_state_dict(L=9, seed=0, dtype=torch.float32))
net = net.to(dev)
for item in spec.split(','):
    ns, B = (int(v) for v in item.split(':'))
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    inputs = []
    for i in range(ns):
        d = synth.make_batch(B, 512, 512, first_pair=B * i, dtype=torch.float32, device=dev)
        inputs.append(tuple(d[k] for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1')))
    torch.cuda.synchronize()

    def rounds(n):
        for _ in range(n):
            for st, inp in zip(streams, inputs):
                with torch.cuda.stream(st):
                    net._run(*inp)
    with torch.no_grad():
        rounds(5)
        torch.cuda.synchronize()
        net.check(dev)
        best = 0.0
        for rep in range(3):
            reps = max(4, 1280 // (ns * B))
            t0 = time.perf_counter()
            rounds(reps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = max(best, reps * ns * B / dt)
        fb = net.check(dev)['sinkhorn_fallback']
    print(f'{ns} stream(s) x B={B}: {best:.0f} pairs/s (best of 3), Sinkhorn fallback seen: {fb}', flush=True)
