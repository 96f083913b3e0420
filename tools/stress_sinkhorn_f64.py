"""Concurrent launches of the fp64 Sinkhorn + extraction (csrc/sinkhorn_f64.hip: workgroups of a pair exchange column sums through the L2) on four streams\nagainst a serial run of the same cases: every output tensor must be bit-equal.  GPU box: python tools/stress_sinkhorn_f64.py"""
import sys, time, torch
sys.path.insert(0, '.')
from mdgat_matcher_amd import ops
torch.manual_seed(0)
cases = [(torch.randn(B, N, M, dtype=torch.float64, device='cuda') * 3, it) for (B, N, M, it) in [(3, 512, 512, 100), (9, 256, 300, 40), (1, 575, 575, 60), (20, 512, 512, 30), (40, 400, 400, 20), (2, 33, 17, 100)]]
ref = [ops.sinkhorn_f64_extract(s, 1.0, it, want_Z=True) for s, it in cases]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
bad = 0
t0 = time.time()
for rnd in range(12):
    outs = []
    for i, (s, it) in enumerate(cases * 2):
        with torch.cuda.stream(streams[(i + rnd) % 4]):
            outs.append((i % len(cases), ops.sinkhorn_f64_extract(s, 1.0, it, want_Z=True)))
    torch.cuda.synchronize()
    for ci, o in outs:
        for a, b in zip(o, ref[ci]):
            if not torch.equal(a, b):
                bad += 1
print(f'{12 * 2 * len(cases)} concurrent launches on 4 streams in {time.time() - t0:.1f} s: {bad} tensors differ from the serial run')
