"""Sinkhorn cluster kernel alone at shapes beyond 512 keypoints (GPU box): ms per launch of 100 iterations."""
import sys, time, torch
sys.path.insert(0, '.')
from mdgat_matcher_amd import ops
for (B, N, M) in ((8, 1024, 1024), (4, 2048, 1024), (8, 2048, 2048), (8, 1024, 2048), (8, 1024, 512), (8, 2048, 512), (64, 512, 512)):
    s = torch.randn(B, N, M, device='cuda') * 3
    for _ in range(3):
        ops.sinkhorn(s, 1.0, 100)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        ops.sinkhorn(s, 1.0, 100)
    torch.cuda.synchronize(); print(B, N, M, round((time.perf_counter() - t) / 5 * 1e3, 3), 'ms')
