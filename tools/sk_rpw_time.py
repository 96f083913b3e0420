#!/usr/bin/env python3
"""Sinkhorn cluster kernel with 16 / 8 / 4 rows per wave (MDGAT_SK_RPW; csrc/sinkhorn.hip: sk_rpw) at small batches, GPU box:
µs per launch (100 iterations) of the fused Sinkhorn + extraction through the forward's own entry, and the one-pair forward.
    python tools/sk_rpw_time.py            (spawns one process per setting)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, os, torch
sys.path.insert(0, os.environ["MDGAT_ROOT"])
from mdgat_matcher_amd import MDGAT, ops, synth
dev = "cuda:0"
def timed(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = []
for (B, n, S) in ((1, 512, 100), (2, 512, 100), (4, 512, 100), (8, 512, 100), (16, 512, 100), (64, 512, 100), (1, 256, 20), (1, 300, 100), (3, 450, 37)):
    s = torch.randn(B, n, n, device=dev) * 2
    ref = ops.sinkhorn(s, 1.0, S, streaming=True)
    Z = ops.sinkhorn(s, 1.0, S)
    err = (Z - ref).abs().max().item()
    t = timed(lambda: ops.sinkhorn(s, 1.0, S))
    out.append(f"B={B} n={n} S={S}: {t:.1f} us (vs streaming kernel max|dZ| {err:.1e})")
L = 9
net = MDGAT(synth.default_config(L=L)).eval()
net.load_state_dict(synth.make_state_dict(L=L, seed=0, dtype=torch.float32))
net = net.to(dev)
for B in (1, 4, 8):
    d = synth.make_batch(B, 512, 512, dtype=torch.float32, device=dev)
    args = (d["keypoints0"], d["scores0"], d["descriptors0"], d["keypoints1"], d["scores1"], d["descriptors1"])
    t = timed(lambda: net._run(*args), n=30, warm=10)
    out.append(f"forward B={B} N=512 L=9 S=100: {t:.1f} us per call")
print("\n".join(out))
'''
for rpw in sys.argv[1:] or ['16', '8', '4']:
    env = dict(os.environ, MDGAT_SK_RPW=rpw, MDGAT_ROOT=ROOT)
    p = subprocess.run([sys.executable, '-c', WORKER], env=env, capture_output=True, text=True, timeout=900)
    print(f'--- MDGAT_SK_RPW={rpw}')
    print(p.stdout.strip() or p.stderr[-2000:])
