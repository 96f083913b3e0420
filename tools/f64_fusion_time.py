"""Exact mode (a float64 module): the fused layer tail (csrc/layer_f64.hip) against the three-launch form, per batch size -
forward time (one lane, then the default two lanes) and the time of the fp64 product class (HIP events on the launch stream,
mdgat_profile).  GPU box:  python tools/f64_fusion_time.py [modes ...]   (modes: 0 = three launches, 1 = fused, 16 / 32 / 64 = fused
with that many keypoints per workgroup)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdgat_matcher_amd import MDGAT, _lib, synth  # noqa: E402

DEV = 'cuda:0'
modes = [int(x) for x in sys.argv[1:]] or [0, 1, 16, 32, 64]
lib = _lib.load()
L, S = 9, 100
net = MDGAT(synth.default_config(L=L, sinkhorn_iterations=S)).double()
net.load_state_dict(synth.make_state_dict(L=L, seed=0))
net = net.eval().to(DEV)
R_FLOPS = 2.0 * (128 * 384 + 256 * 256 + 256 * 128)
for B, N in ((1, 512), (2, 512), (8, 512), (32, 512), (64, 512), (1, 256), (2, 2048)):
    d = synth.make_batch(B, N, N, device=DEV)
    args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
    for mode in modes:
        lib.mdgat_set_f64_layer_fusion(mode)
        row = []
        for lanes in (1, 2):
            net.set_lanes(lanes)
            with torch.no_grad():
                for _ in range(3):
                    net._run(*args)
                torch.cuda.synchronize()
                reps = 20 if B <= 8 else 6
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        net._run(*args)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / reps * 1e3)
                row.append(sorted(ts)[1])
        net.set_lanes(1)
        net.profile(DEV, True)
        with torch.no_grad():
            for _ in range(3):
                net._run(*args)
        prof = net.profile(DEV, False)
        net.set_lanes(2)
        gms, gl = prof['f64_gemm']
        fl = 17 * B * 2 * N * R_FLOPS
        print(f'B={B} N={N} fusion={mode}: forward {row[0]:.3f} ms one lane, {row[1]:.3f} ms two lanes = {B / row[1] * 1e3:.0f} pairs/s | f64_gemm '
              f'{gms / 3:.3f} ms in {gl // 3} launches = {fl / (gms / 3 * 1e-3) / 1e12:.1f} TFLOP/s (layer products only) | att full '
              f'{prof["f64_attention_full"][0] / 3:.3f} topk {prof["f64_attention_topk"][0] / 3:.3f} ms', flush=True)
lib.mdgat_set_f64_layer_fusion(-1)
net.check(DEV)
