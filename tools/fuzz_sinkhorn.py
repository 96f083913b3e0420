"""Randomised Sinkhorn / extraction cases through the per-op entry points against the fp64 oracle (GPU box):
    python tools/fuzz_sinkhorn.py [seconds] [seed]
Shapes 1 ... 1300 (ragged), 0 ... 60 iterations, score scales from 0.01 to 40 units, offsets, bin scores from -30 to 30, both
kernels (cluster / streaming).  Z must be finite and within 1e-4 (relative to max(1, |Z| / 50)) of the oracle."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdgat_matcher_amd import ops, synth
from oracle import mdgat_oracle as O


def run(budget=60.0, seed=0):
    rs = np.random.RandomState(seed)
    t0, cases, fails, worst = time.time(), 0, 0, 0.0
    sizes = [1, 2, 5, 33, 64, 100, 127, 128, 129, 255, 300, 512, 513, 640, 1000, 1300]
    while time.time() - t0 < budget:
        B = int(rs.choice([1, 2, 3, 9]))
        N, M = (int(x) for x in rs.choice(sizes, 2))
        iters = int(rs.choice([0, 1, 2, 5, 20, 60]))
        scale = float(rs.choice([0.01, 1.0, 4.0, 15.0, 40.0]))
        off = float(rs.choice([0.0, 50.0, -50.0]))
        alpha = float(rs.choice([1.0, 0.37, -30.0, 30.0, 0.0]))
        s = torch.from_numpy(rs.standard_normal((B, N, M)) * scale + off)
        ref = O.log_optimal_transport(s, alpha, iters)
        tol = 1e-4 * max(1.0, float(ref.abs().max()) / 50)       # (fp32 potentials of magnitude 100+ resolve 1e-5 at best)
        cases += 1
        for streaming in (False, True):
            Z = ops.sinkhorn(s.cuda(), alpha, iters, streaming=streaming).cpu().double()
            err = float((Z - ref).abs().max()) if bool(torch.isfinite(Z).all()) else float('inf')
            worst = max(worst, err / tol)
            if not err <= tol:
                fails += 1
                print('FAIL', dict(B=B, N=N, M=M, iters=iters, scale=scale, off=off, alpha=alpha, streaming=streaming, err=err, tol=tol))
    print(f'{cases} cases in {time.time() - t0:.0f} s, {fails} failures, worst error / tolerance {worst:.2f}')
    return cases, fails


if __name__ == '__main__':
    torch.set_num_threads(synth.effective_cpu_count())
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
