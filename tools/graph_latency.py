"""Latency of one forward with and without a HIP graph (GPU box): python tools/graph_latency.py [config] [reps]
BASELINE configs[0] (one pair, 256 keypoints, L=4, 20 Sinkhorn iterations) is ~25 short launches: the question is how
much of its latency is launch overhead.  The forward is captured through torch.cuda.CUDAGraph (= hipGraph on ROCm) on
torch's capture stream - the library launches on the stream it is given and allocates nothing itself."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mdgat_matcher_amd import MDGAT, synth  # noqa: E402


def main():
    ci = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    c = bench.CONFIGS[ci]
    dev = torch.device('cuda', 0)
    cfg = synth.default_config(L=c['L'], sinkhorn_iterations=c['S'])
    net = MDGAT(cfg).eval()
    net.load_state_dict(synth.make_state_dict(L=c['L'], seed=0, dtype=torch.float32))
    d = synth.make_batch(c['B'], c['n'], c['n'], dtype=torch.float32, device=dev)
    inputs = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
    with torch.no_grad():
        for _ in range(10):
            ref = net._run(*inputs)
        torch.cuda.synchronize()

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        eager = timed(lambda: net._run(*inputs))
        # one at a time: launch, wait, launch ... (the latency a caller with a single pair sees)
        def one():
            net._run(*inputs)
            torch.cuda.synchronize()
        eager_sync = timed(one)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net._run(*inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(graph):
                out = net._run(*inputs)
        except Exception as e:                      # noqa: BLE001
            print(f'{c["name"]}: eager back-to-back {eager:.4f} ms, eager one-at-a-time {eager_sync:.4f} ms; graph capture FAILED: {e}')
            return
        graphed = timed(graph.replay)

        def one_g():
            graph.replay()
            torch.cuda.synchronize()
        graphed_sync = timed(one_g)
        same = all(torch.equal(a, b) for a, b in zip(ref[:4], out[:4]))
        print(f'{c["name"]}: forward eager {eager:.4f} ms back-to-back / {eager_sync:.4f} ms one at a time; '
              f'as a HIP graph {graphed:.4f} ms back-to-back / {graphed_sync:.4f} ms one at a time; identical outputs: {same}')


if __name__ == '__main__':
    main()
