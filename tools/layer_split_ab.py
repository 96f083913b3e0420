"""A/B of the two layer-tail kernels over batch sizes: csrc/layer_split.hip (channel-split, 32-keypoint workgroups) against
csrc/layer.hip (a wave owns 16 keypoints).  Prints ms per forward (median of `reps` timed loops) for each setting.
usage: python tools/layer_split_ab.py [n=512] [L=9] [S=100] [batches=1,2,4,8,16,32]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from mdgat_matcher_amd import MDGAT, _lib, synth  # noqa: E402

DEV = 'cuda:0'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    batches = [int(b) for b in (sys.argv[4] if len(sys.argv) > 4 else '1,2,4,8,16,32').split(',')]
    lib = _lib.load()
    net = MDGAT(synth.default_config(L=L, sinkhorn_iterations=S))
    net.load_state_dict(synth.make_state_dict(L=L, seed=1))
    net = net.eval().to(DEV)
    for B in batches:
        d = synth.make_batch(B, n, n, dtype=torch.float32, device=DEV)
        args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
        row = {}
        for name, tiles in (('whole', 0), ('split', 1 << 20), ('whole2', 0), ('split2', 1 << 20)):
            lib.mdgat_set_layer_split_tiles(tiles)
            for _ in range(5):
                net._run(*args)
            torch.cuda.synchronize()
            ts = []
            steps = max(5, min(50, 2000 // max(1, B * n // 64)))
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(steps):
                    net._run(*args)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / steps * 1e3)
            row[name] = float(np.median(ts))
        lib.mdgat_set_layer_split_tiles(-1)
        print(f'B={B:3d} n={n} L={L} S={S}  tiles={(2 * B * n + 127) // 128:4d}  ' + '  '.join(f'{k} {v:.4f} ms' for k, v in row.items()), flush=True)


if __name__ == '__main__':
    with torch.no_grad():
        main()
