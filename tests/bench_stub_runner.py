"""Test helper (CPU, no GPU): runs bench.py's main() with MDGAT._run replaced by a recorder, so that the multi-rank control
flow of bench.py - process group, weight broadcast, partition, barriers, max-over-ranks timing, per-rank gather, the one JSON
line of rank 0 - executes under torch.distributed.run with the gloo backend in the build container.  Launched by
tests/test_shard_gloo.py::test_bench_eight_ranks_gloo_stub; never part of a measurement (the line carries "stub": true)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['MDGAT_BENCH_STUB'] = '1'

import torch  # noqa: E402

from mdgat_matcher_amd import MDGAT  # noqa: E402


def _fake_run(self, kpts0, sigma0, fpfh0, kpts1, sigma1, fpfh1, want_Z=False, taps=None, frames=None, normalize=True, token_out=None):
    B, N, M = kpts0.shape[0], kpts0.shape[1], kpts1.shape[1]
    rank = int(os.environ.get('RANK', '0'))
    time.sleep(0.02 if rank == 3 else 0.001)        # rank 3 is the straggler the line must show (by a margin no scheduler noise of a busy box covers)
    log = os.environ.get('MDGAT_BENCH_STUB_LOG')
    if log:
        with open(f'{log}.{rank}', 'a') as f:
            f.write(f'{B} {N} {M}\n')
    z = torch.zeros
    return (z(B, N, dtype=torch.int64), z(B, M, dtype=torch.int64), z(B, N), z(B, M), z(B, N + 1, M + 1) if want_Z else None)


MDGAT._run = _fake_run
import bench  # noqa: E402

if __name__ == '__main__':
    bench.main()
