"""Randomised kNN cases (knn / get_graph_feature, mdgat.py:8-32) through the per-op entry point against the fp64 oracle:
    python tools/fuzz_knn.py [seconds] [seed]
Channel counts 1 ... 128 (128 = the matrix-core path, with and without workspace), frames 1 ... 5000 ragged, k from 1 to
min(M, 1024).  Index lists identical to the oracle's on every row without a near-tie among its k + 1 nearest, adjacency = the
scatter of the returned indices, k ones per row."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdgat_matcher_amd import ops, synth
from oracle import mdgat_oracle as O


def run(budget=60.0, seed=0):
    rs = np.random.RandomState(seed)
    t0, cases, fails, excl = time.time(), 0, 0, 0.0
    while time.time() - t0 < budget:
        C = int(rs.choice([1, 2, 3, 5, 16, 33, 128, 128]))
        B = int(rs.choice([1, 2, 3]))
        N = int(rs.choice([1, 3, 40, 64, 100, 257, 512, 1000, 2048]))
        M = int(rs.choice([1, 2, 9, 64, 100, 300, 513, 2048, 4097, 5000]))
        k = int(rs.randint(1, min(M, 1024) + 1)) if rs.uniform() < 0.7 else min(M, int(rs.choice([1, 9, 20])))
        scale = 20.0 if C <= 3 else 1.0
        x = torch.from_numpy(scale * rs.standard_normal((B, C, N)))
        s = torch.from_numpy(scale * rs.standard_normal((B, C, M)))
        mfma = bool(rs.randint(2))
        tag = dict(B=B, C=C, N=N, M=M, k=k, mfma=mfma)
        cases += 1
        try:
            idx, adj = ops.knn(x.cuda(), s.cuda(), k, adjacency=True, mfma=mfma)
            ref = O.knn(x, s, k)
            inner = -2.0 * torch.matmul(x.transpose(2, 1), s)
            nd = -(x ** 2).sum(1, keepdim=True).transpose(2, 1) - inner - (s ** 2).sum(1, keepdim=True)
            top = nd.topk(min(k + 1, M), dim=-1).values
            eps = 2e-4 if C == 128 else 1e-4
            tol = eps * torch.clamp(top[..., 1:].abs() / 100.0, min=1.0)
            clear = ((top[..., :-1] - top[..., 1:]) > tol).all(-1) if top.shape[-1] > 1 else torch.ones(top.shape[:2], dtype=torch.bool)
            ok = torch.equal(idx.cpu()[clear], ref[clear]) and torch.equal(adj.cpu(), O.knn_adjacency(x, s, k, idx=idx.cpu())) \
                and bool((adj.sum(-1) == k).all())
            excl = max(excl, 1.0 - clear.double().mean().item() - 0.004 * k)
        except Exception as e:                  # noqa: BLE001
            ok = False; print('EXCEPTION', tag, repr(e))
        if not ok:
            fails += 1; print('FAIL', tag)
    print(f'{cases} cases in {time.time() - t0:.0f} s, {fails} failures')
    return cases, fails


if __name__ == '__main__':
    torch.set_num_threads(synth.effective_cpu_count())
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
