"""Randomised shapes / configurations through the whole forward against the fp64 oracle (GPU box):
    python tools/fuzz_forward.py [seconds] [seed]
Every case: B, N, M, L, Sinkhorn iterations, top-k schedule, extraction mode, bin score drawn at random (frames up to 700
keypoints, not multiples of anything).  Z must match the oracle run with the HIP selections forced to 1e-4
(tests/parity_util.py), matches and scores must be what the extraction rules make of that Z, every dynamic row must hold
exactly k keys, and the handle must not report a range violation (a Sinkhorn launch handed to the log-domain kernel
because its scores span more than the scaling form holds is fine - and counted).  tests/test_gpu_forward.py::test_fuzz_short runs 15 s of it."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mdgat_matcher_amd import MDGAT, synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402
from parity_util import attributed_parity  # noqa: E402

SIZES = [1, 7, 31, 64, 100, 128, 129, 200, 256, 300, 512, 513, 700]
MODES = [('triplet_loss', False), ('triplet_loss', True), ('superglue', False), ('superglue', True)]


def one_case(rs):
    B = int(rs.choice([1, 2, 3, 5, 8, 9]))
    N, M = (int(x) for x in rs.choice(SIZES, 2))
    L = int(rs.choice([1, 2, 3]))
    S = int(rs.choice([1, 2, 7, 30]))       # (0 iterations: exp of unnormalised scores overflows the fp32 matching scores beyond 88)
    kmax = min(N, M)
    k = [None if rs.uniform() < 0.4 else int(rs.randint(1, kmax + 1)) for _ in range(int(rs.choice([0, 1, 2, 2 * L])))]
    loss_method, mutual = MODES[int(rs.randint(4))]
    if mutual and B != 1 and loss_method != 'superglue':
        mutual = False                      # (the reference's dustbin-mutual branch only works for batch 1)
    bin_score = float(rs.choice([1.0, 0.37, -2.0, 6.0]))
    wseed, fp = int(rs.randint(100)), int(rs.randint(1000))
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, loss_method=loss_method, mutual_check=mutual, arithmetic='fp32')
    sd = synth.make_state_dict(L=L, seed=wseed, bin_score=bin_score)
    net = MDGAT(cfg).double()
    net.load_state_dict(sd)
    net = net.double().eval().to('cuda:0')
    data = synth.make_batch(B, N, M, first_pair=fp)
    tag = f'B={B} N={N} M={M} L={L} S={S} k={k} {loss_method} mutual={mutual} bin={bin_score} wseed={wseed} first_pair={fp}'
    try:
        r = attributed_parity(net, cfg, sd, data)
        # Z against the oracle (HIP selections forced): the arithmetic.  Matches and scores against the extraction rules of
        # mdgat.py:441-483 applied to the HIP path's OWN Z (oracle code, fp64): the extraction logic, free of the near-ties
        # between two entries of Z that few Sinkhorn iterations and tiny frames produce in numbers.
        m0, m1, s0, s1, Zh = r['out']
        e0, e1, es0, es1 = O.extract_matches(Zh.cpu().double(), loss_method, mutual, cfg['match_threshold'])
        smax = max(1.0, float(es0.abs().max()), float(es1.abs().max()))
        r['matches_vs_own_Z'] = bool(torch.equal(m0.cpu(), e0) and torch.equal(m1.cpu(), e1))
        r['err_mscores'] = max(float((s0.cpu().double() - es0).abs().max()), float((s1.cpu().double() - es1).abs().max()))
        ok = r['errZ'] <= 1e-4 and r['bad_count'] == 0 and r['max_gap'] < 2e-5 and r['matches_vs_own_Z'] and r['err_mscores'] <= 1e-5 * smax
        r['fallback'] = net.check('cuda:0')['sinkhorn_fallback']      # (information: scores beyond the range of the scaling form)
    except Exception as e:                  # noqa: BLE001
        print('EXCEPTION', tag, repr(e))
        ok, r = False, {'errZ': float('nan')}
    return ok, tag, r


def run(budget=60.0, seed=0, verbose=True):
    """Returns (cases run, failures, worst forced-selection max|dZ|)."""
    rs = np.random.RandomState(seed)
    t0, cases, fails, worst, fallbacks = time.time(), 0, 0, 0.0, 0
    while time.time() - t0 < budget:
        cases += 1
        ok, tag, r = one_case(rs)
        fallbacks += bool(r.get('fallback'))
        worst = max(worst, r['errZ'] if r['errZ'] == r['errZ'] else 0.0)
        if not ok:
            fails += 1
            if verbose:
                print('FAIL', tag, {kk: r.get(kk) for kk in ('errZ', 'matches_vs_own_Z', 'err_mscores', 'bad_count', 'max_gap', 'flip_rows')})
    if verbose:
        print(f'{cases} cases in {time.time() - t0:.0f} s, {fails} failures, {fallbacks} with a Sinkhorn range fallback, worst forced-selection max|dZ| {worst:.2e}')
    return cases, fails, worst


if __name__ == '__main__':
    torch.set_num_threads(synth.effective_cpu_count())
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
