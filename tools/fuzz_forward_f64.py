"""Randomised shapes / configurations through the whole forward in the REFERENCE-EXACT mode (arithmetic='fp64') against the
UNFORCED fp64 oracle (GPU box):
    python tools/fuzz_forward_f64.py [seconds] [seed]
Every case: B, N, M, L, Sinkhorn iterations, top-k schedule, extraction mode, bin score drawn at random (frames up to 900
keypoints, not multiples of anything).  The bar is the literal one - no attribution of top-k flips, because there are none:
max|dZ| < 1e-4 against the oracle's own run, the kept keys of every dynamic layer equal to the oracle's (zero rows in its report
when the library's selections are fed back), matches = the extraction rules applied to the library's own Z, and identical to the
oracle's wherever the oracle's deciding entries of Z are more than 1e-4 apart (few Sinkhorn iterations on tiny frames produce
near-ties between two entries of Z: the arg-max of a near-tie is not a property of either implementation).
tests/test_gpu_f64.py::test_fuzz_forward_f64_short runs 20 s of it."""
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
from parity_util import hip_forward_with_selection  # noqa: E402

SIZES = [1, 7, 31, 64, 100, 128, 129, 200, 256, 300, 512, 513, 577, 700, 900]
MODES = [('triplet_loss', False), ('triplet_loss', True), ('superglue', False), ('superglue', True)]


def one_case(rs):
    B = int(rs.choice([1, 2, 3, 5]))
    N, M = (int(x) for x in rs.choice(SIZES, 2))
    L = int(rs.choice([1, 2, 3]))
    S = int(rs.choice([2, 7, 30]))
    kmax = min(N, M)
    k = [None if rs.uniform() < 0.4 else int(rs.randint(1, kmax + 1)) for _ in range(int(rs.choice([1, 2, 2 * L])))]
    loss_method, mutual = MODES[int(rs.randint(4))]
    if mutual and B != 1 and loss_method != 'superglue':
        mutual = False                      # (the reference's dustbin-mutual branch only works for batch 1)
    bin_score = float(rs.choice([1.0, 0.37, -2.0, 6.0]))
    wseed, fp = int(rs.randint(100)), int(rs.randint(1000))
    f64_layers = int(rs.choice([-1, -1, 2 * L]))
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, loss_method=loss_method, mutual_check=mutual)
    sd = synth.make_state_dict(L=L, seed=wseed, bin_score=bin_score)
    net = MDGAT({**cfg, 'arithmetic': 'fp64', 'f64_layers': f64_layers}).double()
    net.load_state_dict(sd)
    net = net.double().eval().to('cuda:0')
    data = synth.make_batch(B, N, M, first_pair=fp)
    tag = f'B={B} N={N} M={M} L={L} S={S} k={k} {loss_method} mutual={mutual} bin={bin_score} wseed={wseed} first_pair={fp} f64_layers={f64_layers}'
    r = {}
    try:
        dev = {kk: v.to('cuda:0') for kk, v in data.items()}
        (m0, m1, s0, s1, Z), forced = hip_forward_with_selection(net, dev)
        net.check('cuda:0')
        cap = {}
        ref = O.mdgat_forward(sd, cfg, data, cap)                                  # the oracle's OWN run
        r['errZ'] = float((Z.cpu().double() - cap['Z']).abs().max())
        cap2 = {}
        O.mdgat_forward(sd, cfg, data, cap2, forced_topk=forced)                   # ... and with the library's selections fed back
        r['flip_rows'] = sum(x['rows'] for reps in cap2.get('topk_report', {}).values() for x in reps)
        r['bad_count'] = sum(x['bad_count'] for reps in cap2.get('topk_report', {}).values() for x in reps)
        tail64 = f64_layers < 0 and max(N, M) <= 575           # every layer, the scores and the Sinkhorn in fp64 too (sinkhorn_f64.hip)
        if tail64:
            # the arg-maxes are decided on the fp64 Z, of which the returned Z is the fp32 rounding (two candidates 1e-9 apart are equal
            # there): the matches must be the ORACLE's own, all of them
            r['matches_vs_own_Z'] = bool(torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1']))
        else:
            e0, e1, es0, es1 = O.extract_matches(Z.cpu().double(), loss_method, mutual, cfg['match_threshold'])
            r['matches_vs_own_Z'] = bool(torch.equal(m0.cpu(), e0) and torch.equal(m1.cpu(), e1))
        # matches against the oracle's: identical unless the oracle's Z holds a near-tie where they differ
        diff0 = (m0.cpu() != ref['matches0'])
        Zo = cap['Z']
        top2 = Zo[:, :-1, :].topk(2, dim=2).values if Zo.shape[2] > 1 else None
        gap0 = (top2[..., 0] - top2[..., 1]) if top2 is not None else torch.ones_like(Zo[:, :-1, 0])
        r['match_diffs_not_near_ties'] = int((diff0 & (gap0 > 1e-4)).sum()) if loss_method != 'superglue' else 0
        ok = r['errZ'] < 1e-4 and r['flip_rows'] == 0 and r['bad_count'] == 0 and r['matches_vs_own_Z'] and r['match_diffs_not_near_ties'] == 0
    except Exception as e:                  # noqa: BLE001
        print('EXCEPTION', tag, repr(e))
        ok, r = False, {'errZ': float('nan')}
    return ok, tag, r


def run(budget=60.0, seed=0, verbose=True):
    """Returns (cases run, failures, worst max|dZ| against the unforced oracle)."""
    rs = np.random.RandomState(seed)
    t0, cases, fails, worst = time.time(), 0, 0, 0.0
    while time.time() - t0 < budget:
        cases += 1
        ok, tag, r = one_case(rs)
        worst = max(worst, r['errZ'] if r['errZ'] == r['errZ'] else 0.0)
        if not ok:
            fails += 1
            if verbose:
                print('FAIL', tag, r)
    if verbose:
        print(f'{cases} cases in {time.time() - t0:.0f} s, {fails} failures, worst max|dZ| against the UNFORCED oracle {worst:.2e}')
    return cases, fails, worst


if __name__ == '__main__':
    torch.set_num_threads(synth.effective_cpu_count())
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
