"""Parity numbers at the BASELINE configs with the default dynamic schedule (GPU box):
    python tools/parity_report.py [pairs_c0 pairs_c1 pairs_c4] > profiles/parity_rN.txt

Per configuration and pair: max|dZ| of the HIP path against the fp64 oracle run with the HIP top-k selections forced,
whether the matches are identical, the number of dynamic-attention rows whose selection differs from the oracle's own
top-k and the largest distance of a disagreeing key from the k-th logit (tests/parity_util.py states the bar), then
the same pair against the PLAIN fp64 oracle (what a flip costs), and - as a yardstick for the arithmetic - the flips
a plain fp32 PyTorch run of the oracle makes against fp64 on the same pairs.

Round 3 adds the CAUSE of the flips: the HIP path's own fp32 descriptors entering every dynamic layer (taps) are fed to
an fp64 q/k projection and an fp64 q.k - the selection that exact arithmetic makes of the HIP path's INPUT.  Rows where
the HIP selection differs from that one are caused inside the layer (projection + split-f16 q.k products, what an fp64
re-evaluation of the near-threshold logits could repair); the remaining flips against the fp64 trajectory come with the
input (error accumulated by the layers before) and no local re-evaluation reaches them.

Round 6: the in-layer re-decision of round 4 (exact_topk, csrc/repair.hip) is retired - it did not change the number of flipped
rows (profiles/parity_r4.txt); the reference-exact mode (a float64 module, arithmetic='fp64') is the answer and
ARITHMETIC=fp64 in the environment runs this report on it.  The default here is the fp32-class path, pinned explicitly.  The
modules are cast to double BEFORE the fp64 state dict is loaded (as tools/make_goldens.py builds the reference), so that the
packed blob is the fp32 rounding of the fp64 weights and not of an fp32 copy of them."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mdgat_matcher_amd import MDGAT, synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402
from parity_util import attributed_parity, local_flips  # noqa: E402


def fp32_flips(sd, cfg, data, own64):
    """selection differences of a plain fp32 PyTorch run (free-running, own trajectory) against the fp64 oracle's"""
    sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    d32 = {k: v.float() for k, v in data.items()}
    cap = {}
    O.mdgat_forward(sd32, cfg, d32, cap, forced_topk={})
    rows = 0
    for i, reps in cap['topk_report'].items():
        for side in range(2):
            rows += int((reps[side]['own'] ^ own64[i][side]).any(-1).sum())
    return rows, cap['Z'].double()


def main():
    torch.set_num_threads(synth.effective_cpu_count())
    counts = [int(x) for x in sys.argv[1:4]] + [8, 8, 1][len(sys.argv[1:4]):]
    configs = [('configs[0] N=256 L=4 S=20', 256, 4, 20, counts[0]), ('configs[1] N=512 L=9 S=100', 512, 9, 100, counts[1]),
               ('configs[4] N=2048 L=9 S=200', 2048, 9, 200, counts[2])]
    print('# tools/parity_report.py: HIP path vs fp64 oracle, default k schedule [128, None, 128, None, 64, None, 64, None], weights seed 0')
    for name, n, L, S, pairs in configs:
        cfg = synth.default_config(L=L, sinkhorn_iterations=S)
        sd = synth.make_state_dict(L=L, seed=0)
        net = MDGAT({**cfg, 'arithmetic': os.environ.get('ARITHMETIC', 'fp32')}).double()
        net.load_state_dict(sd)
        net = net.double().eval().to('cuda:0')
        tot_rows = tot_flip = tot_flip32 = tot_local = literal = would_pass = 0
        worst = worst_gap = worst_plain = worst32 = 0.0
        all_equal = True
        t0 = time.time()
        for p in range(pairs):
            data = synth.make_batch(1, n, n, first_pair=100 + p)
            r = attributed_parity(net, cfg, sd, data)
            cap = {}
            ref = O.mdgat_forward(sd, cfg, data, cap, forced_topk={})
            own64 = {i: (rep[0]['own'], rep[1]['own']) for i, rep in cap['topk_report'].items()}
            plain = (r['out'][4].cpu().double() - cap['Z']).abs().max().item()
            mm = int((r['out'][0].cpu() != ref['matches0']).sum() + (r['out'][1].cpu() != ref['matches1']).sum())
            f32rows, Z32 = fp32_flips(sd, cfg, data, own64) if n <= 512 else (-1, None)
            e32 = (Z32 - cap['Z']).abs().max().item() if Z32 is not None else float('nan')
            loc = sum(local_flips(net, sd, data).values())
            tot_local += max(loc, 0)
            literal += plain < 1e-4
            # a pair whose flips are ALL caused inside their layer would meet the literal bar if the near-threshold logits
            # were re-evaluated exactly (fp64 projection + product of the candidates); one input-borne flip and it does not
            would_pass += (plain < 1e-4) or (0 <= loc and loc >= r['flip_rows'])
            print(f'{name} pair {100 + p}: forced-selection max|dZ| {r["errZ"]:.2e} matches identical {r["matches_equal"]} | '
                  f'rows differing {r["flip_rows"]}/{r["topk_rows"]} max gap {r["max_gap"]:.2e} kept!=k {r["bad_count"]} | '
                  f'vs plain fp64 oracle: max|dZ| {plain:.2e}, matches differing {mm} | fp32 PyTorch: rows differing {f32rows}, max|dZ| {e32:.2e} | '
                  f'rows differing from the fp64 selection of the HIP path\'s own layer input (caused inside the layer): {loc}')
            tot_rows += r['topk_rows']; tot_flip += r['flip_rows']; tot_flip32 += max(f32rows, 0)
            worst = max(worst, r['errZ']); worst_gap = max(worst_gap, r['max_gap']); worst_plain = max(worst_plain, plain)
            worst32 = max(worst32, e32 if e32 == e32 else 0.0)
            all_equal &= r['matches_equal']
            sys.stdout.flush()
        print(f'== {name}: {pairs} pairs, {time.time() - t0:.0f} s: worst forced-selection max|dZ| {worst:.2e} (bar 1e-4), matches identical: {all_equal}, '
              f'top-k rows differing {tot_flip} of {tot_rows} ({tot_flip / max(tot_rows, 1):.2e}), worst gap {worst_gap:.2e}; '
              f'plain fp64 comparison worst max|dZ| {worst_plain:.2e}; fp32 PyTorch rows differing {tot_flip32}, worst max|dZ| {worst32:.2e}; '
              f'pairs within the LITERAL 1e-4 against the plain fp64 oracle: {literal}/{pairs}; flips caused inside the dynamic layer '
              f'(q/k projection + q.k products, given the HIP input): {tot_local} of {tot_flip} - the rest arrives with the layer input; '
              f'pairs that WOULD meet the literal bar with an exact re-evaluation of near-threshold logits inside the layer (upper bound: '
              f'every in-layer flip repaired, none created): {would_pass}/{pairs}')


if __name__ == '__main__':
    main()
