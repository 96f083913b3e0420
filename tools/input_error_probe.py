"""Build-container experiment (CPU): where do the INPUT-BORNE top-k flips come from?

The exact re-decision of near-threshold rows (csrc/repair.hip) removes the flips a dynamic layer causes itself; what is left
against the fp64 reference are rows whose order flips with the layer's input - the error the layers BEFORE have put into the
descriptors.  This script runs the oracle's forward with each product class emulated in a given arithmetic (tools/precision_probe.py:
`lin` weights x activations, `qk`, `pv`) with the reference's selections forced in every dynamic layer (so the trajectory is the
reference's up to arithmetic), and counts, per dynamic layer, the rows whose EXACT (fp64) top-k on the emulated descriptors differs
from the reference's top-k on its own descriptors - the flips no in-layer repair can reach.
    python tools/input_error_probe.py [pairs] [N] [L]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdgat_matcher_amd import synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402
import precision_probe as PP  # noqa: E402


def exact_selection(sd, i, kk, x0, x1):
    out = []
    for x in (x0, x1):       # self layers only in the default schedule
        p = f'gnn.layers.{i}.attn'
        B = x.shape[0]
        q = O._pointwise(sd[f'{p}.proj.0.weight'], sd[f'{p}.proj.0.bias'], x.double()).view(B, 32, 4, -1)
        k = O._pointwise(sd[f'{p}.proj.1.weight'], sd[f'{p}.proj.1.bias'], x.double()).view(B, 32, 4, -1)
        logits = torch.einsum('bdhn,bdhm->bhnm', q, k) / 32 ** 0.5
        out.append(torch.zeros_like(logits, dtype=torch.bool).scatter_(3, logits.topk(kk, dim=3).indices, True))
    return out


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    torch.set_num_threads(8)
    cfg = synth.default_config(L=L, sinkhorn_iterations=2)
    sd = synth.make_state_dict(L=L, seed=0)
    sd32 = {kk: (v.float() if v.dtype == torch.float64 else v) for kk, v in sd.items()}
    sched = O.layer_topk_schedule(L, cfg['k'])
    combos = (('f32', 'f32', 'f32'), ('f16x3', 'f16x3u', 'f16x3u'), ('f16x3', 'f32', 'f32'), ('f32', 'f16x3u', 'f16x3u'))
    tot = {c: 0 for c in combos}
    rows = 0
    with torch.no_grad():
        for p in range(pairs):
            data = synth.make_batch(1, n, n, first_pair=100 + p)
            d32 = {kk: (v.float() if v.dtype == torch.float64 else v) for kk, v in data.items()}
            cap64 = {}
            O.mdgat_forward(sd, cfg, data, cap64, forced_topk={})
            forced = {i: (r[0]['own'], r[1]['own']) for i, r in cap64['topk_report'].items()}
            for c in combos:
                _, cap = PP.run(c[0], c[1], c[2], sd32, cfg, d32, forced)
                flips = 0
                for i, kk in enumerate(sched):
                    if kk is None:
                        continue
                    x0 = cap[f'layer{i - 1}_desc0'] if i else cap['enc0']
                    x1 = cap[f'layer{i - 1}_desc1'] if i else cap['enc1']
                    s0, s1 = exact_selection(sd, i, kk, x0, x1)
                    flips += int((s0 ^ forced[i][0]).any(-1).sum()) + int((s1 ^ forced[i][1]).any(-1).sum())
                    if c == combos[0]:
                        rows += s0.shape[1] * s0.shape[2] * 2
                tot[c] += flips
                print(f'pair {100 + p} lin={c[0]} qk={c[1]} pv={c[2]}: input-borne flips {flips}', flush=True)
    print(f'== {pairs} pairs, N={n}, L={L}, {rows} dynamic rows: input-borne flips per arithmetic ' +
          ', '.join(f'{c[0]}/{c[1]}/{c[2]}: {tot[c]}' for c in combos))


if __name__ == '__main__':
    main()
