"""Attribution of top-k flips for the end-to-end parity tests (GPU side).

``dynamic_attention`` (mdgat.py:196-210) is discontinuous: when the k-th and (k+1)-th largest logit of a row are
closer than the arithmetic error of an implementation, that implementation keeps the other key and the row's
message moves by ~p_k |v_a - v_b| - orders of magnitude more than the arithmetic error that caused it.  The tests
therefore split the comparison with the fp64 oracle in two statements that together are the north-star bar:

1. with the HIP path's selections (``mdgat_taps.topk_sel``) forced into the oracle, Z agrees to 1e-4 EVERYWHERE and
   the matches are identical: the arithmetic is within the bar;
2. every row where the HIP selection differs from the oracle's own top-k (on the same, forced, trajectory) differs
   only in keys whose fp64 logit lies within GAP_EPS of the k-th largest: the selection is the reference's up to
   near-ties below the resolution of fp32-class arithmetic, and every forced selection holds exactly k keys.
Measured (profiles/parity_r2.txt): such rows are 1-2.5e-4 of all dynamic rows, their gaps < 8e-6, and a plain fp32
PyTorch run of the oracle disagrees with fp64 on just as many rows - it is what fp32 resolution costs, not this
implementation's split-f16 products.

The attributed form never stands alone: ``assert_plain`` holds the SAME output to an unconditional comparison with
the unforced fp64 result (the reference's golden output or the plain oracle) - matches bit-identical, the plain
max|dZ| and the fraction of entries beyond 1e-4 bounded, and the literal 1e-4 bar on every pair in which no selection
flipped - so a regression of the selection or the extraction cannot hide behind the forced trajectory.  And the
tapped kernel instantiations (``mdgat_taps.topk_sel``) must give the bits of the untapped ones.
"""
import torch

from mdgat_matcher_amd import ops
from oracle import mdgat_oracle as O

Z_TOL = 1e-4
# Largest |logit - k-th logit| (natural units of q.k / sqrt(32), logits are O(10)) a disagreeing key may have: the
# accumulated fp32-class error of the logits after up to 17 layers (measured worst case 7.5e-6: profiles/parity_r2.txt).
GAP_EPS = 2e-5
# Unconditional bounds against the UNFORCED fp64 result.  One flipped near-tie moves one keypoint's message by
# ~p_k |v_a - v_b| and Z by a few 1e-4 around that keypoint (measured worst case over 34 pairs: 8.1e-4 at configs[1],
# 1.6e-3 at N=2048; profiles/parity_r2.txt, parity_r3.txt); an arithmetic or selection bug moves it by 1e-2 and more.
TAP_EPS = 2e-5            # tapped vs shipped kernels on Z (exact-tie rows only - the ~1e-6 rounding of the correction on a
                          # message, carried through the remaining layers: measured <= 6.7e-6; everything else is bit-identical)
PLAIN_MAX = 2e-3          # max|dZ| with flips present
PLAIN_FRAC = 5e-3         # fraction of Z entries beyond 1e-4 with flips present (measured: up to 2.8e-3 of ONE pair's entries
                          # at N = 2048 with 17 flipped rows; 8-pair batches stay below 1e-3)


def hip_forward_with_selection(net, dev_data):
    """Run the HIP forward with Z and the top-k selection tap; returns ((m0, m1, s0, s1, Z), forced_topk dict)."""
    k0 = dev_data['keypoints0']
    B, N, M = k0.shape[0], k0.shape[1], dev_data['keypoints1'].shape[1]
    sched = net._topk_schedule()
    words = ops.topk_sel_words(B, N, M)
    sel = torch.zeros(len(sched) * words, dtype=torch.int32, device=k0.device)
    outs = net._run(k0, dev_data['scores0'], dev_data['descriptors0'], dev_data['keypoints1'], dev_data['scores1'],
                    dev_data['descriptors1'], want_Z=True, taps={'topk_sel': sel})
    torch.cuda.synchronize()
    forced = {}
    for i, kk in enumerate(sched):
        if kk > 0:
            m0, m1 = ops.topk_sel_to_masks(sel[i * words:(i + 1) * words], B, N, M, cross=bool(i & 1))
            forced[i] = (m0.cpu(), m1.cpu())
    return outs, forced


def attributed_parity(net, cfg, sd, data_cpu, device='cuda:0'):
    """HIP forward vs the fp64 oracle with the HIP selections forced.  Returns a dict of measured quantities."""
    dev = {k: v.to(device) for k, v in data_cpu.items()}
    (m0, m1, s0, s1, Z), forced = hip_forward_with_selection(net, dev)
    # The selection tap runs separate kernel instantiations (TAP = true): what is compared with the oracle must be what
    # ships.  Matches: identical.  Floats: identical except where a row's k-th place falls on EXACTLY equal fp32 logits
    # (about one row in 10^5): the tapped kernels drop the surplus tied key before their softmax pass, the shipped ones
    # take its share out of the written row afterwards (attention.hip, "exactly k keys") - the same selection, the
    # correction's rounding apart (~1e-6 on the message; TAP_EPS bounds what the later layers make of it in Z).
    plain = net._run(dev['keypoints0'], dev['scores0'], dev['descriptors0'], dev['keypoints1'], dev['scores1'],
                     dev['descriptors1'], want_Z=True)
    assert torch.equal(m0, plain[0]) and torch.equal(m1, plain[1]), 'tapped and untapped forward differ in the matches'
    for a, b, what in zip((s0, s1, Z), plain[2:], ('mscores0', 'mscores1', 'Z')):
        assert torch.equal(a, b) or (a - b).abs().max().item() <= TAP_EPS, \
            f'tapped and untapped forward differ in {what} by {(a - b).abs().max().item():.2e}'
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data_cpu, cap, forced_topk=forced)
    rows = total = bad = 0
    max_gap = 0.0
    per_pair = torch.zeros(Z.shape[0], dtype=torch.int64)
    for reps in cap.get('topk_report', {}).values():
        for r in reps:
            rows += r['rows']
            total += r['total_rows']
            bad += r['bad_count']
            max_gap = max(max_gap, r['max_gap'])
            per_pair += r['rows_per_pair']
    return {
        'flips_per_pair': per_pair,
        'errZ': (Z.cpu().double() - cap['Z']).abs().max().item(),
        'matches_equal': bool(torch.equal(m0.cpu(), ref['matches0']) and torch.equal(m1.cpu(), ref['matches1'])),
        'err_mscores': max((s0.cpu().double() - ref['matching_scores0']).abs().max().item(),
                           (s1.cpu().double() - ref['matching_scores1']).abs().max().item()),
        'flip_rows': rows, 'topk_rows': total, 'bad_count': bad, 'max_gap': max_gap,
        'out': (m0, m1, s0, s1, Z), 'Z_forced': cap['Z'], 'ref_forced': ref,
    }


def assert_attributed(res, tag=''):
    print(f'[parity] {tag}: max|dZ| (HIP selection forced) {res["errZ"]:.3e}, mscores {res["err_mscores"]:.3e}, '
          f'matches identical {res["matches_equal"]}; top-k rows differing from the fp64 selection '
          f'{res["flip_rows"]} of {res["topk_rows"]}, largest gap to the k-th logit {res["max_gap"]:.3e}')
    assert res['errZ'] <= Z_TOL, res['errZ']
    assert res['matches_equal']
    assert res['err_mscores'] <= Z_TOL
    assert res['bad_count'] == 0          # every dynamic row kept exactly k keys
    assert res['max_gap'] < GAP_EPS, res['max_gap']
    # near-ties below GAP_EPS are rare: at most 4x the measured mean rate of 2e-4 (profiles/parity_r2.txt)
    assert res['flip_rows'] <= max(4, int(8e-4 * res['topk_rows'])), (res['flip_rows'], res['topk_rows'])


def near_tie_mismatches(Z, m0, m1, ref_m0, ref_m1, tol):
    """Entries of matches0 / matches1 (default extraction branch, mdgat.py:461-464, 480-483: arg-max over a row / column of Z including
    the dustbin) that differ from the reference's, each checked to be a NEAR TIE by the HIP path's own Z: the reference's choice lies
    within ``tol`` of the maximum this path found.  Returns the list of (side, pair, index, mine, ref, gap); asserts that every
    mismatch is such a near tie.  (An arg-max is decided by the gap between its two best candidates; a Z that is accurate to 6e-6 -
    the exact mode, whose Sinkhorn runs in fp32 - cannot order candidates the reference's fp64 Z separates by 1e-6.)"""
    import numpy as np
    Zc = Z.cpu().double().numpy()
    B, N1, M1 = Zc.shape
    N, M = N1 - 1, M1 - 1
    out = []
    for side, mine, ref in ((0, m0.cpu().numpy(), np.asarray(ref_m0)), (1, m1.cpu().numpy(), np.asarray(ref_m1))):
        for b, i in zip(*np.nonzero(mine != ref)):
            lim = M if side == 0 else N
            a = int(mine[b, i]) if mine[b, i] >= 0 else lim          # -1 = the dustbin = the last column / row
            r = int(ref[b, i]) if ref[b, i] >= 0 else lim
            gap = (Zc[b, i, a] - Zc[b, i, r]) if side == 0 else (Zc[b, a, i] - Zc[b, r, i])
            out.append((side, int(b), int(i), int(mine[b, i]), int(ref[b, i]), float(gap)))
            assert 0.0 <= gap < tol, f'matches{side}[{b}, {i}] = {mine[b, i]} (reference {ref[b, i]}) is not a near tie: Z gap {gap:.3e}'
    return out


def assert_plain(res, ref_Z, ref_m0, ref_m1, ref_s0, ref_s1, tag='', z_index=None):
    """Unconditional comparison of a forward (``res`` from attributed_parity) with the UNFORCED fp64 result: the
    reference's golden output or the plain oracle.  ``ref_Z`` may be a subsample: ``z_index(Z) -> array`` then picks
    the same entries out of the HIP Z.  Asserts, whatever the top-k selections did:
      * matches0 / matches1 bit-identical (the north star's argmax bar);
      * max|dZ| < PLAIN_MAX, fraction of entries beyond 1e-4 < PLAIN_FRAC, the same two bounds on the matching scores;
      * the literal 1e-4 bar on Z and the matching scores of every pair in which no top-k selection differs.
    Returns (number of pairs, pairs meeting the literal 1e-4 bar on Z)."""
    import numpy as np
    m0, m1, s0, s1, Z = res['out']
    Zc = Z.cpu().double().numpy()
    mine = z_index(Zc) if z_index is not None else Zc
    ref_Z = np.asarray(ref_Z)
    B = Zc.shape[0]
    err = np.abs(mine - ref_Z).reshape(B, -1)
    es0 = np.abs(s0.cpu().double().numpy() - np.asarray(ref_s0))
    es1 = np.abs(s1.cpu().double().numpy() - np.asarray(ref_s1))
    mm = int((m0.cpu().numpy() != np.asarray(ref_m0)).sum() + (m1.cpu().numpy() != np.asarray(ref_m1)).sum())
    literal = int((err.max(1) < Z_TOL).sum())
    print(f'[parity] {tag} vs the UNFORCED fp64 result: matches differing {mm}; max|dZ| {err.max():.3e}, entries beyond 1e-4 '
          f'{(err > Z_TOL).mean():.2e}; mscores {max(es0.max(), es1.max()):.3e}; pairs within the literal 1e-4: {literal}/{B} '
          f'(pairs without a flipped selection: {int((res["flips_per_pair"] == 0).sum())}/{B})')
    assert mm == 0, f'{mm} matches differ from the reference'
    assert err.max() < PLAIN_MAX and (err > Z_TOL).mean() < PLAIN_FRAC, (err.max(), (err > Z_TOL).mean())
    assert max(es0.max(), es1.max()) < PLAIN_MAX
    assert (np.concatenate([es0.ravel(), es1.ravel()]) > Z_TOL).mean() < 10 * PLAIN_FRAC
    for b in range(B):
        if int(res['flips_per_pair'][b]) == 0:       # same selections as fp64: the plain bar applies to this pair
            assert err[b].max() < Z_TOL, (b, err[b].max())
            assert es0[b].max() < Z_TOL and es1[b].max() < Z_TOL
    return B, literal


def assert_plain_vs_oracle(res, cfg, sd, data_cpu, tag=''):
    """assert_plain against the oracle run WITHOUT forced selections (the fp64 reference restated, pinned to the
    reference's goldens by tests/test_oracle_golden.py)."""
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data_cpu, cap)
    return assert_plain(res, cap['Z'].numpy(), ref['matches0'].numpy(), ref['matches1'].numpy(),
                        ref['matching_scores0'].numpy(), ref['matching_scores1'].numpy(), tag)


def local_flips(net, sd, data_cpu, device='cuda:0'):
    """{dynamic layer: rows whose HIP selection differs from the fp64 top-k of the HIP path's OWN fp32 layer input}: the flips
    caused INSIDE the layer (q / k projection + q.k products) of the fp32-class path - the rest arrive with the layer's input
    (tools/parity_report.py)."""
    dev = {k: v.to(device) for k, v in data_cpu.items()}
    k0 = dev['keypoints0']
    B, N, M = k0.shape[0], k0.shape[1], dev['keypoints1'].shape[1]
    sched = net._topk_schedule()
    L2 = len(sched)
    words = ops.topk_sel_words(B, N, M)
    sel = torch.zeros(L2 * words, dtype=torch.int32, device=k0.device)
    xl = torch.empty(L2, B, N + M, 128, device=k0.device)
    xe = torch.empty(B, N + M, 128, device=k0.device)
    net._run(k0, dev['scores0'], dev['descriptors0'], dev['keypoints1'], dev['scores1'], dev['descriptors1'], want_Z=False,
             taps={'topk_sel': sel, 'x_layers': xl, 'x_enc': xe})
    torch.cuda.synchronize()
    xl = torch.cat([xl, xe[None]]).cpu().double()          # index -1 = the encoder output = the input of layer 0
    out = {}
    for i, kk in enumerate(sched):
        if kk <= 0:
            continue
        masks = ops.topk_sel_to_masks(sel[i * words:(i + 1) * words], B, N, M, cross=bool(i & 1))
        rows = 0
        for side in range(2):
            x = (xl[i - 1][:, :N] if side == 0 else xl[i - 1][:, N:]).transpose(1, 2)      # [B, 128, n]: the HIP path's input
            other = (xl[i - 1][:, N:] if side == 0 else xl[i - 1][:, :N]).transpose(1, 2)
            src = other if (i & 1) else x
            p = f'gnn.layers.{i}.attn'
            q = O._pointwise(sd[f'{p}.proj.0.weight'], sd[f'{p}.proj.0.bias'], x).view(B, 32, 4, -1)
            kk_ = O._pointwise(sd[f'{p}.proj.1.weight'], sd[f'{p}.proj.1.bias'], src).view(B, 32, 4, -1)
            logits = torch.einsum('bdhn,bdhm->bhnm', q, kk_) / 32 ** 0.5
            own = torch.zeros_like(logits, dtype=torch.bool).scatter_(3, logits.topk(kk, dim=3).indices, True)
            rows += int((own ^ masks[side].cpu()).any(-1).sum())
        out[i] = rows
    return out
