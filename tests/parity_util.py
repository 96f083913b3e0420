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
"""
import torch

from mdgat_matcher_amd import ops
from oracle import mdgat_oracle as O

Z_TOL = 1e-4
# Largest |logit - k-th logit| (natural units of q.k / sqrt(32), logits are O(10)) a disagreeing key may have: the
# accumulated fp32-class error of the logits after up to 17 layers (measured worst case 7.5e-6: profiles/parity_r2.txt).
GAP_EPS = 2e-5


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
    cap = {}
    ref = O.mdgat_forward(sd, cfg, data_cpu, cap, forced_topk=forced)
    rows = total = bad = 0
    max_gap = 0.0
    for reps in cap.get('topk_report', {}).values():
        for r in reps:
            rows += r['rows']
            total += r['total_rows']
            bad += r['bad_count']
            max_gap = max(max_gap, r['max_gap'])
    return {
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
