#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (fp64, CPU) in the build container.

Only runs where ``/root/reference`` exists (never on the GPU box).  ``models/mdgat.py`` is imported
unmodified; because it hard-codes ``torch.device('cuda')`` (mdgat.py:25, 200, 466-467, 473-474)
the module-level name ``torch`` *inside the imported module* is replaced by a thin proxy whose
``device(...)`` answers CPU and whose ``zeros``/``zeros_like``/``arange`` drop a cuda ``device=``.
No reference file is edited or copied; only inputs (by seed) and outputs are written.

Fixtures (fp64 ``.npz``; weights are NOT stored - they are regenerated from the seed by
``mdgat_matcher_amd.synth.make_state_dict``):

* ``fwd_*``      full forward, tiny shapes: every stage tensor (encoder out, each layer, final
                 projection, pre-OT scores, Z) + matches/mscores for the 4 extraction variants.
* ``cfg_*``      BASELINE config shapes (N=256 L=4 S=20; N=512 L=9 S=100): matches, mscores,
                 Z sub-sampled (every 8th row/col + dustbin row/col) and per-row LSE checksums.
* ``op_*``       stand-alone ops: Sinkhorn, attention vs dynamic_attention, knn.
* ``edge_*``     empty keypoints early-out, k == M, all-dustbin frame.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get('MDGAT_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')

from mdgat_matcher_amd import synth  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    import models.mdgat as M  # the reference, unmodified
    cpu = torch.device('cpu')

    class _TorchProxy(types.ModuleType):
        def __getattr__(self, name):
            return getattr(torch, name)

    proxy = _TorchProxy('torch_cpu_proxy')
    proxy.device = lambda *a, **k: cpu

    def _strip(fn):
        def inner(*a, **k):
            if 'device' in k:
                k['device'] = cpu
            return fn(*a, **k)
        return inner
    for name in ('zeros', 'zeros_like', 'arange', 'ones', 'ones_like'):
        setattr(proxy, name, _strip(getattr(torch, name)))
    M.torch = proxy
    return M


def build_ref_net(M, cfg, sd):
    net = M.MDGAT(cfg).double().eval()
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net


def run_ref(M, net, data, capture=True):
    """Forward with hooks capturing stage tensors."""
    cap = {}
    hooks = []
    if capture:
        hooks.append(net.kenc.register_forward_hook(lambda m, i, o: cap.setdefault('kenc', []).append(o.detach().clone())))
        hooks.append(net.denc.register_forward_hook(lambda m, i, o: cap.setdefault('denc', []).append(o.detach().clone())))
        hooks.append(net.final_proj.register_forward_hook(lambda m, i, o: cap.setdefault('mdesc', []).append(o.detach().clone())))
        for li, layer in enumerate(net.gnn.layers):
            hooks.append(layer.register_forward_hook(
                lambda m, i, o, li=li: cap.setdefault(f'delta{li}', []).append((i[0].detach().clone(), o.detach().clone()))))
    orig_lot = M.log_optimal_transport

    def lot(scores, alpha, iters):
        cap['scores'] = scores.detach().clone()
        Z = orig_lot(scores, alpha, iters)
        cap['Z'] = Z.detach().clone()
        return Z
    M.log_optimal_transport = lot
    try:
        with torch.no_grad():
            d = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
            out = net(d)
    finally:
        M.log_optimal_transport = orig_lot
        for h in hooks:
            h.remove()
    return out, cap


def stage_dict(cap, L):
    st = {}
    st['enc0'] = (cap['denc'][0] + cap['kenc'][0]).numpy()
    st['enc1'] = (cap['denc'][1] + cap['kenc'][1]).numpy()
    for li in range(2 * L):
        (x0, d0), (x1, d1) = cap[f'delta{li}']
        st[f'layer{li}_desc0'] = (x0 + d0).numpy()
        st[f'layer{li}_desc1'] = (x1 + d1).numpy()
    st['mdesc0'], st['mdesc1'] = cap['mdesc'][0].numpy(), cap['mdesc'][1].numpy()
    st['scores'] = cap['scores'].numpy()
    st['Z'] = cap['Z'].numpy()
    return st


def out_arrays(out, tag):
    return {
        f'{tag}_matches0': out['matches0'].numpy().astype(np.int64),
        f'{tag}_matches1': out['matches1'].numpy().astype(np.int64),
        f'{tag}_mscores0': out['matching_scores0'].numpy().astype(np.float64),
        f'{tag}_mscores1': out['matching_scores1'].numpy().astype(np.float64),
    }


VARIANTS = {  # tag -> (loss_method, mutual_check)
    'default': ('triplet_loss', False),
    'mutual': ('triplet_loss', True),
    'sg': ('superglue', False),
    'sgmutual': ('superglue', True),
}


def gen_forward(M, name, B, n, m, L, S, k, seed=0, bin_score=1.0, first_pair=0):
    sd = synth.make_state_dict(L=L, seed=seed, bin_score=bin_score)
    data = synth.make_batch(B, n, m, first_pair=first_pair)
    arrays = {'meta': np.array([B, n, m, L, S, seed, first_pair], dtype=np.int64),
              'k': np.array([-1 if x is None else x for x in k], dtype=np.int64),
              'bin_score': np.array(bin_score)}
    for tag, (loss_method, mutual) in VARIANTS.items():
        if mutual and B != 1:
            continue  # reference's mutual branch of 469-478 only works for batch 1 (mask shape)
        if n != m:
            # the reference's triplet and superglue LOSS code (mdgat.py:494-539) raises a shape
            # error when N != M; 'gap_loss' shares the default extraction branch (459-483) and
            # its loss runs for ragged pairs, so ragged fixtures use it and skip 'superglue'.
            if loss_method == 'superglue':
                continue
            loss_method = 'gap_loss'
        cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, loss_method=loss_method, mutual_check=mutual)
        net = build_ref_net(M, cfg, sd)
        # gt_matches are all -1 (synth.make_batch): every loss branch (487-594) then runs without
        # an out-of-range index; the loss value itself is not captured (training-only, out of scope).
        out, cap = run_ref(M, net, data, capture=(tag == 'default'))
        if tag == 'default':
            arrays.update(stage_dict(cap, L))
        arrays.update(out_arrays(out, tag))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print('wrote', name, {k2: v.shape for k2, v in arrays.items() if k2 in ('Z', 'scores')})


def gen_config(M, name, B, n, m, L, S, k, seed=0, first_pair=0, bin_score=1.0, sub=8):
    sd = synth.make_state_dict(L=L, seed=seed, bin_score=bin_score)
    data = synth.make_batch(B, n, m, first_pair=first_pair)
    cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S)
    net = build_ref_net(M, cfg, sd)
    out, cap = run_ref(M, net, data, capture=False)
    Z = cap['Z'].numpy()
    arrays = {'meta': np.array([B, n, m, L, S, seed, first_pair], dtype=np.int64),
              'k': np.array([-1 if x is None else x for x in k], dtype=np.int64), 'bin_score': np.array(bin_score)}
    arrays.update(out_arrays(out, 'default'))
    if sub != 8:
        arrays['sub'] = np.array(sub)            # stride of the Z / score sub-samples (8 where the key is absent)
    arrays['Z_sub'] = Z[:, ::sub, ::sub].copy()
    arrays['Z_lastrow'] = Z[:, -1, :].copy()
    arrays['Z_lastcol'] = Z[:, :, -1].copy()
    arrays['Z_row_lse'] = torch.logsumexp(cap['Z'], dim=2).numpy()
    arrays['Z_col_lse'] = torch.logsumexp(cap['Z'], dim=1).numpy()
    arrays['scores_sub'] = cap['scores'].numpy()[:, ::sub, ::sub].copy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print('wrote', name, Z.shape)


def gen_config_variants(M, name, n, m, L, S, k, seed=0, first_pair=0):
    """One pair at a BASELINE-sized shape through every extraction branch the reference can run for it (mdgat.py:441-483):
    matches and scores per variant + the Z the default variant produced (sub-sampled).  Ragged pairs (N != M): the triplet /
    superglue LOSS code raises there, so 'gap_loss' stands for the default branch and the superglue variants are skipped."""
    sd = synth.make_state_dict(L=L, seed=seed)
    data = synth.make_batch(1, n, m, first_pair=first_pair)
    arrays = {'meta': np.array([1, n, m, L, S, seed, first_pair], dtype=np.int64),
              'k': np.array([-1 if x is None else x for x in k], dtype=np.int64)}
    for tag, (loss_method, mutual) in VARIANTS.items():
        if n != m:
            if loss_method == 'superglue':
                continue
            loss_method = 'gap_loss'
        cfg = synth.default_config(L=L, k=k, sinkhorn_iterations=S, loss_method=loss_method, mutual_check=mutual)
        out, cap = run_ref(M, build_ref_net(M, cfg, sd), data, capture=False)
        arrays.update(out_arrays(out, tag))
        if tag == 'default':
            Z = cap['Z'].numpy()
            arrays['Z_sub'] = Z[:, ::8, ::8].copy()
            arrays['Z_lastrow'] = Z[:, -1, :].copy()
            arrays['Z_lastcol'] = Z[:, :, -1].copy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print('wrote', name, [t for t in VARIANTS if f'{t}_matches0' in arrays])


def gen_ops(M):
    rs = np.random.RandomState(42)
    arrays = {}
    # Sinkhorn on random scores
    for (n, m, iters, alpha) in ((7, 5, 20, 1.0), (64, 64, 100, 0.37), (48, 64, 50, 1.0), (512, 512, 100, 1.0)):
        s = torch.from_numpy(rs.standard_normal((2, n, m)) * 3.0)
        Z = M.log_optimal_transport(s, torch.tensor(alpha, dtype=torch.float64), iters)
        tag = f'sk_{n}x{m}'
        if n == 512:
            arrays[tag + '_seed'] = np.array([4242])
            s = torch.from_numpy(np.random.RandomState(4242).standard_normal((1, n, m)) * 3.0)
            Z = M.log_optimal_transport(s, torch.tensor(alpha, dtype=torch.float64), iters)
            arrays[tag + '_Z_sub'] = Z.numpy()[:, ::8, ::8].copy()
            arrays[tag + '_Z_lastrow'] = Z.numpy()[:, -1, :].copy()
            arrays[tag + '_Z_lastcol'] = Z.numpy()[:, :, -1].copy()
        else:
            arrays[tag + '_scores'] = s.numpy()
            arrays[tag + '_Z'] = Z.numpy()
        arrays[tag + '_meta'] = np.array([iters, alpha])
    # attention / dynamic attention on random q/k/v, [B, dh, H, N]
    q = torch.from_numpy(rs.standard_normal((2, 32, 4, 40)) * 1.5)
    k = torch.from_numpy(rs.standard_normal((2, 32, 4, 56)) * 1.5)
    v = torch.from_numpy(rs.standard_normal((2, 32, 4, 56)))
    full, _ = M.attention(q, k, v)
    arrays.update(att_q=q.numpy(), att_k=k.numpy(), att_v=v.numpy(), att_full=full.numpy())
    for kk in (1, 8, 56):
        dyn, prob = M.dynamic_attention(q, k, v, kk)
        arrays[f'att_dyn{kk}'] = dyn.numpy()
        arrays[f'att_dyn{kk}_nnz'] = (prob > 0).sum(-1).numpy()
    # knn (dead code in the reference, named by north_star)
    for C in (3, 128):
        x = torch.from_numpy(rs.standard_normal((2, C, 50)))
        s = torch.from_numpy(rs.standard_normal((2, C, 70)))
        idx = M.knn(x, s, 9)
        A = M.get_graph_feature(x, s, 9)
        arrays[f'knn{C}_x'], arrays[f'knn{C}_s'] = x.numpy(), s.numpy()
        arrays[f'knn{C}_idx'], arrays[f'knn{C}_adj'] = idx.numpy(), A.numpy()
    np.savez_compressed(os.path.join(OUT, 'op_vectors.npz'), **arrays)
    print('wrote op_vectors')


def gen_edges(M):
    arrays = {}
    L = 1
    sd = synth.make_state_dict(L=L, seed=3)
    # empty keypoints early-out (mdgat.py:374-382)
    cfg = synth.default_config(L=L, k=[], sinkhorn_iterations=5)
    net = build_ref_net(M, cfg, sd)
    data = synth.make_batch(1, 8, 8)
    data['keypoints0'] = data['keypoints0'][:, :0]
    with torch.no_grad():
        out = net(data)
    arrays['empty_matches0'] = out['matches0'].numpy()
    arrays['empty_matches1'] = out['matches1'].numpy()
    arrays['empty_mscores0'] = out['matching_scores0'].numpy()
    arrays['empty_mscores1'] = out['matching_scores1'].numpy()
    arrays['empty_skip'] = np.array(out['skip_train'])
    # k == M: dynamic attention over all keys == full attention
    cfg_full = synth.default_config(L=L, k=[], sinkhorn_iterations=10)
    cfg_kM = synth.default_config(L=L, k=[32, 32], sinkhorn_iterations=10)
    data = synth.make_batch(1, 32, 32)
    o_full, c_full = run_ref(M, build_ref_net(M, cfg_full, sd), data, capture=False)
    o_kM, c_kM = run_ref(M, build_ref_net(M, cfg_kM, sd), data, capture=False)
    arrays['keqM_Z_full'] = c_full['Z'].numpy()
    arrays['keqM_Z_dyn'] = c_kM['Z'].numpy()
    # all-dustbin frame: a huge bin score sends every row to the dustbin (mdgat.py:465-467 quirk)
    sd_bin = synth.make_state_dict(L=L, seed=3, bin_score=50.0)
    o_bin, c_bin = run_ref(M, build_ref_net(M, cfg_full, sd_bin), data, capture=False)
    arrays['alldust_Z'] = c_bin['Z'].numpy()
    arrays['alldust_matches0'] = o_bin['matches0'].numpy()
    arrays['alldust_matches1'] = o_bin['matches1'].numpy()
    arrays['alldust_mscores0'] = o_bin['matching_scores0'].numpy().astype(np.float64)
    arrays['alldust_mscores1'] = o_bin['matching_scores1'].numpy().astype(np.float64)
    arrays['alldust_mscores_is_int'] = np.array(not o_bin['matching_scores0'].dtype.is_floating_point)
    np.savez_compressed(os.path.join(OUT, 'edge_cases.npz'), **arrays)
    print('wrote edge_cases', 'int-quirk:', arrays['alldust_mscores_is_int'])


CHECK_DEFAULT = ('fwd_n64_L1_S1', 'fwd_n64_L4_S20', 'fwd_n64_L5_S20', 'fwd_n48m64_L4_S20', 'op_vectors', 'edge_cases', 'cfg_n256_L4_S20')


def compare_dirs(fresh, committed, names):
    """Fixture guard: the files this script writes today against the committed ones - same keys, integers identical, floats to
    1e-12 (the reference's fp64 sums may reassociate with the thread count).  Returns the list of disagreements."""
    bad = []
    for nm in names:
        a, b = np.load(os.path.join(fresh, nm + '.npz')), np.load(os.path.join(committed, nm + '.npz'))
        if set(a.files) != set(b.files):
            bad.append(f'{nm}: keys differ: only regenerated {sorted(set(a.files) - set(b.files))}, only committed {sorted(set(b.files) - set(a.files))}')
        for key in sorted(set(a.files) & set(b.files)):
            x, y = a[key], b[key]
            if x.shape != y.shape or x.dtype != y.dtype:
                bad.append(f'{nm}[{key}]: {x.dtype}{x.shape} regenerated vs {y.dtype}{y.shape} committed')
            elif np.issubdtype(x.dtype, np.floating):
                fin = np.isfinite(x) & np.isfinite(y)
                if not np.array_equal(np.isfinite(x), np.isfinite(y)) or (fin.any() and np.abs(x[fin] - y[fin]).max() > 1e-12 * max(1.0, np.abs(y[fin]).max())):
                    bad.append(f'{nm}[{key}]: values differ by {np.abs(x[fin] - y[fin]).max():.3e}')
            elif not np.array_equal(x, y):
                bad.append(f'{nm}[{key}]: integer values differ')
    return bad


def main():
    """python tools/make_goldens.py [--check] [fixture name ...]  (no names: all of them; --check: regenerate into a scratch
    directory and compare with the committed files instead of overwriting them - default set: the small fixtures)"""
    global OUT
    check = '--check' in sys.argv[1:]
    sys.argv = [a for a in sys.argv if a != '--check']
    committed = OUT
    if check:
        import tempfile
        OUT = tempfile.mkdtemp(prefix='mdgat_goldens_')
        if len(sys.argv) == 1:
            sys.argv += list(CHECK_DEFAULT)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    M = import_reference()
    small_k = [16, None, 16, None, 8, None, 8, None]
    only = set(sys.argv[1:])
    jobs = [
        ('fwd_n64_L1_S1', lambda nm: gen_forward(M, nm, 1, 64, 64, 1, 1, small_k)),
        ('fwd_n64_L4_S20', lambda nm: gen_forward(M, nm, 1, 64, 64, 4, 20, small_k)),
        ('fwd_n64_L5_S20', lambda nm: gen_forward(M, nm, 2, 64, 64, 5, 20, small_k)),
        ('fwd_n48m64_L4_S20', lambda nm: gen_forward(M, nm, 1, 48, 64, 4, 20, small_k, bin_score=0.37, first_pair=5)),
        # BASELINE configs[0] / configs[1] shapes, 8 reference-held pairs each (round 3: the headline shape had 2)
        ('cfg_n256_L4_S20', lambda nm: gen_config(M, nm, 8, 256, 256, 4, 20, synth.DEFAULT_K)),
        ('cfg_n512_L9_S100', lambda nm: gen_config(M, nm, 8, 512, 512, 9, 100, synth.DEFAULT_K)),
        # BASELINE configs[4]: one pair (round 3) + three more (round 4; sub-sampled every 16th entry to stay small)
        ('cfg_n2048_L9_S200', lambda nm: gen_config(M, nm, 1, 2048, 2048, 9, 200, synth.DEFAULT_K)),
        ('cfg_n2048_L9_S200_b', lambda nm: gen_config(M, nm, 3, 2048, 2048, 9, 200, synth.DEFAULT_K, first_pair=1, sub=16)),
        # other weights, other bin score
        ('cfg_n512_L9_S100_seed7', lambda nm: gen_config(M, nm, 4, 512, 512, 9, 100, synth.DEFAULT_K, seed=7, first_pair=40, bin_score=0.37)),
        # a batch of the headline shape large enough to run in slices on two lanes in the library (40 pairs: 32 + 8), so that the kernels
        # big launches get (round 6: full attention with one wave per 32 queries) are pinned to the reference's outputs too
        ('cfg_n512_L9_S100_b40', lambda nm: gen_config(M, nm, 40, 512, 512, 9, 100, synth.DEFAULT_K, first_pair=200, sub=32)),
        # ... and one with other weights, another bin score and a keypoint count that is no multiple of the kernels' tiles (24 pairs of 400: one slice)
        ('cfg_n400_L9_S100_b24', lambda nm: gen_config(M, nm, 24, 400, 400, 9, 100, synth.DEFAULT_K, seed=7, first_pair=300, bin_score=0.37, sub=32)),
        # ... and configs[0]'s shape as a batch of 64 (every layer dynamic or cross at L = 4; weights seed 3)
        ('cfg_n256_L4_S20_b64', lambda nm: gen_config(M, nm, 64, 256, 256, 4, 20, synth.DEFAULT_K, seed=3, first_pair=500, sub=32)),
        ('var_n256_L4_S20', lambda nm: gen_config_variants(M, nm, 256, 256, 4, 20, synth.DEFAULT_K, first_pair=11)),
        ('var_n512_L9_S100', lambda nm: gen_config_variants(M, nm, 512, 512, 9, 100, synth.DEFAULT_K, first_pair=12)),
        ('var_n400m512_L9_S100', lambda nm: gen_config_variants(M, nm, 400, 512, 9, 100, synth.DEFAULT_K, first_pair=13)),
        ('op_vectors', lambda nm: gen_ops(M)),
        ('edge_cases', lambda nm: gen_edges(M)),
    ]
    unknown = only - {nm for nm, _ in jobs}
    assert not unknown, f'unknown fixtures {sorted(unknown)}'
    done = []
    for nm, fn in jobs:
        if not only or nm in only:
            fn(nm)
            done.append(nm)
    if check:
        bad = compare_dirs(OUT, committed, done)
        import shutil
        shutil.rmtree(OUT, ignore_errors=True)
        for line in bad:
            print('MISMATCH', line)
        print(f'checked {len(done)} fixtures against {committed}: ' + ('OK' if not bad else f'{len(bad)} disagreements'))
        sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
