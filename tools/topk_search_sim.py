"""CPU simulation of the threshold search of the dynamic attention kernels (attention.hip: topk_threshold, EXACT form) on the
logits of the bench inputs' dynamic layers (fp64 oracle, rounded to fp32), 16 rows in lockstep like a wave:
    python tools/topk_search_sim.py [pairs] [variant ...]
Prints, per variant, the counting passes per wave (probes while any of its 16 rows is still probing + the finishing pass) and
the distribution of probes per row.  Used to judge changes to the steering of the search before building them."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdgat_matcher_amd import synth  # noqa: E402
from oracle import mdgat_oracle as O  # noqa: E402

F = np.float32


def dynamic_layer_logits(pairs=1, n=512, L=9):
    """[(k, logits [rows, n] fp32)] for every dynamic layer and frame of `pairs` bench pairs; rows ordered (pair, head, query)."""
    cfg = synth.default_config(L=L)
    sd = {k: v.double() for k, v in synth.make_state_dict(L=L, seed=0).items()}
    data = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in synth.make_batch(pairs, n, n).items()}
    cap = {}
    O.mdgat_forward(sd, cfg, data, capture=cap)
    sched = O.layer_topk_schedule(L, cfg['k'])
    out = []
    for i, k in enumerate(sched):
        if k is None:
            continue
        prev = (cap[f'layer{i - 1}_desc0'], cap[f'layer{i - 1}_desc1'])
        for side in range(2):
            x = prev[side]
            src = prev[1 - side] if i % 2 else prev[side]
            p = f'gnn.layers.{i}.attn'
            q = O._pointwise(sd[f'{p}.proj.0.weight'], sd[f'{p}.proj.0.bias'], x).view(pairs, 32, 4, -1)
            kk = O._pointwise(sd[f'{p}.proj.1.weight'], sd[f'{p}.proj.1.bias'], src).view(pairs, 32, 4, -1)
            lg = torch.einsum('bdhn,bdhm->bhnm', q, kk) * (math.log2(math.e) / 32 ** 0.5)
            out.append((k, lg.reshape(-1, lg.shape[-1]).numpy().astype(F)))
    return out


def ordinal(f):
    b = f.view(np.int32).astype(np.int64)
    return np.where(b < 0, b ^ 0x7fffffff, b)


def from_ordinal(o):
    o = o.astype(np.int64)
    b = np.where(o < 0, o ^ 0x7fffffff, o).astype(np.int32)
    return b.view(F)


def search(S, k, variant='shipped'):
    """Returns (passes per wave [waves], probes per row [rows], thresholds [rows])."""
    rows, nk = S.shape
    zq = F(-torch.distributions.Normal(0, 1).icdf(torch.tensor(k / nk, dtype=torch.float64)).item())
    m = S.max(1)
    sub = S[:, ::4]
    mu = (sub.sum(1, dtype=F) * F(4.0 / nk)).astype(F)
    sd = np.sqrt(np.maximum((sub * sub).sum(1, dtype=F) * F(4.0 / nk) - mu * mu, F(1e-12))).astype(F)
    inv_sd = (F(1) / sd).astype(F)
    lo, hv = S.min(1).copy(), m.copy()
    clo = np.full(rows, nk, np.int64)
    chi = np.ones(rows, np.int64)
    state = np.zeros(rows, np.int64)
    hv_est = np.ones(rows, bool)
    lo_meas = np.zeros(rows, bool)
    thr = np.full(rows, -np.inf, F)
    state[k - chi == 1] = 2
    t = (mu + zq * sd).astype(F)
    if variant == 'cornish':
        # third-moment correction of the first probe (Cornish-Fisher)
        z3 = (((sub - mu[:, None]) * inv_sd[:, None]) ** 3).mean(1).astype(F)
        t = (mu + (zq + (zq * zq - 1) * z3 / 6) * sd).astype(F)
    nprobe = np.zeros(rows, np.int64)
    c_prev, t_prev = np.full(rows, nk, np.int64), lo.copy()
    waves = rows // 16
    passes = np.zeros(waves, np.int64)
    cap = int(variant[5:]) if variant.startswith('defer') else 80
    for it in range(80):
        if it >= cap:
            # straggler deferral: after `cap` probes the wave stops; rows still probing go to a list that a
            # repair.hip-style launch finishes one row per wave (state 9)
            state[state == 0] = 9
            break
        probing = state == 0
        wave_active = probing.reshape(waves, 16).any(1)
        if not wave_active.any():
            break
        if it >= 9 and (it & 1):
            ol, oh = ordinal(lo), ordinal(hv)
            t = from_ordinal(ol + ((oh - ol) >> 1))
        bad = ~((t > lo) & (t < hv))
        with np.errstate(all='ignore'):
            ti = (lo + (hv - lo) * ((clo - k).astype(F) + F(0.5)) / (clo - chi).astype(F)).astype(F)
        t = np.where(bad, ti, t)
        bad = ~((t > lo) & (t < hv))
        t = np.where(bad, (F(0.5) * lo + F(0.5) * hv).astype(F), t)
        collapsed = ~((t > lo) & (t < hv))
        t = np.where(collapsed & hv_est, hv, t)
        c = (S >= t[:, None]).sum(1)
        passes += wave_active
        nprobe += probing
        p = probing
        # collapsed
        sel = p & collapsed
        thr[sel] = np.where(hv_est[sel] & (c[sel] >= k), hv[sel], lo[sel])
        state[sel] = 1
        sel = p & ~collapsed & (c == k)
        thr[sel] = t[sel]
        state[sel] = 1
        upd = p & ~collapsed & (c != k)
        up = upd & (c > k)
        dn = upd & (c < k)
        lo[up], clo[up], lo_meas[up] = t[up], c[up], True
        hv[dn], chi[dn], hv_est[dn] = t[dn], c[dn], False
        fin = upd & (k - chi == 1)
        state[fin] = 2
        if variant in ('finish2', 'finish2lo'):
            fin2 = upd & ~fin & (k - chi == 2) & ~hv_est
            state[fin2] = 4
            fin = fin | fin2
        if variant == 'finish2lo':
            fin3 = upd & ~fin & (clo - k == 1)
            state[fin3] = 5
            fin = fin | fin3
        go = upd & ~fin
        z = (t - mu) * inv_sd
        dens = F(nk) * F(0.3989422804) * inv_sd * np.exp2(F(-0.7213475204) * z * z)
        tn = (t + (c - k).astype(F) / np.maximum(dens, F(1e-3) * F(nk) * inv_sd)).astype(F)
        narrow = lo_meas & ~hv_est & (clo - chi <= 48)
        if variant.startswith('bias'):
            # aim between the two terminal counts k and k - 1 (bias = 0.5), or further
            bias = F(float(variant[4:]))
            tn = (t + ((c - k).astype(F) + bias) / np.maximum(dens, F(1e-3) * F(nk) * inv_sd)).astype(F)
        if variant.startswith('last'):
            # density from the last two probes when they are close (secant through the two most recent counts)
            bias = F(float(variant[4:]))
            with np.errstate(all='ignore'):
                dl = (c_prev - c).astype(F) / (t - t_prev)
            ok = (it > 0) & np.isfinite(dl) & (dl > 0) & (np.abs(c_prev - c) <= 64) & (c_prev != c)
            d = np.where(ok, dl, np.maximum(dens, F(1e-3) * F(nk) * inv_sd))
            tn = (t + ((c - k).astype(F) + bias) / d).astype(F)
        c_prev, t_prev = np.where(probing, c, c_prev), np.where(probing, t, t_prev)
        if variant == 'secant':
            # local density from the bracket once both ends are measured
            both = lo_meas & ~hv_est
            with np.errstate(all='ignore'):
                d2 = (clo - chi).astype(F) / (hv - lo)
            tn = np.where(both, (t + (c - k).astype(F) / d2).astype(F), tn)
            narrow = np.zeros(rows, bool)
        nxt = np.where(((it & 3) == 3) | narrow, lo, tn)
        t = np.where(go, nxt, t).astype(F)
    # finishing passes: state 2 = one pass; 4 = two; 5 = three
    done = state == 1
    assert ((S[done] >= thr[done][:, None]).sum(1) >= k).all()
    search.deferred = int((state == 9).sum())
    st = state.reshape(waves, 16)
    passes += np.maximum((st == 2).any(1) * 1, (st == 4).any(1) * 2) + (st == 5).any(1) * 3
    search.last_probe_passes = passes - np.maximum((st == 2).any(1) * 1, (st == 4).any(1) * 2) - (st == 5).any(1) * 3
    return passes, nprobe, thr


if __name__ == '__main__':
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    variants = sys.argv[2:] or ['shipped']
    torch.set_num_threads(8)
    layers = dynamic_layer_logits(pairs)
    for v in variants:
        tot, n = 0.0, 0
        hist = np.zeros(16, np.int64)
        deferred = rows_all = 0
        for k, S in layers:
            passes, nprobe, _ = search(S, k, v)
            deferred += getattr(search, 'deferred', 0)
            rows_all += S.shape[0]
            tot += passes.sum()
            n += len(passes)
            hist += np.bincount(np.minimum(nprobe, 15), minlength=16)
        if v.startswith('defer'):
            print(f'   deferred rows: {deferred} of {rows_all} = {deferred / rows_all:.4f}')
        print(f'{v}: (last layer: probes per wave {search.last_probe_passes.mean():.2f}) {tot / n:.2f} counting passes per wave over {n} waves; probes per row: mean {np.dot(hist, np.arange(16)) / hist.sum():.2f}, histogram {hist.tolist()}')
