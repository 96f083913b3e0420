"""Host-side weight packer: reference state dict -> one flat fp32 blob in the layout
``mdgat_blob_layout`` (csrc/api.hip) expects.

All folding is done in fp64 and rounded to fp32 once:

* eval-mode BatchNorm1d (``mdgat.py:43``) is folded into the preceding Conv1d(k=1):
  ``W' = diag(g / sqrt(var + eps)) W``, ``b' = (b - mean) g / sqrt(var + eps) + beta``;
* the three projections of ``MultiHeadedAttention`` (``mdgat.py:221, 227-232``) are stacked into one
  [384 x 128] matrix whose rows are re-ordered from the reference's interleaved head layout
  (``view(B, 32, 4, N)``: channel c -> dim c // 4, head c % 4) to head-major (which, head, dim);
* ``merge`` (``mdgat.py:220, 237``) has no non-linearity before ``mlp.0`` (``mdgat.py:248``), so it is
  folded into the message half of ``mlp.0``: ``W1 [x ; Wm msg + bm] = W1x x + (W1m Wm) msg + W1m bm``
  (with Wm's input columns permuted to the head-major message layout);
* GAUGE FIXING.  The network's function does not change when a hidden channel is scaled by s > 0 and the weights that
  read it by 1 / s - behind a ReLU (positively homogeneous), between v and ``merge`` (linear), and, dimension by
  dimension, between q and k (the logits are bilinear).  A checkpoint may sit anywhere in that family; the kernels carry
  every operand as two f16 halves (|value| < 65504, values below ~10^-3 lose bits to the f16 denormal floor).  Each such
  channel that is 32x or more away from weight-row norm 1 (q against k: from equal norms) is therefore brought there by a
  power of two - exact in binary floating point, so the packed network computes the same numbers as the checkpoint's -
  before the blob is rounded
  (``tools/fuzz_checkpoint.py``: rescalings by 10^-3 ... 10^3 leave the results unchanged);
* the last layers of the two encoders are summed (``mdgat.py:392-393``) by concatenating them along K:
  ``[denc.6 | kenc.9] [hd ; hk]``.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

BN_EPS = 1e-5
D = 128
H = 4
DH = 32


def _round4(n: int) -> int:
    return (n + 3) & ~3


def blob_layout(L: int) -> Dict[str, int]:
    """Offsets (in floats) - the Python twin of mdgat_blob_layout() in csrc/api.hip."""
    lay = {}
    o = 0

    def take(name, n):
        nonlocal o
        lay[name] = o
        o += _round4(n)
    take('kenc0_w', 32 * 4); take('kenc0_b', 32)
    take('denc0_w', 64 * 33); take('denc0_b', 64)
    take('kenc1_w', 64 * 32); take('kenc1_b', 64)
    take('kenc2_w', 128 * 64); take('kenc2_b', 128)
    take('denc1_w', 128 * 64); take('denc1_b', 128)
    take('encl_w', 128 * 256); take('encl_b', 128)
    lay['layer0'] = o
    lo = 0
    for name, n in (('qkv_w', 384 * 128), ('qkv_b', 384), ('mlp1_w', 256 * 256), ('mlp1_b', 256),
                    ('mlp2_w', 128 * 256), ('mlp2_b', 128)):
        lay[name] = lo
        lo += _round4(n)
    lay['layer_stride'] = lo
    o += lo * 2 * L
    take('final_w', 128 * 128); take('final_b', 128)
    take('bin_score', 1)
    lay['total'] = o
    return lay


def strip_module_prefix(sd):
    """Checkpoints saved from DataParallel carry a ``module.`` prefix (test.py:158-159)."""
    if any(k.startswith('module.') for k in sd):
        return {k[len('module.'):] if k.startswith('module.') else k: v for k, v in sd.items()}
    return sd


def _np(t):
    return t.detach().to('cpu', torch.float64).numpy()


def _fold_bn(sd, conv, bn):
    w = _np(sd[f'{conv}.weight'])[:, :, 0]
    b = _np(sd[f'{conv}.bias'])
    g = _np(sd[f'{bn}.weight']) / np.sqrt(_np(sd[f'{bn}.running_var']) + BN_EPS)
    return w * g[:, None], (b - _np(sd[f'{bn}.running_mean'])) * g + _np(sd[f'{bn}.bias'])


def _plain(sd, conv):
    return _np(sd[f'{conv}.weight'])[:, :, 0], _np(sd[f'{conv}.bias'])


# head-major index h*32 + d  <-  reference channel d*4 + h
HEAD_MAJOR = np.array([d * H + h for h in range(H) for d in range(DH)])


GAUGE_SLACK = 5         # a channel is rescaled only when it is 2^5 = 32x or more away from unit scale


def _pow2(x):
    """Nearest power of two of every (positive, finite) entry - 1 where the entry is zero, not finite or within 2^-GAUGE_SLACK ...
    2^GAUGE_SLACK of one: the f16 halves span nine orders of magnitude, a channel a few octaves off unit scale is as good
    as one at it, and leaving it alone keeps the packed weights of an ordinary checkpoint the plain fp32 roundings."""
    x = np.asarray(x, dtype=np.float64)
    ok = np.isfinite(x) & (x > 0)
    e = np.round(np.log2(np.where(ok, x, 1.0)))
    return np.where(ok & (np.abs(e) >= GAUGE_SLACK), np.exp2(e), 1.0)


def _row_scale(w, b=None):
    """Power-of-two scale per output channel that brings the channel's weight row (and bias) to norm ~1."""
    n2 = (w ** 2).sum(axis=1) + (0.0 if b is None else b ** 2)
    return _pow2(1.0 / np.sqrt(np.where(n2 > 0, n2, 1.0)))


def _fix_hidden(w, b, w_next, cols=None):
    """Gauge of a hidden layer: rows of (w, b) x s, the columns ``cols`` of ``w_next`` that read the layer / s."""
    s = _row_scale(w, b)
    cols = slice(None) if cols is None else cols
    w_next = w_next.copy()
    w_next[:, cols] = w_next[:, cols] / s[None, :]
    return w * s[:, None], b * s, w_next


def check_supported(sd, L):
    shapes = {
        'kenc.encoder.0.weight': (32, 4, 1), 'kenc.encoder.3.weight': (64, 32, 1),
        'kenc.encoder.6.weight': (128, 64, 1), 'kenc.encoder.9.weight': (128, 128, 1),
        'denc.encoder.0.weight': (64, 33, 1), 'denc.encoder.3.weight': (128, 64, 1),
        'denc.encoder.6.weight': (128, 128, 1), 'final_proj.weight': (128, 128, 1),
    }
    for k, shp in shapes.items():
        if k not in sd:
            raise KeyError(f'state dict lacks {k} (only descriptor="FPFH" checkpoints are supported)')
        if tuple(sd[k].shape) != shp:
            raise ValueError(f'{k} has shape {tuple(sd[k].shape)}; the HIP path implements the default widths {shp}')
    if f'gnn.layers.{2 * L - 1}.mlp.3.weight' not in sd or f'gnn.layers.{2 * L}.mlp.3.weight' in sd:
        raise ValueError(f'state dict does not hold exactly 2L={2 * L} GNN layers')


def pack_state_dict(sd, L: int, dtype=np.float32) -> np.ndarray:
    """The packed blob.  ``dtype=np.float64`` returns the folded weights BEFORE their rounding to fp32 - what the library's
    reference-exact mode loads next to the fp32 blob (``mdgat_load_weights_f64``; same layout)."""
    sd = strip_module_prefix(sd)
    check_supported(sd, L)
    lay = blob_layout(L)
    blob = np.zeros(lay['total'], dtype=np.float64)

    def put(off, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        blob[off:off + a.size] = a

    # encoders: BN folded, then the gauge of every hidden layer fixed (module docstring)
    wk0, bk0 = _fold_bn(sd, 'kenc.encoder.0', 'kenc.encoder.1')
    wk1, bk1 = _fold_bn(sd, 'kenc.encoder.3', 'kenc.encoder.4')
    wk2, bk2 = _fold_bn(sd, 'kenc.encoder.6', 'kenc.encoder.7')
    wk3, bk3 = _plain(sd, 'kenc.encoder.9')
    wd0, bd0 = _fold_bn(sd, 'denc.encoder.0', 'denc.encoder.1')
    wd1, bd1 = _fold_bn(sd, 'denc.encoder.3', 'denc.encoder.4')
    wd2, bd2 = _plain(sd, 'denc.encoder.6')
    wk0, bk0, wk1 = _fix_hidden(wk0, bk0, wk1)
    wk1, bk1, wk2 = _fix_hidden(wk1, bk1, wk2)
    wk2, bk2, wk3 = _fix_hidden(wk2, bk2, wk3)
    wd0, bd0, wd1 = _fix_hidden(wd0, bd0, wd1)
    wd1, bd1, wd2 = _fix_hidden(wd1, bd1, wd2)
    put(lay['kenc0_w'], wk0); put(lay['kenc0_b'], bk0)
    put(lay['denc0_w'], wd0); put(lay['denc0_b'], bd0)
    put(lay['kenc1_w'], wk1); put(lay['kenc1_b'], bk1)
    put(lay['kenc2_w'], wk2); put(lay['kenc2_b'], bk2)
    put(lay['denc1_w'], wd1); put(lay['denc1_b'], bd1)
    put(lay['encl_w'], np.concatenate([wd2, wk3], axis=1)); put(lay['encl_b'], bd2 + bk3)

    for i in range(2 * L):
        base = lay['layer0'] + i * lay['layer_stride']
        p = f'gnn.layers.{i}'
        ws, bs = [], []
        for j in range(3):
            w, b = _plain(sd, f'{p}.attn.proj.{j}')
            ws.append(w[HEAD_MAJOR]); bs.append(b[HEAD_MAJOR])
        # q against k, dimension by dimension: equal row norms (the logits see the product of the two)
        nq = np.sqrt((ws[0] ** 2).sum(axis=1) + bs[0] ** 2)
        nk = np.sqrt((ws[1] ** 2).sum(axis=1) + bs[1] ** 2)
        sq = _pow2(np.sqrt(np.where((nq > 0) & (nk > 0), nk / np.where(nq > 0, nq, 1.0), 1.0)))
        ws[0], bs[0] = ws[0] * sq[:, None], bs[0] * sq
        ws[1], bs[1] = ws[1] / sq[:, None], bs[1] / sq
        # v against merge, channel by channel
        sv = _row_scale(ws[2], bs[2])
        ws[2], bs[2] = ws[2] * sv[:, None], bs[2] * sv
        put(base + lay['qkv_w'], np.concatenate(ws, axis=0)); put(base + lay['qkv_b'], np.concatenate(bs))
        wm, bm = _plain(sd, f'{p}.attn.merge')
        wm = wm[:, HEAD_MAJOR] / sv[None, :]       # input columns in head-major message order, the v gauge undone
        w1, b1 = _fold_bn(sd, f'{p}.mlp.0', f'{p}.mlp.1')
        w1x, w1m = w1[:, :D], w1[:, D:]
        w1f, b1f = np.concatenate([w1x, w1m @ wm], axis=1), b1 + w1m @ bm
        w2, b2 = _plain(sd, f'{p}.mlp.3')
        w1f, b1f, w2 = _fix_hidden(w1f, b1f, w2)
        put(base + lay['mlp1_w'], w1f); put(base + lay['mlp1_b'], b1f)
        put(base + lay['mlp2_w'], w2); put(base + lay['mlp2_b'], b2)
    w, b = _plain(sd, 'final_proj'); put(lay['final_w'], w); put(lay['final_b'], b)
    blob[lay['bin_score']] = float(_np(sd['bin_score']))
    return blob.astype(dtype)


def resolve_topk_schedule(L: int, k_list: List[Optional[int]]) -> List[int]:
    """Per-layer k of AttentionalGNN.forward (mdgat.py:268-272); 0 stands for full attention."""
    n = len(k_list)
    sched = []
    for i in range(2 * L):
        k = k_list[i - 2 * L + n] if i > 2 * L - 1 - n else None
        sched.append(0 if k is None else int(k))
    return sched
