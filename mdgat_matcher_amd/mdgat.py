"""Drop-in ``MDGAT`` module and ``match()`` API over the gfx950 HIP library.

Host-side mirror of ``/root/reference/models/mdgat.py:315-603`` (class ``MDGAT``) for the
``descriptor == 'FPFH'`` inference path:

* same constructor config dict (``test.py:137-151``), same parameter and buffer names and shapes, so
  ``load_state_dict(checkpoint['net'])`` works (also through ``torch.nn.DataParallel``, whose keys carry a
  ``module.`` prefix - ``test.py:158-159``);
* same ``forward(data: dict) -> dict`` contract: keys ``keypoints0/1, descriptors0/1, scores0/1`` in,
  ``matches0/1`` (int64, -1 = unmatched), ``matching_scores0/1`` (module dtype) and ``loss`` out, the
  empty-keypoint early-out of ``mdgat.py:374-382`` included;
* the module's dtype is the arithmetic request, as it is in the reference: ``net.double()`` (what ``test.py:193`` and
  ``test_registration_metric.py:194`` call before every forward) runs the library's reference-exact mode - fp64 inputs,
  fp64 weights and fp64 matrix-core arithmetic, so that every ``logits.topk(k)`` (``mdgat.py:202``) selects what the
  reference's fp64 run selects (``include/mdgat_hip.h``: ``MDGAT_ARITH_FP64``) and an fp64 Sinkhorn whose arg-maxes
  are the reference's (``config['sinkhorn_arithmetic']``) at every frame size the library takes; a float32 module runs the fp32-class throughput path (5x the rate, Z within
  1e-4 except around the ~1.5 keypoints per pair whose top-k near-tie falls the other way).  ``config['arithmetic']`` =
  ``'fp32'`` / ``'fp64'`` (not a reference key) or ``MDGAT_ARITHMETIC`` in the environment pin one path whatever the dtype;
  results are cast to the module's dtype either way.

All arithmetic happens in ``libmdgat_hip.so``; PyTorch only owns device memory and streams.  There is no
CPU path: tensors that are not on a gfx950 device raise.  Training (loss / backward) is out of scope:
``loss`` is returned as a zero scalar and ``train()`` mode raises in ``forward``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib, pack

_D = 128


def _mlp_modules(channels):
    """Parameter container with the reference's Sequential indices (conv 3i, BN 3i+1, ReLU 3i+2)."""
    mods = []
    last = len(channels) - 1
    for i in range(1, len(channels)):
        mods.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < last:
            mods.append(nn.BatchNorm1d(channels[i]))
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class _Encoder(nn.Module):
    def __init__(self, cin, hidden, cout):
        super().__init__()
        self.encoder = _mlp_modules([cin, *hidden, cout])
        nn.init.constant_(self.encoder[-1].bias, 0.0)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.merge = nn.Conv1d(d, d, kernel_size=1)
        self.proj = nn.ModuleList([nn.Conv1d(d, d, kernel_size=1) for _ in range(3)])


class _Propagation(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.attn = _Attn(d)
        self.mlp = _mlp_modules([2 * d, 2 * d, d])
        nn.init.constant_(self.mlp[-1].bias, 0.0)


class _GNN(nn.Module):
    def __init__(self, d, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([_Propagation(d) for _ in range(n_layers)])


class _DeviceState:
    """Per-device library handle + packed weights + scratch (one per (module, device)).  The scratch is per STREAM: the
    library only enqueues, so forwards issued on two streams run concurrently on the device and must not share it."""

    MAX_WORKSPACES = 8                  # streams remembered per device (least recently used first out)

    def __init__(self, handle, device, f64=False):
        self.handle = handle
        self.device = device
        self.f64 = f64                  # the handle computes in MDGAT_ARITH_FP64 (fixed at mdgat_create)
        self.workspaces = {}            # stream handle -> uint8 tensor, in order of last use
        self.lock = threading.Lock()

    def workspace_for(self, stream: int, need: int, dev):
        """Scratch of the forward being enqueued on ``stream``.  Cached per raw stream handle, bounded (PyTorch hands stream
        handles out of a pool: unbounded, the cache would pin ~0.4 GB per handle ever seen).  While a stream is being CAPTURED
        into a graph the scratch is allocated from the graph's private pool and must live exactly as long as the graph: it is
        handed back uncached (the graph keeps its allocation alive), and an eager call on a recycled handle can never be given
        memory a graph still replays into."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return torch.empty(need, dtype=torch.uint8, device=dev)
        ws = self.workspaces.pop(stream, None)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self.workspaces[stream] = ws    # (most recently used last)
        while len(self.workspaces) > self.MAX_WORKSPACES:
            self.workspaces.pop(next(iter(self.workspaces)))
        return ws

    def close(self):
        self.workspaces.clear()
        if self.handle:
            _lib.load().mdgat_destroy(self.handle)
            self.handle = None


def _sync_raw_stream(handle: int, dev):
    """Synchronise a stream known by its raw handle (the keys of _DeviceState.workspaces)."""
    torch.cuda.ExternalStream(handle, device=dev).synchronize()


def _close_states(states):
    for st in list(states.values()):
        st.close()
    states.clear()


class MDGAT(nn.Module):
    default_config = {
        'descriptor_dim': 128,
        'keypoint_encoder': [32, 64, 128],
        'descritor_encoder': [64, 128],
        'GNN_layers': ['self', 'cross'] * 9,
        'sinkhorn_iterations': 100,
        'match_threshold': 0.2,
    }

    def __init__(self, config):
        super().__init__()
        self.config = {**self.default_config, **config}
        # the reference reads these with config[...] and raises KeyError when absent (mdgat.py:329, 353, 362-367)
        self.descriptor = config['descriptor']
        self.lr = config['lr']
        self.loss_method = config['loss_method']
        self.k = config['k']
        self.mutual_check = config['mutual_check']
        self.triplet_loss_gamma = config['triplet_loss_gamma']
        self.train_step = config['train_step']
        L = self.config['L']
        # not a reference key: 'fp32' (default, the parity path) or 'f16' (single-f16 attention products, a throughput
        # mode outside the parity bar - BASELINE.json configs[2]; 'bf16' is accepted as an alias: the operands are f16,
        # same matrix-core rate, three more mantissa bits)
        self.attention_dtype = str(self.config.get('attention_dtype', 'fp32'))
        if self.attention_dtype not in ('fp32', 'f16', 'bf16'):
            raise ValueError(f"attention_dtype={self.attention_dtype!r}: expected 'fp32' or 'f16'")
        # not a reference key either: how the library executes a batch (include/mdgat_hip.h: mdgat_set_lanes).  0 = its default
        # (two lanes: the halves of a batch in flight on two streams); results do not depend on it
        self.lanes = int(self.config.get('lanes', 0))
        if self.lanes not in (0, 1, 2):
            raise ValueError(f'lanes={self.lanes}: expected 1 or 2 (0: library default)')
        # not a reference key: which arithmetic the forward runs in (include/mdgat_hip.h: mdgat_arithmetic).
        #   'auto' (default): the MODULE'S DTYPE decides, as it does in the reference - a float64 module (test.py:193 and
        #            test_registration_metric.py:194 call net.double() before every forward) runs the reference's own
        #            arithmetic for the encoders and the layers up to the last dynamic one, where the top-k selection of
        #            mdgat.py:202 is decided (csrc/f64.hip); a float32 module runs the fp32-class throughput path;
        #   'fp32' / 'fp64': that path whatever the dtype (bench.py pins its headline this way).
        # MDGAT_ARITHMETIC in the environment replaces the default for modules whose config does not carry the key.
        # 'f64_layers' (optional): how many leading layers run in fp64 (None: through the last layer with a k; 0: encoders only).
        self.arithmetic = str(self.config.get('arithmetic') or os.environ.get('MDGAT_ARITHMETIC') or 'auto')
        if self.arithmetic not in ('auto', 'fp32', 'fp64'):
            raise ValueError(f"arithmetic={self.arithmetic!r}: expected 'auto', 'fp32' or 'fp64'")
        f64_layers = self.config.get('f64_layers')
        self.f64_layers = None if f64_layers is None or int(f64_layers) < 0 else int(f64_layers)
        # 'sinkhorn_arithmetic' (optional; MDGAT_SINKHORN_ARITHMETIC in the environment): the exact mode's TAIL - every layer, final_proj,
        # the score matrix and the optimal transport in fp64, every arg-max of the extraction decided on the fp64 Z (csrc/sinkhorn_f64.hip).
        #   'auto' (default): on for frames of at most 2175 keypoints (beyond 575 the Sinkhorn streams its couplings from memory, one launch
        #   per iteration), else the fp32-class tail behind the last dynamic layer;
        #   'fp64': required (larger frames are refused);  'fp32': the fp32-class tail (Z good to 7e-6: inside the bar of 1e-4, but an
        #   arg-max whose two candidates lie closer than that may fall the other way - one in 40 960 on a reference-held batch).
        self.sinkhorn_arithmetic = str(self.config.get('sinkhorn_arithmetic') or os.environ.get('MDGAT_SINKHORN_ARITHMETIC') or 'auto')
        if self.sinkhorn_arithmetic not in ('auto', 'fp32', 'fp64'):
            raise ValueError(f"sinkhorn_arithmetic={self.sinkhorn_arithmetic!r}: expected 'auto', 'fp32' or 'fp64'")
        if self.arithmetic == 'fp64' and self.attention_dtype != 'fp32':
            raise ValueError("arithmetic='fp64' and attention_dtype='f16' exclude each other")
        if self.descriptor != 'FPFH':
            raise NotImplementedError(
                f"descriptor={self.descriptor!r}: only the 'FPFH' hot path is implemented on MI355X "
                "(pointnet / FPFH_gloabal / FPFH_only variants are out of scope, see DESIGN.md)")
        d = self.config['descriptor_dim']
        if d != _D or list(self.config['keypoint_encoder']) != [32, 64, 128] or \
                list(self.config['descritor_encoder']) != [64, 128]:
            raise NotImplementedError('the HIP kernels implement the default widths: descriptor_dim=128, '
                                      'keypoint_encoder=[32,64,128], descritor_encoder=[64,128]')
        self.kenc = _Encoder(4, self.config['keypoint_encoder'], d)
        self.denc = _Encoder(33, self.config['descritor_encoder'], d)
        self.gnn = _GNN(d, 2 * L)
        self.final_proj = nn.Conv1d(d, d, kernel_size=1, bias=True)
        self.register_parameter('bin_score', nn.Parameter(torch.tensor(1.)))
        # shared (by reference) with DataParallel replicas: device index -> _DeviceState
        self._states: Dict[int, _DeviceState] = {}
        self._states_lock = threading.RLock()
        # Also shared with the replicas (lists, so that a replica's shallow __dict__ copy sees later updates):
        # [0] the packed fp32 host blob of the owner's parameters.  torch.nn.parallel.replicate() strips the parameters
        #     off the replicas (they become plain attributes; state_dict() of a replica holds only buffers), so a
        #     replica cannot pack - the owner packs in _replicate_for_data_parallel(), before the replicas run;
        # [1] True while a blob installed by load_packed() (e.g. received by an RCCL broadcast) stands in for this
        #     module's own parameters: casts / moves of the module must not throw it away.
        self._blob_holder = [None, False]
        self._blob64_holder = [None]        # exact mode: the same blob before its rounding to fp32 (shared like [0] above)
        self._sig_holder = [self._signature()]
        # replicas never run __init__, so only the original module owns (and finally frees) the handles
        weakref.finalize(self, _close_states, self._states)

    # ------------------------------------------------------------------ copy / pickle
    _RUNTIME_ATTRS = ('_states', '_states_lock', '_blob_holder', '_blob64_holder', '_sig_holder')

    def __getstate__(self):
        """copy.deepcopy(net) / torch.save(net) (the reference's nn.Module supports both): library handles, locks and packed
        blobs are runtime state of THIS object and are rebuilt on first use of the copy."""
        d = self.__dict__.copy()
        for k in self._RUNTIME_ATTRS:
            d.pop(k, None)
        return d

    def __setstate__(self, d):
        super().__setstate__(d)
        self._states = {}
        self._states_lock = threading.RLock()
        self._blob_holder = [None, False]
        self._blob64_holder = [None]
        self._sig_holder = [self._signature()]
        weakref.finalize(self, _close_states, self._states)

    # ------------------------------------------------------------------ cache invalidation
    def _invalidate(self):
        with self._states_lock:
            for st in self._states.values():
                st.close()
            self._states.clear()
            self._blob_holder[0] = None
            self._blob_holder[1] = False
            self._blob64_holder[0] = None

    def exact(self) -> bool:
        """Does a forward of this module, as it stands, run the reference-exact (fp64) mode?  'auto' follows the module's
        dtype (net.double() -> True) unless attention_dtype='f16' was asked for, which is a throughput mode by definition."""
        arith = getattr(self, 'arithmetic', 'auto')
        if arith == 'auto':
            return self.bin_score.dtype == torch.float64 and getattr(self, 'attention_dtype', 'fp32') == 'fp32'
        return arith == 'fp64'

    def _signature(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version, t.dtype, t.device) for t in ts)

    def _invalidate_if_changed(self):
        # test.py:193 calls net.double().eval() before EVERY forward: a cast that changes nothing must not
        # throw the packed weights away
        sig = self._signature()
        if sig != self._sig_holder[0]:
            self._sig_holder[0] = sig
            if self._blob_holder[1]:
                # the weights in use were installed by load_packed(): this module's own parameters (random init on
                # every rank but the broadcasting one) are not what runs, so moving / casting them changes nothing
                return
            self._invalidate()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if hasattr(self, '_sig_holder'):
            self._invalidate_if_changed()
        return out

    def load_state_dict(self, state_dict, *a, **k):
        out = super().load_state_dict(state_dict, *a, **k)
        self._sig_holder[0] = self._signature()
        self._invalidate()          # new parameters: they replace whatever ran before, a load_packed() blob included
        return out

    def repack(self):
        """Call after modifying parameters in place (nothing else tracks in-place edits); also ends the reign of a
        blob installed by load_packed()."""
        self._invalidate()

    def _replicate_for_data_parallel(self):
        # torch.nn.DataParallel (test.py:158) calls this on the owner, in the caller's thread, before every forward
        # with more than one device: pack here, where the parameters still are parameters
        if not self._blob_holder[1]:
            self._host_blob()
        return super()._replicate_for_data_parallel()

    # ------------------------------------------------------------------ library state
    def _extract_mode(self):
        if self.loss_method == 'superglue':
            return _lib.EXTRACT_THRESHOLD_MUTUAL if self.mutual_check else _lib.EXTRACT_THRESHOLD
        return _lib.EXTRACT_DUSTBIN_MUTUAL if self.mutual_check else _lib.EXTRACT_DUSTBIN

    def _topk_schedule(self):
        return pack.resolve_topk_schedule(self.config['L'], list(self.k))

    def packed_weights(self, dtype=None):
        """fp32 blob (numpy) of the current parameters in the library's layout (``dtype=numpy.float64``: before the
        rounding to fp32 - what arithmetic='fp64' loads in addition)."""
        import numpy as np
        return pack.pack_state_dict(self.state_dict(), self.config['L'], dtype=dtype or np.float32)

    def _host_blob(self):
        """The packed blob, made once per set of parameters and shared with DataParallel replicas."""
        with self._states_lock:
            if self._blob_holder[0] is None:
                if 'bin_score' not in self._parameters:
                    raise RuntimeError('this MDGAT is a DataParallel replica without packed weights: the owner module '
                                       'packs them in _replicate_for_data_parallel() - was replicate() bypassed?')
                self._blob_holder[0] = self.packed_weights()
            if self._blob64_holder[0] is None and self.exact() and 'bin_score' in self._parameters and not self._blob_holder[1]:
                import numpy as np
                self._blob64_holder[0] = self.packed_weights(np.float64)
            return self._blob_holder[0]

    def _state_for(self, device: torch.device, blob_device_tensor: Optional[torch.Tensor] = None) -> _DeviceState:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self._states_lock:
            st = self._states.get(idx)
            if st is not None and (st.f64 == self.exact() or blob_device_tensor is not None):
                return st
            if st is not None:
                # the module's dtype changed under a blob installed by load_packed() (casts keep such a blob): the handle's
                # arithmetic is fixed at creation, so it is rebuilt from the host copies of the blob(s)
                self._states.pop(idx).close()
            lib = _lib.load()
            L = self.config['L']
            cfg = _lib.MdgatConfig()
            cfg.L = L
            cfg.sinkhorn_iters = int(self.config['sinkhorn_iterations'])
            sched = self._topk_schedule()
            for i, kk in enumerate(sched):
                cfg.topk[i] = kk
            cfg.extract_mode = self._extract_mode()
            cfg.match_threshold = float(self.config['match_threshold'])
            cfg.attention_mode = 0 if self.attention_dtype == 'fp32' else 1
            f64 = self.exact()
            cfg.arithmetic = _lib.ARITH_FP64 if f64 else _lib.ARITH_FP32
            fl = getattr(self, 'f64_layers', None)
            cfg.f64_layers = 0 if fl is None else (_lib.F64_ENCODERS_ONLY if fl == 0 else int(fl))     # (C ABI: 0 = automatic)
            cfg.f64_sinkhorn = {'auto': 0, 'fp64': 1, 'fp32': -1}[getattr(self, 'sinkhorn_arithmetic', 'auto')]
            handle = C.c_void_p()
            _lib.check(lib.mdgat_create(C.byref(cfg), idx, C.byref(handle)), 'mdgat_create')
            st = _DeviceState(handle, idx, f64)
            try:
                if self.lanes:
                    _lib.check(lib.mdgat_set_lanes(handle, self.lanes), 'mdgat_set_lanes')
                if blob_device_tensor is not None:
                    n = blob_device_tensor.numel()
                    _lib.check(lib.mdgat_load_weights(handle, C.c_void_p(blob_device_tensor.data_ptr()), n, 1),
                               'mdgat_load_weights')
                else:
                    blob = self._host_blob()
                    assert blob.size == lib.mdgat_blob_floats(L), (blob.size, lib.mdgat_blob_floats(L))
                    _lib.check(lib.mdgat_load_weights(handle, blob.ctypes.data_as(C.c_void_p), blob.size, 0),
                               'mdgat_load_weights')
                if f64:
                    blob64 = self._blob64_holder[0]
                    if blob64 is None:
                        raise RuntimeError('the exact mode (a float64 module, or arithmetic=\'fp64\') needs the fp64 blob: '
                                           'load_packed(blob, blob64) on ranks that received their weights by broadcast')
                    _lib.check(lib.mdgat_load_weights_f64(handle, blob64.ctypes.data_as(C.c_void_p), blob64.size, 0),
                               'mdgat_load_weights_f64')
            except Exception:
                st.close()
                raise
            self._states[idx] = st
            return st

    def set_lanes(self, lanes: int):
        """1: every kernel of a batch on the caller's stream; 2 (library default): the halves of a batch in flight on two
        streams (csrc/api.hip: forward_batched).  Applies to this module's handles on every device, now and later."""
        if lanes not in (1, 2):
            raise ValueError(f'lanes={lanes}: expected 1 or 2')
        with self._states_lock:
            self.lanes = lanes
            for st in self._states.values():
                with st.lock:
                    _lib.check(_lib.load().mdgat_set_lanes(st.handle, lanes), 'mdgat_set_lanes')

    def load_packed(self, blob: torch.Tensor, blob64: Optional[torch.Tensor] = None):
        """Install an already packed fp32 blob that lives on a GPU (e.g. received by an RCCL broadcast,
        see shard.broadcast_weights) instead of packing this module's own parameters.  arithmetic='fp64' needs
        ``blob64`` as well: the same blob in float64 (``packed_weights(numpy.float64)``)."""
        assert blob.is_cuda and blob.dtype == torch.float32 and blob.is_contiguous()
        if self.exact():
            if blob64 is None or blob64.dtype != torch.float64 or blob64.numel() != blob.numel():
                raise ValueError("the exact mode (a float64 module, or arithmetic='fp64'): load_packed needs blob64, the float64 "
                                 'blob of the same layout')
        idx = blob.device.index
        with self._states_lock:
            old = self._states.pop(idx, None)
            if old is not None:
                old.close()
            # a host copy as well: any other device of this process (DataParallel replicas, a later .to()) loads the SAME
            # weights from it - never this module's own parameters, which are random init on a rank that received a blob
            self._blob_holder[0] = blob.detach().cpu().numpy().copy()
            self._blob64_holder[0] = blob64.detach().cpu().numpy().copy() if blob64 is not None else None
            self._blob_holder[1] = True     # stands until load_state_dict() / repack(): see _invalidate_if_changed
            self._sig_holder[0] = self._signature()
            for other in [i for i in self._states if i != idx]:
                self._states.pop(other).close()
            return self._state_for(blob.device, blob)

    # ------------------------------------------------------------------ forward
    def forward(self, data):
        kpts0, kpts1 = data['keypoints0'], data['keypoints1']
        out_dtype = self.bin_score.dtype
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:      # mdgat.py:374-382
            shape0, shape1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return {
                'matches0': kpts0.new_full(shape0, -1, dtype=torch.int)[0],
                'matches1': kpts1.new_full(shape1, -1, dtype=torch.int)[0],
                'matching_scores0': kpts0.new_zeros(shape0, dtype=torch.float64)[0],
                'matching_scores1': kpts1.new_zeros(shape1, dtype=torch.float64)[0],
                'skip_train': True,
            }
        if self.training:
            raise NotImplementedError('mdgat_matcher_amd implements inference only: call .eval() (training, the '
                                      'losses of mdgat.py:486-594 and backward are out of scope)')
        token = [0]
        res = self._run(kpts0, data['scores0'], data['descriptors0'], kpts1, data['scores1'], data['descriptors1'], token_out=token)
        m0, m1, s0, s1 = res[:4]
        s0, s1 = s0.to(out_dtype), s1.to(out_dtype)
        if self.loss_method != 'superglue':
            # mdgat.py:464-467: `if valid0.sum() == 0` - a host-side test in the reference too (it synchronises) - returns
            # INTEGER zero scores (torch.zeros_like(indices)) when no frame-0 keypoint of the whole batch is matched.  The
            # kernels have already zeroed the scores; the dtype follows here.
            # (the library answers from a host-mapped word its extraction kernels write - mdgat_matched_any - so the test costs a
            # stream synchronisation, as in the reference, but no reduction kernel and no copy)
            torch.cuda.current_stream(m0.device).synchronize()
            nothing_matched = not self._matched_any(m0.device, token[0])
            self.check(m0.device, synchronize=False)        # (synchronised above: report this call's status now)
            if nothing_matched:
                s0, s1 = torch.zeros_like(m0), torch.zeros_like(m1)
        else:
            self.check(m0.device)                           # the dict API reports on the failing call in every branch
        return {
            'matches0': m0,
            'matches1': m1,
            'matching_scores0': s0,
            'matching_scores1': s1,
            'loss': m0.new_zeros((), dtype=out_dtype),     # losses are training-only: not computed
        }

    def _matched_any(self, device, token=0) -> bool:
        """Did the forward that carried ``token`` on ``device`` (already synchronised by the caller) match any frame-0 keypoint?
        (mdgat.py:465.)  The token is per call - read under the handle's lock right after the enqueue (``_run``) - so forwards
        other threads or streams put on the same handle in the meantime do not change the answer; 0 = the handle's last call."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self._states_lock:
            st = self._states.get(idx)
        flag = C.c_uint(0)
        _lib.check(_lib.load().mdgat_matched_any(st.handle, int(token), C.byref(flag)), 'mdgat_matched_any')
        return bool(flag.value)

    def check(self, device=None, synchronize=True):
        """Status of the asynchronous forwards on ``device`` since the last check.  Raises ``RuntimeError`` if an
        activation left the f16 operand range (|v| >= 6e4) or a non-finite value reached a kernel - the outputs of those
        calls are invalid (``mdgat_async_status``); returns ``{'sinkhorn_fallback': bool}`` otherwise (informational: a
        Sinkhorn launch that lost a partner workgroup was redone by the streaming kernel, results valid).  ``forward``
        (the dict API of the reference) calls this itself; after ``match()`` / ``match_frames()`` - which never
        synchronise - call it once the results are needed."""
        dev = torch.device(device) if device is not None else self.bin_score.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with self._states_lock:
            st = self._states.get(idx)
        if st is None:
            return {'sinkhorn_fallback': False}
        if synchronize:
            # Every stream this module's forwards were enqueued on (the per-stream workspaces remember them) - not just the
            # current one: the status words are per handle, and a caller that ran _run / match_frames on stream A and asks from
            # stream B would otherwise read (and clear) them before A's kernels have written.  Other streams of the process
            # are left alone (no device-wide synchronisation).
            torch.cuda.current_stream(dev).synchronize()
            with st.lock:
                handles = list(st.workspaces)
            cur = torch.cuda.current_stream(dev).cuda_stream
            for hnd in handles:
                if hnd == cur:
                    continue
                if hnd == 0:        # the legacy default stream: pool streams are non-blocking and do not wait for it
                    torch.cuda.default_stream(dev).synchronize()
                else:
                    _sync_raw_stream(hnd, dev)
        fb, rg = C.c_uint(0), C.c_uint(0)
        _lib.check(_lib.load().mdgat_async_status(st.handle, 1, C.byref(fb), C.byref(rg)), 'mdgat_matcher_amd')
        return {'sinkhorn_fallback': bool(fb.value)}

    @staticmethod
    def _f32(t, device):
        return t.to(device=device, dtype=torch.float32).contiguous()

    def _run(self, kpts0, sigma0, fpfh0, kpts1, sigma1, fpfh1, want_Z=False, taps=None, frames=None, normalize=True, token_out=None):
        """One forward through the library on the current stream.  Either six arrays (keypoints / saliency / FPFH per
        frame) or ``frames=(records0, records1)`` raw [B, N, 37] loader records.  Asynchronous; returns device tensors
        ``(matches0, matches1, mscores0, mscores1, Z or None)``."""
        probe = frames[0] if frames is not None else kpts0
        if not probe.is_cuda:
            raise RuntimeError('mdgat_matcher_amd runs on MI355X (gfx950) only: inputs must be on a CUDA/HIP '
                               'device; there is no CPU fallback')
        dev = probe.device
        st = self._state_for(dev)
        f64 = st.f64
        if frames is not None:
            if frames[0].shape[-1] != 37 or frames[1].shape[-1] != 37 or frames[0].dim() != 3:
                raise ValueError('expected frame records [B, N, 37] = xyz | saliency | 33-D FPFH (load_data.py:152-165)')
            ins = [self._f32(frames[0], dev), self._f32(frames[1], dev)]
            B, N, M = ins[0].shape[0], ins[0].shape[1], ins[1].shape[1]
        else:
            if fpfh0.shape[-1] != 33 or fpfh1.shape[-1] != 33 or kpts0.shape[-1] != 3 or kpts1.shape[-1] != 3:
                raise ValueError('expected keypoints [B, N, 3] and 33-D FPFH descriptors [B, N, 33]')
            in_dtype = torch.float64 if f64 else torch.float32
            ins = [t.to(device=dev, dtype=in_dtype).contiguous() for t in (kpts0, sigma0, fpfh0, kpts1, sigma1, fpfh1)]
            B, N, M = kpts0.shape[0], kpts0.shape[1], kpts1.shape[1]
        lib = _lib.load()
        with torch.cuda.device(dev), st.lock:
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = st.workspace_for(stream, lib.mdgat_workspace_bytes(st.handle, B, N, M), dev)
            m0 = torch.empty((B, N), dtype=torch.int64, device=dev)
            m1 = torch.empty((B, M), dtype=torch.int64, device=dev)
            s0 = torch.empty((B, N), dtype=torch.float32, device=dev)
            s1 = torch.empty((B, M), dtype=torch.float32, device=dev)
            Z = torch.empty((B, N + 1, M + 1), dtype=torch.float32, device=dev) if want_Z else None
            tap_struct = None
            if taps is not None:
                tap_struct = _lib.MdgatTaps()
                for name in _lib.TAP_NAMES:
                    t = taps.get(name)
                    setattr(tap_struct, name, t.data_ptr() if t is not None else None)
            outs = (m0.data_ptr(), m1.data_ptr(), s0.data_ptr(), s1.data_ptr(), Z.data_ptr() if Z is not None else None,
                    C.byref(tap_struct) if tap_struct is not None else None, ws.data_ptr(), ws.numel(), stream)
            if frames is not None:
                rc = lib.mdgat_forward_frames(st.handle, B, N, M, ins[0].data_ptr(), ins[1].data_ptr(), int(bool(normalize)), *outs)
            elif f64:
                rc = lib.mdgat_forward_f64(st.handle, B, N, M, *[t.data_ptr() for t in ins], *outs)
            else:
                rc = lib.mdgat_forward(st.handle, B, N, M, *[t.data_ptr() for t in ins], *outs)
            if token_out is not None:
                token_out[0] = int(lib.mdgat_last_token(st.handle))     # (still under st.lock: this call's token)
            _lib.check(rc, 'mdgat_forward_frames' if frames is not None else 'mdgat_forward_f64' if f64 else 'mdgat_forward')
        return m0, m1, s0, s1, Z

    def profile(self, device, enable: bool):
        """Switch the library's per-kernel-class HIP-event timing of the forward on/off for ``device`` and
        return what was accumulated since the previous call: ``{class: (total_ms, launches)}``."""
        st = self._state_for(torch.device(device))
        ms = (C.c_double * len(_lib.PROF_CLASSES))()
        n = (C.c_longlong * len(_lib.PROF_CLASSES))()
        with st.lock:
            _lib.check(_lib.load().mdgat_profile(st.handle, int(bool(enable)), ms, n), 'mdgat_profile')
        return {name: (ms[i], n[i]) for i, name in enumerate(_lib.PROF_CLASSES)}

    @torch.no_grad()
    def match_frames(self, frames0, frames1, normalize=True, return_scores=False):
        """Match straight from the loader's raw keypoint records (``load_data.py:146-165``): ``frames`` are
        ``[B, N, 37]`` (or ``[N, 37]``) float32 rows ``xyz | saliency | FPFH``, exactly the content of the KITTI
        keypoint ``.bin`` files.  The record split and the FPFH L2 normalisation (``load_data.py:290-292``) happen
        inside the encoder kernel.  Returns ``(matches0, matches1, mscores0, mscores1[, Z])``."""
        single = frames0.dim() == 2
        if single:
            frames0, frames1 = frames0[None], frames1[None]
        m0, m1, s0, s1, Z = self._run(None, None, None, None, None, None, want_Z=return_scores, frames=(frames0, frames1),
                                      normalize=normalize)
        outs = [m0, m1, s0, s1] + ([Z] if return_scores else [])
        if single:
            outs = [o[0] for o in outs]
        return tuple(outs)

    # ------------------------------------------------------------------ match() API
    @torch.no_grad()
    def match(self, kpts0, desc0, kpts1, desc1, scores0=None, scores1=None, return_scores=False):
        """``match(kpts0, desc0, kpts1, desc1)`` convenience API named by the north star.

        kpts [B, N, 3] (or [N, 3]), desc = 33-D FPFH rows (L2-normalised as load_data.py:290-292 does),
        scores = per-keypoint saliency, which the keypoint encoder consumes (mdgat.py:184-188) and is
        therefore required.  Returns ``(matches0, matches1, mscores0, mscores1[, Z])``; ``Z`` is the
        (N+1) x (M+1) log assignment matrix of log_optimal_transport."""
        if scores0 is None or scores1 is None:
            raise ValueError('match() needs the keypoint saliency scores0/scores1 (KeypointEncoder input)')
        single = kpts0.dim() == 2
        if single:
            kpts0, desc0, kpts1, desc1 = kpts0[None], desc0[None], kpts1[None], desc1[None]
            scores0, scores1 = scores0[None], scores1[None]
        m0, m1, s0, s1, Z = self._run(kpts0, scores0, desc0, kpts1, scores1, desc1, want_Z=return_scores)
        outs = [m0, m1, s0, s1] + ([Z] if return_scores else [])
        if single:
            outs = [o[0] for o in outs]
        return tuple(outs)


def match(model: MDGAT, kpts0, desc0, kpts1, desc1, scores0=None, scores1=None, return_scores=False):
    """Functional form of :meth:`MDGAT.match`."""
    return model.match(kpts0, desc0, kpts1, desc1, scores0, scores1, return_scores=return_scores)
