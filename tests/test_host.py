"""CPU-side tests: host logic, weight packing, and that the C-ABI library loads and exports every
symbol include/mdgat_hip.h declares (no GPU compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from mdgat_matcher_amd import MDGAT, _lib, pack, synth
from oracle import mdgat_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'mdgat_hip.h')).read()
    declared = set(re.findall(r'\b(mdgat_[a-z_0-9]+)\s*\(', hdr))
    declared -= {'mdgat_status', 'mdgat_extract_mode'}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert isinstance(_lib.last_error(), str)


def test_blob_layout_matches_library():
    lib = _lib.load()
    for L in (0, 1, 4, 9, 32):
        assert lib.mdgat_blob_floats(L) == pack.blob_layout(L)['total']
    assert C.sizeof(_lib.MdgatConfig) == 4 * (2 + 64 + 6)     # L, iters, topk[64], extract_mode, threshold, attention_mode, arithmetic, f64_layers, f64_sinkhorn


def test_state_dict_names_match_reference_fixture():
    net = MDGAT(synth.default_config(L=9))
    sd = synth.make_state_dict(L=9)
    assert set(net.state_dict()) == set(sd) and len(sd) == 348          # SURVEY.md section 5 [probe]
    assert all(tuple(net.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    assert sum(p.numel() for p in net.parameters()) == 3045921          # SURVEY.md section 2.1 [probe]
    net.load_state_dict(sd, strict=True)
    dp = torch.nn.DataParallel(net)
    dp.load_state_dict({'module.' + k: v for k, v in sd.items()}, strict=True)
    assert float(MDGAT(synth.default_config(L=1)).bin_score) == 1.0     # mdgat.py:359
    assert float(MDGAT(synth.default_config(L=1)).kenc.encoder[-1].bias.abs().sum()) == 0.0   # mdgat.py:182


def test_config_contract():
    cfg = synth.default_config(L=2)
    for key in ('descriptor', 'lr', 'loss_method', 'k', 'mutual_check', 'triplet_loss_gamma', 'train_step', 'L'):
        bad = dict(cfg)
        del bad[key]
        with pytest.raises(KeyError):
            MDGAT(bad)
    with pytest.raises(NotImplementedError):
        MDGAT(synth.default_config(L=2, descriptor='pointnet'))
    net = MDGAT(cfg)
    assert net.config['sinkhorn_iterations'] == 100 and net.config['match_threshold'] == 0.2
    # the one key the reference does not have
    assert net.attention_dtype == 'fp32' and MDGAT(dict(cfg, attention_dtype='f16')).attention_dtype == 'f16'
    with pytest.raises(ValueError):
        MDGAT(dict(cfg, attention_dtype='fp8'))


def test_create_rejects_bad_config_before_touching_a_device():
    lib = _lib.load()
    for field, value in (('L', 33), ('L', -1), ('attention_mode', 7), ('extract_mode', 4)):
        c = _lib.MdgatConfig()
        c.L = 1
        setattr(c, field, value)
        h = C.c_void_p()
        assert lib.mdgat_create(C.byref(c), 0, C.byref(h)) == _lib.ERR_BAD_ARG, field
        assert not h.value and lib.mdgat_last_error()


def test_no_cpu_fallback():
    net = MDGAT(synth.default_config(L=1, k=[])).eval()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(synth.make_batch(1, 8, 8))
    net.train()
    with pytest.raises(NotImplementedError):
        net(synth.make_batch(1, 8, 8))


def test_empty_keypoints_early_out_on_cpu(golden_dir):
    g = np.load(os.path.join(golden_dir, 'edge_cases.npz'))
    net = MDGAT(synth.default_config(L=1, k=[])).double().eval()
    data = synth.make_batch(1, 8, 8)
    data['keypoints0'] = data['keypoints0'][:, :0]
    out = net(data)
    assert out['skip_train'] is True
    np.testing.assert_array_equal(out['matches0'].numpy(), g['empty_matches0'])
    np.testing.assert_array_equal(out['matches1'].numpy(), g['empty_matches1'])
    np.testing.assert_array_equal(out['matching_scores1'].numpy(), g['empty_mscores1'])
    assert out['matches1'].dtype == torch.int32 and out['matching_scores1'].dtype == torch.float64


def test_topk_schedule():
    assert pack.resolve_topk_schedule(9, synth.DEFAULT_K) == [0] * 10 + [128, 0, 128, 0, 64, 0, 64, 0]
    for L, k in ((4, synth.DEFAULT_K), (2, []), (3, [5, None]), (1, [7, 8, 9, 10])):
        ref = [0 if x is None else x for x in O.layer_topk_schedule(L, k)]
        assert pack.resolve_topk_schedule(L, k) == ref


def test_packed_weights_reproduce_the_oracle_layer_math():
    """BN folding, head de-interleave and the merge->mlp.0 fold, checked in fp64 numpy against the
    oracle on one attentional-propagation layer and the encoders."""
    L = 2
    sd = synth.make_state_dict(L=L, seed=11)
    lay = pack.blob_layout(L)
    sd64 = {k: v for k, v in sd.items()}
    # pack in fp64 (bypass the final fp32 cast) by re-running the packer's math
    blob = pack.pack_state_dict(sd64, L).astype(np.float64)
    rs = np.random.RandomState(0)
    B, n = 1, 24
    data = synth.make_batch(B, n, n)
    cap = {}
    O.mdgat_forward(sd, synth.default_config(L=L, k=[]), data, cap)

    def W(name, shape, base=0):
        off = base + lay[name]
        return blob[off:off + int(np.prod(shape))].reshape(shape)
    # encoders
    k = data['keypoints0'][0].numpy(); s = data['scores0'][0].numpy(); f = data['descriptors0'][0].numpy()
    hk0 = np.maximum(np.concatenate([k, s[:, None]], 1) @ W('kenc0_w', (32, 4)).T + W('kenc0_b', (32,)), 0)
    hd0 = np.maximum(f @ W('denc0_w', (64, 33)).T + W('denc0_b', (64,)), 0)
    hk1 = np.maximum(hk0 @ W('kenc1_w', (64, 32)).T + W('kenc1_b', (64,)), 0)
    hk2 = np.maximum(hk1 @ W('kenc2_w', (128, 64)).T + W('kenc2_b', (128,)), 0)
    hd1 = np.maximum(hd0 @ W('denc1_w', (128, 64)).T + W('denc1_b', (128,)), 0)
    x0 = np.concatenate([hd1, hk2], 1) @ W('encl_w', (128, 256)).T + W('encl_b', (128,))
    assert np.abs(x0 - cap['enc0'][0].numpy().T).max() < 2e-5     # blob is fp32-rounded
    # layer 0 (self): attention in head-major layout with folded merge
    base = lay['layer0']
    x = cap['enc0'][0].numpy().T                                   # [n, 128]
    qkv = x @ W('qkv_w', (384, 128), base).T + W('qkv_b', (384,), base)
    q, kk, v = (qkv[:, i * 128:(i + 1) * 128].reshape(n, 4, 32) for i in range(3))
    logits = np.einsum('nhd,mhd->hnm', q, kk) / np.sqrt(32)
    p = np.exp(logits - logits.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    msg = np.einsum('hnm,mhd->nhd', p, v).reshape(n, 128)
    hid = np.maximum(np.concatenate([x, msg], 1) @ W('mlp1_w', (256, 256), base).T + W('mlp1_b', (256,), base), 0)
    xn = x + hid @ W('mlp2_w', (128, 256), base).T + W('mlp2_b', (128,), base)
    assert np.abs(xn - cap['layer0_desc0'][0].numpy().T).max() < 5e-5
    assert abs(blob[lay['bin_score']] - 1.0) < 1e-7


def test_pack_rejects_unsupported():
    sd = synth.make_state_dict(L=2)
    with pytest.raises(ValueError):
        pack.pack_state_dict(sd, 3)
    bad = dict(sd)
    bad['kenc.encoder.0.weight'] = torch.zeros(16, 4, 1)
    with pytest.raises(ValueError):
        pack.pack_state_dict(bad, 2)


def test_double_eval_does_not_repack():
    net = MDGAT(synth.default_config(L=1, k=[]))
    net.load_state_dict(synth.make_state_dict(L=1))
    net.double().eval()
    sig = net._sig_holder[0]
    sentinel = object()
    net._states[99] = type('S', (), {'close': lambda self: None})()
    net.double().eval()                   # test.py:193 does this before every forward
    assert net._sig_holder[0] == sig and 99 in net._states
    net.float()
    assert 99 not in net._states          # a real cast drops the packed weights


def test_oracle_postproc_restatements():
    """The oracle's restatements of utils_test.solve_icp and the loader's ground-truth matcher on a known case:
    exact correspondences under a known rigid motion."""
    from oracle import mdgat_oracle as O
    rs = np.random.RandomState(3)
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    t = np.array([1.0, -2.0, 0.5])
    P = rs.standard_normal((40, 3)) * 10
    Q = P @ R.T + t
    T = O.solve_icp(P, Q)
    assert np.abs(T[:3, :3] - R).max() < 1e-12 and np.abs(T[:3, 3] - t).max() < 1e-11
    Tp, n, inl, ratio, te, re = O.pose_from_matches(Q, P, np.arange(40), T_gt=T)
    assert n == 40 and inl == 40 and te < 1e-10 and (re < 1e-6 or np.isnan(re))
    # ground-truth matcher: frame 1 = permuted frame 0 -> the permutation comes back, mutual or not
    perm = rs.permutation(40)
    for mutual in (False, True):
        m0, m1, rep = O.gt_matches(P, P[perm], threshold=0.5, mutual=mutual)
        assert rep == 40 and (P[perm][m0] == P).all() and (m1 == perm).all()


def _emulate_replicate(net):
    """What torch.nn.parallel.replicate() does to a module tree, minus the device broadcast (which needs GPUs): every
    module is shallow-copied by _replicate_for_data_parallel(), the copies are re-wired, and parameters become plain
    tensor attributes of the copies (replicate.py: 'parameters in replicas are no longer leaves')."""
    mods = list(net.modules())
    idx = {m: i for i, m in enumerate(mods)}
    copies = [m._replicate_for_data_parallel() for m in mods]
    for m, c in zip(mods, copies):
        for key, child in m._modules.items():
            c._modules[key] = None if child is None else copies[idx[child]]
        for key, param in m._parameters.items():
            if param is not None:
                setattr(c, key, param.detach().clone())
    return copies[0]


def test_dataparallel_replica_uses_the_owners_packed_weights():
    """DataParallel with more than one device replicates the module; replicas have no parameters
    (state_dict() holds buffers only), so a replica must not pack - the owner packs when it is replicated and the
    replicas share that blob."""
    L = 2
    net = MDGAT(synth.default_config(L=L, k=[]))
    net.load_state_dict(synth.make_state_dict(L=L, seed=3))
    net = net.double().eval()
    ref = net.packed_weights()
    rep = _emulate_replicate(net)
    assert 'bin_score' not in rep._parameters and len(rep.state_dict()) < len(net.state_dict())
    with pytest.raises(KeyError):
        rep.packed_weights()                       # what the round-1 code did on the replica
    np.testing.assert_array_equal(rep._host_blob(), ref)
    assert rep._states is net._states and rep._blob_holder is net._blob_holder
    # new parameters on the owner invalidate the shared blob; the next replication packs the new ones
    net.load_state_dict(synth.make_state_dict(L=L, seed=4))
    assert net._blob_holder[0] is None
    rep2 = _emulate_replicate(net)
    np.testing.assert_array_equal(rep2._host_blob(), net.packed_weights())
    assert not np.array_equal(rep2._host_blob(), ref)
    # a replica made behind the owner's back says what is wrong instead of a KeyError from the packer
    net._invalidate()
    orphan = MDGAT._replicate_for_data_parallel.__wrapped__(net) if hasattr(MDGAT._replicate_for_data_parallel, '__wrapped__') \
        else torch.nn.Module._replicate_for_data_parallel(net)
    orphan._parameters = {}
    with pytest.raises(RuntimeError, match='replica'):
        orphan._host_blob()


def test_load_packed_state_survives_casts():
    """a device state installed by load_packed() (weights received by the RCCL broadcast) must
    survive net.double() / .to() / .float() - test.py:193 calls net.double().eval() before every forward - and end
    with load_state_dict() or repack()."""
    class _State:
        closed = False
        def close(self):
            self.closed = True
    L = 1
    net = MDGAT(synth.default_config(L=L, k=[]))
    st = _State()
    net._states[0] = st
    net._blob_holder[1] = True                    # what load_packed() records
    net.double()
    net.float()
    net.double().eval()
    assert net._states.get(0) is st and not st.closed
    net.load_state_dict(synth.make_state_dict(L=L, seed=1))
    assert st.closed and not net._states and net._blob_holder[1] is False
    # without load_packed a real change of the parameters does invalidate, a no-op cast does not
    st2 = _State()
    net._states[0] = st2
    net.double()
    net.double().eval()
    assert net._states.get(0) is st2 and not st2.closed
    net.float()
    assert st2.closed and not net._states


# ----------------------------------------------------------------------------------------------------------------
# round 3: host logic around the library, with the library replaced by a recorder at the _lib boundary
class _FakeLib:
    """Records what reaches the C ABI; `status` = (sinkhorn_fallback, range_violation) the next mdgat_async_status reports."""

    def __init__(self):
        self.calls, self.loaded, self.status, self.matched = [], [], (0, 0), 0
        self.token, self.matched_by_token, self.loaded64 = 0, {}, []

    def mdgat_create(self, cfg, idx, handle):
        self.calls.append(('create', idx))
        handle._obj.value = 4000 + idx
        return 0

    def mdgat_blob_floats(self, L):
        return pack.blob_layout(L)['total']

    def mdgat_load_weights(self, handle, ptr, n, on_device):
        addr = ptr.value if hasattr(ptr, 'value') else C.cast(ptr, C.c_void_p).value
        self.loaded.append(np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_float)), shape=(n,)).copy())
        self.calls.append(('load', int(on_device)))
        return 0

    def mdgat_workspace_bytes(self, handle, B, N, M):
        return 256 * B

    def mdgat_load_weights_f64(self, handle, ptr, n, on_device):
        addr = ptr.value if hasattr(ptr, 'value') else C.cast(ptr, C.c_void_p).value
        self.loaded64.append(np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_double)), shape=(n,)).copy())
        self.calls.append(('load64', int(on_device)))
        return 0

    def mdgat_forward(self, *a):
        self.token += 1
        self.calls.append(('forward', a[1], a[-2], a[-1]))          # B, workspace bytes, stream
        return 0

    def mdgat_forward_f64(self, *a):
        self.token += 1
        self.calls.append(('forward_f64', a[1], a[-2], a[-1]))
        return 0

    def mdgat_last_token(self, handle):
        return self.token

    def mdgat_async_status(self, handle, clear, fb, rg):
        fb._obj.value, rg._obj.value = self.status
        if clear:
            self.status = (0, 0)
        return _lib.ERR_UNSUPPORTED if rg._obj.value else 0

    def mdgat_matched_any(self, handle, token, matched):
        self.calls.append(('matched_any', token))
        matched._obj.value = self.matched_by_token.get(token, self.matched)
        return 0

    def mdgat_destroy(self, handle):
        self.calls.append(('destroy',))

    def mdgat_last_error(self):
        return b'activations outside the f16 operand range'


def _stubbed(fake, stream_holder):
    """Context in which CPU tensors pose as device memory and the library is `fake`."""
    import contextlib
    from unittest import mock
    cm = contextlib.ExitStack()
    cm.enter_context(mock.patch.object(_lib, 'load', lambda: fake))
    cm.enter_context(mock.patch.object(torch.Tensor, 'is_cuda', new=property(lambda self: True)))
    cm.enter_context(mock.patch.object(torch.cuda, 'current_device', lambda: 0))
    cm.enter_context(mock.patch.object(torch.cuda, 'device', lambda d: contextlib.nullcontext()))
    cm.enter_context(mock.patch.object(torch.cuda, 'synchronize', lambda d=None: None))
    cm.enter_context(mock.patch.object(torch.cuda, 'current_stream', lambda d=None: mock.Mock(cuda_stream=stream_holder[0])))
    from mdgat_matcher_amd import mdgat as _m
    synced = stream_holder[1] if len(stream_holder) > 1 else []
    cm.enter_context(mock.patch.object(_m, '_sync_raw_stream', lambda h, d: synced.append(h)))
    return cm


def test_workspace_per_stream_and_status_check():
    """Forwards issued on different streams get different workspaces (they may overlap on the device); check() reports
    the asynchronous status: information about a Sinkhorn fallback, RuntimeError for the f16 range guard; forward()
    applies the integer-zero quirk of mdgat.py:465-467 and checks the status itself."""
    fake, stream = _FakeLib(), [11]
    net = MDGAT(synth.default_config(L=1, k=[])).eval()
    d = synth.make_batch(2, 8, 8)
    args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
    with _stubbed(fake, stream):
        net._run(*args)
        stream[0] = 22
        net._run(*args)
        stream[0] = 11
        net._run(*args)
        st = net._states[0]
        assert set(st.workspaces) == {11, 22} and st.workspaces[11].data_ptr() != st.workspaces[22].data_ptr()
        assert [c[3] for c in fake.calls if c[0] == 'forward'] == [11, 22, 11]
        stream.append([])                                                       # (records the raw streams check() synchronises)
        with _stubbed(fake, stream):
            assert net.check('cuda:0') == {'sinkhorn_fallback': False}
            assert stream[1] == [22]      # every stream the module's forwards ran on is synchronised, not only the current one (11)
        fake.status = (1, 0)
        assert net.check('cuda:0') == {'sinkhorn_fallback': True}
        fake.status = (0, 1)
        with pytest.raises(RuntimeError, match='f16 operand range'):
            net.check('cuda:0')
        assert net.check('cuda:0') == {'sinkhorn_fallback': False}          # reported once
        # forward(): the library says whether the call matched anything (mdgat_matched_any); "nothing matched" -> integer zeros
        fake.matched = 0
        out = net(d)
        assert out['matching_scores0'].dtype == torch.int64 and out['matching_scores1'].dtype == torch.int64      # mdgat.py:465-467
        assert not out['matching_scores0'].any() and not out['matching_scores1'].any()
        fake.matched = 1
        out = net(d)
        assert out['matching_scores0'].dtype == torch.float32 and out['matching_scores1'].dtype == torch.float32
        fake.status = (0, 1)
        with pytest.raises(RuntimeError, match='f16 operand range'):
            net(d)                                                          # the failing call raises, not the next one
        net._invalidate()


def test_load_packed_keeps_a_host_copy_for_other_devices():
    """a blob installed by load_packed() must be what EVERY device / replica of the process runs - never the
    module's own (random-init) parameters."""
    fake, stream = _FakeLib(), [0]
    net = MDGAT(synth.default_config(L=1, k=[])).eval()
    other = MDGAT(synth.default_config(L=1, k=[]))
    other.load_state_dict(synth.make_state_dict(L=1, seed=9, dtype=torch.float32))
    blob = torch.from_numpy(other.packed_weights())
    with _stubbed(fake, stream):
        from unittest import mock
        with mock.patch.object(torch.Tensor, 'device', new=property(lambda self: torch.device('cuda', 0))):
            net.load_packed(blob)
        assert fake.calls[-1] == ('load', 1) and net._blob_holder[1] is True
        np.testing.assert_array_equal(net._blob_holder[0], blob.numpy())
        st1 = net._state_for(torch.device('cuda', 1))                       # another device of the process
        assert fake.calls[-1] == ('load', 0) and st1.device == 1
        np.testing.assert_array_equal(fake.loaded[-1], blob.numpy())         # the installed blob, not net's own parameters
        net._invalidate()


def test_import_path_shim_resolves_models_mdgat(tmp_path):
    """`from models.superglue import SuperGlue; from models.mdgat import MDGAT` (test.py:11-12) with <repo>/integration ahead of
    a reference checkout on sys.path: those two resolve to this implementation, the rest of the reference's `models` package
    (models.pointnet...) to the checkout."""
    import importlib
    import sys
    shim = os.path.join(ROOT, 'integration')
    checkout = tmp_path / 'reference_checkout'                    # stands for the reference tree: models/{__init__,mdgat,superglue,pointnet/..}
    (checkout / 'models' / 'pointnet').mkdir(parents=True)
    (checkout / 'models' / '__init__.py').write_text('')
    (checkout / 'models' / 'mdgat.py').write_text('MDGAT = "the reference class"\n')
    (checkout / 'models' / 'superglue.py').write_text('SuperGlue = "the reference class"\n')
    (checkout / 'models' / 'pointnet' / '__init__.py').write_text('')
    (checkout / 'models' / 'pointnet' / 'pointnet_util.py').write_text('WHERE = "checkout"\n')
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'models' or k.startswith('models.')}
    sys.path.insert(0, str(checkout))
    sys.path.insert(0, shim)
    try:
        mod = importlib.import_module('models.mdgat')
        assert mod.MDGAT is MDGAT
        sg = importlib.import_module('models.superglue').SuperGlue
        assert issubclass(sg, MDGAT)
        net = sg(synth.default_config(L=2))                       # a k entry in the config is ignored: every layer fully connected
        assert net.k == [] and net._topk_schedule() == [0, 0, 0, 0]
        assert set(net.state_dict()) == set(MDGAT(synth.default_config(L=2)).state_dict())
        assert importlib.import_module('models.pointnet.pointnet_util').WHERE == 'checkout'
        for name in ('MLP', 'attention', 'dynamic_attention', 'log_optimal_transport', 'knn', 'get_graph_feature', 'match'):
            assert callable(getattr(mod, name)), name
        seq = mod.MLP([4, 32, 64, 128])
        assert [type(m).__name__ for m in seq] == ['Conv1d', 'BatchNorm1d', 'ReLU', 'Conv1d', 'BatchNorm1d', 'ReLU', 'Conv1d']
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            mod.log_optimal_transport(torch.zeros(1, 4, 4), 1.0, 3)         # CPU tensors: the product has no CPU path
    finally:
        sys.path.remove(shim)
        sys.path.remove(str(checkout))
        for k in [k for k in sys.modules if k == 'models' or k.startswith('models.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_deepcopy_and_pickle_drop_the_runtime_state():
    """The reference's MDGAT is a plain nn.Module: copy.deepcopy(net) and torch.save(net) work.  Here the module also owns
    library handles, locks and a packed blob - none of which may travel into a copy (a copied handle would be freed twice)."""
    import copy
    import io
    net = MDGAT(synth.default_config(L=1, k=[4, None])).double().eval()
    net.load_state_dict(synth.make_state_dict(L=1, seed=2))
    net._host_blob()                                        # runtime state exists
    net._states[0] = object()                               # (stands for a device state)
    for clone in (copy.deepcopy(net), torch.load(io.BytesIO(_saved(net)), weights_only=False)):
        assert clone is not net and clone._states == {} and clone._blob_holder == [None, False]
        assert clone._states is not net._states and clone._states_lock is not net._states_lock
        assert clone.config == net.config and not clone.training and clone.bin_score.dtype == torch.float64
        for (ka, va), (kb, vb) in zip(net.state_dict().items(), clone.state_dict().items()):
            assert ka == kb and torch.equal(va, vb) and va.data_ptr() != vb.data_ptr()
        np.testing.assert_array_equal(clone.packed_weights(), net.packed_weights())
    net._states.clear()


def _saved(net):
    import io
    buf = io.BytesIO()
    torch.save(net, buf)
    return buf.getvalue()


def test_two_device_dataparallel_with_stubbed_library():
    """torch.nn.DataParallel over two devices (test.py:158, train.py:192-196: replicate -> scatter -> parallel_apply in threads
    -> gather), with the library replaced by the recorder: the owner packs once when it is replicated, each device gets ONE
    handle loaded with the owner's blob (never a replica's own state), the two replicas run concurrently in threads on their own
    handle, and their outputs concatenate along dim 0 like DataParallel.gather does."""
    import threading
    from unittest import mock
    fake, stream = _FakeLib(), [7]
    L = 1
    net = MDGAT(synth.default_config(L=L, k=[])).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=6))
    net = net.eval()
    expect = net.packed_weights()
    d = synth.make_batch(4, 8, 8)
    keys = ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1')
    shards = [{k: d[k][2 * i:2 * i + 2] for k in keys} for i in range(2)]          # DataParallel.scatter: dim 0
    tl = threading.local()
    with _stubbed(fake, stream), \
            mock.patch.object(torch.Tensor, 'device', new=property(lambda self: torch.device('cuda', getattr(tl, 'dev', 0)))), \
            mock.patch.object(torch.Tensor, 'to', new=lambda self, *a, **k: self), \
            mock.patch.object(torch, 'empty', new=lambda *a, **k: torch.zeros(*a, **{kk: v for kk, v in k.items() if kk != 'device'})):
        packed_before = net._blob_holder[0] is not None
        replicas = [_emulate_replicate(net) for _ in range(2)]                      # the owner's _replicate_for_data_parallel hook runs
        assert not packed_before and net._blob_holder[0] is not None               # packed once, by the owner, when replicated
        outs, errs = [None, None], []

        def worker(i):
            try:
                tl.dev = i
                with torch.no_grad():
                    outs[i] = replicas[i]._run(*[shards[i][k] for k in keys])
            except Exception as e:                                                  # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        assert sorted(c for c in fake.calls if c[0] == 'create') == [('create', 0), ('create', 1)]
        assert len(fake.loaded) == 2 and all(np.array_equal(b, expect) for b in fake.loaded)
        assert sorted(net._states) == [0, 1] and all(r._states is net._states for r in replicas)
        # (a float64 module: the exact mode - each device's handle got the owner's fp64 blob as well)
        assert sum(1 for c in fake.calls if c[0] == 'forward_f64') == 2 and all(c[1] == 2 for c in fake.calls if c[0] == 'forward_f64')
        assert len(fake.loaded64) == 2 and all(np.array_equal(b, net.packed_weights(np.float64)) for b in fake.loaded64)
        gathered = torch.cat([o[0] for o in outs], dim=0)                           # DataParallel.gather of matches0
        assert gathered.shape == (4, 8)
        net._invalidate()


def test_forward_asks_about_its_own_token():
    """the "nothing matched" test of mdgat.py:465 must be about THIS call - forward() reads its call's token under the
    handle lock and asks mdgat_matched_any about that token, so a forward another thread enqueues on the same handle between this
    call's synchronisation and its question does not change the answer."""
    import threading
    fake, stream = _FakeLib(), [11]
    net = MDGAT(synth.default_config(L=1, k=[])).eval()
    d = synth.make_batch(2, 8, 8)
    with _stubbed(fake, stream):
        net(d)
        assert fake.calls[-2][0] == 'matched_any' or fake.calls[-1][0] == 'matched_any'
        asked = [c for c in fake.calls if c[0] == 'matched_any']
        assert asked[-1][1] == 1                                    # the first forward's token
        # an interleaved caller: a second thread's forward lands between the enqueue and the question of the first
        real_sync = torch.cuda.current_stream

        def interloper():
            args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
            net._run(*args)
        orig = MDGAT._matched_any

        def racing(self, device, token=0):
            t = threading.Thread(target=interloper)
            t.start(); t.join()                                     # token 3 is now the handle's last token
            return orig(self, device, token)
        fake.matched_by_token = {2: 0, 3: 1}
        from unittest import mock
        with mock.patch.object(MDGAT, '_matched_any', racing):
            out = net(d)                                            # token 2: nothing matched, whatever token 3 did
        assert out['matching_scores0'].dtype == torch.int64
        assert [c for c in fake.calls if c[0] == 'matched_any'][-1][1] == 2
        net._invalidate()


def test_fp64_arithmetic_host_logic():
    """arithmetic='fp64': the handle is created with MDGAT_ARITH_FP64, BOTH blobs are loaded (the fp64 one is the folded weights
    before their rounding), the inputs reach mdgat_forward_f64 as float64, and load_packed insists on the fp64 blob."""
    fake, stream = _FakeLib(), [11]
    cfgs = []
    orig_create = fake.mdgat_create

    def create(cfg, idx, handle):
        cfgs.append((cfg._obj.arithmetic, cfg._obj.f64_layers))
        return orig_create(cfg, idx, handle)
    fake.mdgat_create = create
    net = MDGAT(synth.default_config(L=2, arithmetic='fp64')).double()
    net.load_state_dict(synth.make_state_dict(L=2, seed=5))
    net = net.eval()
    d = synth.make_batch(2, 8, 8)
    with _stubbed(fake, stream):
        out = net(d)
        assert cfgs == [(_lib.ARITH_FP64, 0)]         # f64_layers 0 = automatic: what a zero-initialised C config holds
        assert [c[0] for c in fake.calls if c[0].startswith('load')] == ['load', 'load64']
        assert any(c[0] == 'forward_f64' for c in fake.calls) and not any(c[0] == 'forward' for c in fake.calls)
        b32, b64 = fake.loaded[0], fake.loaded64[0]
        assert b64.dtype == np.float64 and b32.shape == b64.shape
        np.testing.assert_array_equal(b32, b64.astype(np.float32))          # the fp32 blob is the rounding of the fp64 one
        assert np.abs(b64 - b32).max() > 0                                   # ... and the fp64 one carries more
        assert out['matches0'].dtype == torch.int64
        with pytest.raises(ValueError, match='blob64'):
            net.load_packed(torch.from_numpy(b32))
        net.load_packed(torch.from_numpy(b32), torch.from_numpy(b64))
        assert fake.calls[-1] == ('load64', 0) and fake.calls[-2] == ('load', 1)
        net._invalidate()
    with pytest.raises(ValueError):
        MDGAT(synth.default_config(L=1, arithmetic='fp16'))
    with pytest.raises(ValueError):
        MDGAT(synth.default_config(L=1, arithmetic='fp64', attention_dtype='f16'))


def test_module_dtype_is_the_arithmetic_request(monkeypatch):
    """No 'arithmetic' key (the reference's config has none, test.py:137-151): a float64 module - net.double(), test.py:193 - runs the
    reference-exact mode (MDGAT_ARITH_FP64 handle, both blobs, float64 inputs through mdgat_forward_f64), a float32 module the
    throughput path; the handle follows the module when it is cast; an explicit key or MDGAT_ARITHMETIC in the environment pins
    the path whatever the dtype; f64_layers reaches the C ABI as 0 = automatic / -1 = encoders only."""
    monkeypatch.delenv('MDGAT_ARITHMETIC', raising=False)
    fake, stream = _FakeLib(), [11]
    fake.matched = 1
    cfgs = []
    orig_create = fake.mdgat_create

    def create(cfg, idx, handle):
        cfgs.append((cfg._obj.arithmetic, cfg._obj.f64_layers))
        return orig_create(cfg, idx, handle)
    fake.mdgat_create = create
    d = synth.make_batch(1, 8, 8)
    sd = synth.make_state_dict(L=2, seed=5)
    with _stubbed(fake, stream):
        net = MDGAT(synth.default_config(L=2))
        assert net.arithmetic == 'auto' and not net.exact()
        net.load_state_dict(sd)
        net = net.eval()
        net({k: v.float() for k, v in d.items()})
        assert cfgs[-1] == (_lib.ARITH_FP32, 0)
        assert any(c[0] == 'forward' for c in fake.calls) and not any(c[0] == 'forward_f64' for c in fake.calls)
        net.double().eval()                                   # test.py:193
        assert net.exact()
        out = net(d)
        assert cfgs[-1] == (_lib.ARITH_FP64, 0)
        assert any(c[0] == 'forward_f64' for c in fake.calls) and ('load64', 0) in fake.calls
        assert out['matching_scores0'].dtype == torch.float64
        n_create = len(cfgs)
        net.double().eval()                                   # every iteration of the reference's loop: nothing is rebuilt
        net(d)
        assert len(cfgs) == n_create
        net.float()
        net({k: v.float() for k, v in d.items()})
        assert cfgs[-1] == (_lib.ARITH_FP32, 0) and len(cfgs) == n_create + 1
        net._invalidate()
        # pinned by the config, whatever the dtype
        for arith, want in (('fp32', _lib.ARITH_FP32), ('fp64', _lib.ARITH_FP64)):
            for cast in ('float', 'double'):
                p = MDGAT(synth.default_config(L=2, arithmetic=arith))
                p.load_state_dict(sd)
                p = getattr(p, cast)().eval()
                p(d)
                assert cfgs[-1][0] == want, (arith, cast)
                p._invalidate()
        # pinned by the environment for modules without the key; the key wins over the environment
        monkeypatch.setenv('MDGAT_ARITHMETIC', 'fp32')
        p = MDGAT(synth.default_config(L=2)).double().eval()
        assert p.arithmetic == 'fp32' and not p.exact()
        assert MDGAT(synth.default_config(L=2, arithmetic='fp64')).exact()
        monkeypatch.delenv('MDGAT_ARITHMETIC')
        # attention_dtype='f16' is a throughput request: a float64 module does not turn it into the exact mode
        assert not MDGAT(synth.default_config(L=2, attention_dtype='f16')).double().exact()
        # f64_layers: None / negative = automatic (0 in the C config), 0 = encoders only (-1), n = n
        for given, want in ((None, 0), (-1, 0), (0, _lib.F64_ENCODERS_ONLY), (3, 3)):
            over = {} if given is None else {'f64_layers': given}
            p = MDGAT(synth.default_config(L=2, **over)).double()
            p.load_state_dict(sd)
            p.eval()(d)
            assert cfgs[-1] == (_lib.ARITH_FP64, want), (given, cfgs[-1])
            p._invalidate()
