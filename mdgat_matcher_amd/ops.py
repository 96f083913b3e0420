"""Per-op Python bindings over the C ABI (torch tensors in, torch tensors out; all on a gfx950 device).

These call the same kernels ``mdgat_forward`` launches; they exist for unit parity tests and for
callers that want a single stage (e.g. Sinkhorn on their own score matrix)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('mdgat_matcher_amd ops run on MI355X only (no CPU fallback)')


def sinkhorn(scores: torch.Tensor, bin_score: float, iters: int, streaming: bool = False) -> torch.Tensor:
    """log_optimal_transport (mdgat.py:288-308): scores [B, N, M] -> Z [B, N+1, M+1] (fp32).

    N, M <= 2048 run on the register-resident cluster kernel (128 x 512 tiles per workgroup; needs a small workspace,
    allocated here); ``streaming=True`` selects the one-workgroup-per-pair streaming kernel."""
    _need_cuda(scores)
    s = scores.to(torch.float32).contiguous()
    B, N, M = s.shape
    Z = torch.empty((B, N + 1, M + 1), dtype=torch.float32, device=s.device)
    lib = _lib.load()
    with torch.cuda.device(s.device):
        need = 0 if streaming else lib.mdgat_sinkhorn_workspace_bytes(B, N, M)
        ws = torch.empty(need, dtype=torch.uint8, device=s.device) if need else None
        _lib.check(lib.mdgat_sinkhorn(B, N, M, s.data_ptr(), float(bin_score), int(iters), Z.data_ptr(),
                                      ws.data_ptr() if ws is not None else None, need, _stream(s)), 'mdgat_sinkhorn')
    return Z


def sinkhorn_f64(scores: torch.Tensor, bin_score: float, iters: int) -> torch.Tensor:
    """log_optimal_transport (mdgat.py:288-308) in fp64 (csrc/sinkhorn_f64.hip): scores [B, N, M] float64 -> Z [B, N+1, M+1] float64."""
    _need_cuda(scores)
    s = scores.to(torch.float64).contiguous()
    B, N, M = s.shape
    Z = torch.empty((B, N + 1, M + 1), dtype=torch.float64, device=s.device)
    lib = _lib.load()
    with torch.cuda.device(s.device):
        need = lib.mdgat_sinkhorn_f64_workspace_bytes(B, N, M)
        ws = torch.empty(need + 256, dtype=torch.uint8, device=s.device)
        off = (-ws.data_ptr()) % 256
        _lib.check(lib.mdgat_sinkhorn_f64(B, N, M, s.data_ptr(), float(bin_score), int(iters), Z.data_ptr(), ws.data_ptr() + off, need, _stream(s)),
                   'mdgat_sinkhorn_f64')
    return Z


def sinkhorn_f64_extract(scores: torch.Tensor, bin_score: float, iters: int, mode: int = _lib.EXTRACT_DUSTBIN, match_threshold: float = 0.2,
                         want_Z: bool = False):
    """fp64 Sinkhorn + match extraction with every arg-max decided on the fp64 Z: (matches0, matches1, mscores0, mscores1[, Z fp32])."""
    _need_cuda(scores)
    s = scores.to(torch.float64).contiguous()
    B, N, M = s.shape
    m0 = torch.empty((B, N), dtype=torch.int64, device=s.device)
    m1 = torch.empty((B, M), dtype=torch.int64, device=s.device)
    s0 = torch.empty((B, N), dtype=torch.float32, device=s.device)
    s1 = torch.empty((B, M), dtype=torch.float32, device=s.device)
    Z = torch.empty((B, N + 1, M + 1), dtype=torch.float32, device=s.device) if want_Z else None
    lib = _lib.load()
    with torch.cuda.device(s.device):
        need = lib.mdgat_sinkhorn_f64_workspace_bytes(B, N, M)
        ws = torch.empty(need + 256, dtype=torch.uint8, device=s.device)
        off = (-ws.data_ptr()) % 256
        _lib.check(lib.mdgat_sinkhorn_f64_extract(B, N, M, s.data_ptr(), float(bin_score), int(iters), int(mode), float(match_threshold), m0.data_ptr(),
                                                  m1.data_ptr(), s0.data_ptr(), s1.data_ptr(), Z.data_ptr() if Z is not None else None,
                                                  ws.data_ptr() + off, need, _stream(s)), 'mdgat_sinkhorn_f64_extract')
    return (m0, m1, s0, s1, Z) if want_Z else (m0, m1, s0, s1)


def extract(Z: torch.Tensor, mode: int = _lib.EXTRACT_DUSTBIN, match_threshold: float = 0.2):
    """Match extraction (mdgat.py:441-483) from Z [B, N+1, M+1]."""
    _need_cuda(Z)
    z = Z.to(torch.float32).contiguous()
    B, N, M = z.shape[0], z.shape[1] - 1, z.shape[2] - 1
    m0 = torch.empty((B, N), dtype=torch.int64, device=z.device)
    m1 = torch.empty((B, M), dtype=torch.int64, device=z.device)
    s0 = torch.empty((B, N), dtype=torch.float32, device=z.device)
    s1 = torch.empty((B, M), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().mdgat_extract(B, N, M, z.data_ptr(), int(mode), float(match_threshold), m0.data_ptr(),
                                             m1.data_ptr(), s0.data_ptr(), s1.data_ptr(), _stream(z)), 'mdgat_extract')
    return m0, m1, s0, s1


def topk_sel_words(B: int, N: int, M: int) -> int:
    """uint32 words of one layer's slice of the top-k selection tap (``mdgat_taps.topk_sel``)."""
    return int(_lib.load().mdgat_topk_sel_words(B, N, M))


def topk_sel_to_masks(sel: torch.Tensor, B: int, N: int, M: int, cross: bool):
    """Unpack one layer's selection tap (int32 words [B][4][N+M][W]) into boolean masks
    ``(mask0 [B, 4, N, keys of frame 0's source], mask1 [B, 4, M, keys of frame 1's source])``: True where the dynamic
    layer kept the key (the index set of ``logits.topk(k)``, mdgat.py:202)."""
    W = (max(N, M) + 31) // 32
    w = sel.reshape(B, 4, N + M, W).to(torch.int64) & 0xFFFFFFFF
    bits = ((w[..., None] >> torch.arange(32, device=w.device)) & 1).bool().reshape(B, 4, N + M, W * 32)
    nk0, nk1 = (M, N) if cross else (N, M)
    return bits[:, :, :N, :nk0], bits[:, :, N:, :nk1]


def attention(qkv: torch.Tensor, N: int, M: int, cross: bool, topk: int = 0, return_selection: bool = False):
    """attention / dynamic_attention (mdgat.py:190-210).  qkv [B, N+M, 3, 4, 32] -> message [B, N+M, 128]
    (with ``return_selection``: also the boolean masks of the keys a dynamic layer kept, see topk_sel_to_masks)."""
    _need_cuda(qkv)
    x = qkv.to(torch.float32).contiguous()
    B, P = x.shape[0], x.shape[1]
    assert P == N + M and tuple(x.shape[2:]) == (3, 4, 32)
    msg = torch.empty((B, P, 128), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        need = lib.mdgat_attention_workspace_bytes(B, N, M)
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        sel = torch.empty(topk_sel_words(B, N, M), dtype=torch.int32, device=x.device) if return_selection else None
        _lib.check(lib.mdgat_attention_sel(B, N, M, int(bool(cross)), int(topk), x.data_ptr(), msg.data_ptr(),
                                           sel.data_ptr() if sel is not None else None,
                                           ws.data_ptr(), need, _stream(x)), 'mdgat_attention')
    if return_selection:
        return msg, topk_sel_to_masks(sel, B, N, M, cross)
    return msg


class QkProbe:
    """Measurement only (bench.py ``roofline_qk``): the Q K^T phase of the streamed full-attention kernel in isolation
    (``mdgat_attention_qk_probe``).  ``QkProbe(qkv, N, M)`` converts fp32 q/k/v [B, N+M, 3, 4, 32] to the library's split
    layout once; ``run(cross)`` launches only the probe kernel on the current stream."""

    def __init__(self, qkv: torch.Tensor, N: int, M: int):
        _need_cuda(qkv)
        x = qkv.to(torch.float32).contiguous()
        self.B, self.N, self.M = x.shape[0], N, M
        assert x.shape[1] == N + M and tuple(x.shape[2:]) == (3, 4, 32)
        self.lib = _lib.load()
        with torch.cuda.device(x.device):
            self.need = self.lib.mdgat_attention_workspace_bytes(self.B, N, M)
            self.ws = torch.empty(self.need, dtype=torch.uint8, device=x.device)
            self.msg = torch.zeros((self.B, N + M, 128), dtype=torch.float32, device=x.device)
            _lib.check(self.lib.mdgat_attention_qk_probe(self.B, N, M, 0, x.data_ptr(), self.msg.data_ptr(), self.ws.data_ptr(),
                                                         self.need, _stream(x)), 'mdgat_attention_qk_probe')

    def run(self, cross: bool = False, nq_sets: int = 0):
        """nq_sets = 0: the Q K^T phase of the shipped kernel (softmax and P.V knocked out); 1 / 2: the standalone phase kernel
        with 32 / 64 queries per wave (``mdgat_attention_qk_probe_sets``)."""
        with torch.cuda.device(self.ws.device):
            if nq_sets:
                _lib.check(self.lib.mdgat_attention_qk_probe_sets(self.B, self.N, self.M, int(bool(cross)), int(nq_sets), self.ws.data_ptr(),
                                                                  self.msg.data_ptr(), self.ws.data_ptr(), self.need, _stream(self.ws)),
                           'mdgat_attention_qk_probe_sets')
            else:
                _lib.check(self.lib.mdgat_attention_qk_probe(self.B, self.N, self.M, int(bool(cross)), self.ws.data_ptr(),
                                                             self.msg.data_ptr(), self.ws.data_ptr(), self.need, _stream(self.ws)),
                           'mdgat_attention_qk_probe')
        return self.msg


def mfma_sustained(device, reps: int = 4000) -> dict:
    """Measurement only (bench.py ``roofline.sustained_*``): the f16 MFMA rate ``device`` sustains on random operands
    (``mdgat_mfma_probe``: nothing but matrix instructions, two waves per SIMD) and the shader clock it runs at meanwhile."""
    import ctypes as C
    device = torch.device(device)
    lib = _lib.load()
    with torch.cuda.device(device):
        ws = torch.empty(1 << 18, dtype=torch.uint8, device=device)
        ms, flops, ticks = C.c_float(), C.c_double(), C.c_longlong()
        _lib.check(lib.mdgat_mfma_probe(reps, ws.data_ptr(), ws.numel(), C.byref(ms), C.byref(flops), C.byref(ticks),
                                        torch.cuda.current_stream(device).cuda_stream), 'mdgat_mfma_probe')
    return {'tflops': flops.value / (ms.value * 1e-3) / 1e12, 'clock_ghz': ticks.value / (ms.value * 1e6),
            'ms': ms.value, 'ticks_per_mfma_per_simd': ticks.value / (24.0 * reps * 2)}


def pointwise(A: torch.Tensor, W: torch.Tensor, bias=None, relu=False, residual=None) -> torch.Tensor:
    """Conv1d(k=1) over points: A [rows, K] x W [Cout, K]^T (+bias, ReLU, +residual) -> [rows, Cout]."""
    _need_cuda(A, W)
    a = A.to(torch.float32).contiguous()
    w = W.to(torch.float32).contiguous()
    rows, K = a.shape
    cout = w.shape[0]
    assert w.shape[1] == K
    b = bias.to(torch.float32).contiguous() if bias is not None else None
    r = residual.to(torch.float32).contiguous() if residual is not None else None
    out = torch.empty((rows, cout), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().mdgat_pointwise(rows, cout, K, a.data_ptr(), K, w.data_ptr(), K,
                                               b.data_ptr() if b is not None else None, int(relu),
                                               r.data_ptr() if r is not None else None, cout, out.data_ptr(), cout,
                                               _stream(a)), 'mdgat_pointwise')
    return out


def knn(x: torch.Tensor, src: torch.Tensor, k: int, adjacency: bool = False, mfma: bool = True):
    """knn / get_graph_feature (mdgat.py:8-32).  Channel-major inputs like the reference: x [B, C, N], src [B, C, M].
    ``mfma=False`` keeps C = 128 off the matrix cores (distances computed inside the selection kernel)."""
    _need_cuda(x, src)
    xp = x.to(torch.float32).transpose(1, 2).contiguous()
    sp = src.to(torch.float32).transpose(1, 2).contiguous()
    B, N, Cc = xp.shape
    M = sp.shape[1]
    idx = torch.empty((B, N, k), dtype=torch.int64, device=x.device)
    adj = torch.empty((B, N, M), dtype=torch.int64, device=x.device) if adjacency else None
    lib = _lib.load()
    with torch.cuda.device(x.device):
        need = lib.mdgat_knn_workspace_bytes(B, Cc, N, M) if mfma else 0      # C == 128: inner products on the matrix cores
        ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
        _lib.check(lib.mdgat_knn(B, Cc, N, M, int(k), xp.data_ptr(), sp.data_ptr(), idx.data_ptr(),
                                 adj.data_ptr() if adj is not None else None,
                                 ws.data_ptr() if ws is not None else None, need, _stream(x)), 'mdgat_knn')
    return (idx, adj) if adjacency else idx


def pose_from_matches(kpts0: torch.Tensor, kpts1: torch.Tensor, matches0: torch.Tensor, T_gt=None, inlier_dist: float = 1.0):
    """solve_icp + calculate_error (utils/utils_test.py:41-110) for a batch: kpts [B, N, 3] / [B, M, 3], matches0
    [B, N] int64 (-1 = unmatched).  Returns (T [B, 4, 4] float64 mapping frame 1 onto frame 0, stats [B, 5] float64 =
    matches, inliers, inlier ratio, translation error, rotation error; the errors are NaN without ``T_gt``)."""
    _need_cuda(kpts0, kpts1, matches0)
    k0 = kpts0.to(torch.float32).contiguous()
    k1 = kpts1.to(torch.float32).contiguous()
    m0 = matches0.to(torch.int64).contiguous()
    B, N, M = k0.shape[0], k0.shape[1], k1.shape[1]
    T = torch.empty((B, 4, 4), dtype=torch.float64, device=k0.device)
    stats = torch.empty((B, 5), dtype=torch.float64, device=k0.device)
    g = T_gt.to(device=k0.device, dtype=torch.float64).contiguous() if T_gt is not None else None
    with torch.cuda.device(k0.device):
        _lib.check(_lib.load().mdgat_pose(B, N, M, k0.data_ptr(), k1.data_ptr(), m0.data_ptr(),
                                          g.data_ptr() if g is not None else None, float(inlier_dist), T.data_ptr(),
                                          stats.data_ptr(), _stream(k0)), 'mdgat_pose')
    return T, stats


def gt_matches(kpts0: torch.Tensor, kpts1: torch.Tensor, T0=None, T1=None, threshold: float = 0.5, mutual: bool = False):
    """Ground-truth matches of the loader (load_data.py:238-285): kpts [B, N, 3] / [B, M, 3] in the sensor frame,
    T0 / T1 [B, 4, 4] float64 sensor -> world (None = identity).  Returns (gt_matches0 [B, N], gt_matches1 [B, M],
    rep [B]) as int64, -1 = no match."""
    _need_cuda(kpts0, kpts1)
    k0 = kpts0.to(torch.float32).contiguous()
    k1 = kpts1.to(torch.float32).contiguous()
    B, N, M = k0.shape[0], k0.shape[1], k1.shape[1]
    g0 = torch.empty((B, N), dtype=torch.int64, device=k0.device)
    g1 = torch.empty((B, M), dtype=torch.int64, device=k0.device)
    rep = torch.empty((B,), dtype=torch.int64, device=k0.device)
    t0 = T0.to(device=k0.device, dtype=torch.float64).contiguous() if T0 is not None else None
    t1 = T1.to(device=k0.device, dtype=torch.float64).contiguous() if T1 is not None else None
    with torch.cuda.device(k0.device):
        _lib.check(_lib.load().mdgat_gt_matches(B, N, M, k0.data_ptr(), k1.data_ptr(),
                                                t0.data_ptr() if t0 is not None else None,
                                                t1.data_ptr() if t1 is not None else None, float(threshold), int(bool(mutual)),
                                                g0.data_ptr(), g1.data_ptr(), rep.data_ptr(), _stream(k0)), 'mdgat_gt_matches')
    return g0, g1, rep


# ---- fp64 kernels of the reference-exact mode (csrc/f64.hip; MDGAT(arithmetic='fp64') launches the same ones) ----
def pointwise_f64(a: torch.Tensor, w: torch.Tensor, bias=None, relu: bool = False, residual=None) -> torch.Tensor:
    """Conv1d(k=1) over points in fp64 (mdgat.py:34-46 after BN folding): a [M, K] x w [N, K]^T (+ bias)(ReLU)(+ residual)."""
    _need_cuda(a, w)
    a = a.to(torch.float64).contiguous()
    w = w.to(torch.float64).contiguous()
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    bias = bias.to(torch.float64).contiguous() if bias is not None else None
    residual = residual.to(torch.float64).contiguous() if residual is not None else None
    out = torch.empty((M, N), dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().mdgat_pointwise_f64(M, N, K, a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                                   int(bool(relu)), residual.data_ptr() if residual is not None else None, N,
                                                   out.data_ptr(), N, _stream(a)), 'mdgat_pointwise_f64')
    return out


def attention_f64(qkv: torch.Tensor, N: int, M: int, cross: bool, topk: int = 0, return_selection: bool = False):
    """attention / dynamic_attention (mdgat.py:190-210) in fp64.  qkv [B, N+M, 3, 4, 32] float64 -> message [B, N+M, 128] float64
    (with ``return_selection``: also the masks of the keys a dynamic layer kept, see topk_sel_to_masks)."""
    _need_cuda(qkv)
    x = qkv.to(torch.float64).contiguous()
    B, P = x.shape[0], x.shape[1]
    assert P == N + M and tuple(x.shape[2:]) == (3, 4, 32)
    msg = torch.empty((B, P, 128), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        sel = torch.empty(topk_sel_words(B, N, M), dtype=torch.int32, device=x.device) if return_selection else None
        _lib.check(_lib.load().mdgat_attention_f64(B, N, M, int(bool(cross)), int(topk), x.data_ptr(), msg.data_ptr(),
                                                   sel.data_ptr() if sel is not None else None, _stream(x)), 'mdgat_attention_f64')
    if return_selection:
        return msg, topk_sel_to_masks(sel, B, N, M, cross)
    return msg


def mfma_f64_probe(device, reps: int = 2000):
    """Measurement only: (ms, flops, shader ticks) of the v_mfma_f64_16x16x4_f64 probe loop on ``device``."""
    dev = torch.device(device)
    ws = torch.empty(1 << 17, dtype=torch.uint8, device=dev)
    ms, fl, tk = C.c_float(0), C.c_double(0), C.c_longlong(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mdgat_mfma_f64_probe(int(reps), ws.data_ptr(), ws.numel(), C.byref(ms), C.byref(fl), C.byref(tk),
                                                    torch.cuda.current_stream(dev).cuda_stream), 'mdgat_mfma_f64_probe')
    return ms.value, fl.value, tk.value
