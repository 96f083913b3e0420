// Multi-head scaled-dot-product attention over keypoints, full (mdgat.py:190-194, attention) and
// dynamic (mdgat.py:196-210, dynamic_attention: keep the k largest logits of every row, softmax over
// those, zero elsewhere).  The N x M logits / probabilities are never written to memory.
//
// Arithmetic: fp32-equivalent products on the f16 matrix cores.  Every operand x is carried as two
// halves x = hi + lo / 2048 (hi = f16(x), lo = f16((x - hi) * 2048): 22 mantissa bits) and a product
// is three v_mfma_f32_32x32x16_f16: hi.hi into the main fp32 accumulator, hi.lo + lo.hi into a second
// one that is added with weight 1/2048.  The dropped lo.lo term is 2^-22 relative - the same class as
// the rounding of an fp32 FMA chain (tools/precision_probe.py: max|dZ| 1.3e-5 either way) at 16/3 times
// the rate of v_mfma_f32_32x32x2_f32.
//
// gfx950 mapping.  One workgroup = 4 waves owns one (pair, frame, head): the head's K rows (hi | lo
// halves, 144-byte padded rows) and V^T rows (keys contiguous, per plane) for all <= 512 keys are
// staged once in LDS (140 KB) and the workgroup loops over its 128-query tiles; a wave owns 32
// queries and computes S^T = K Q^T ("swapped" product): in the 32x32 C/D fragment layout a lane then
// holds, for ITS query (lane & 31), 16 logits per 32-key block - the row of a query lives in two lanes
// (l, l+32) x 16 registers per block, up to 256 registers for 512 keys (one wave per SIMD, 512-register
// budget).  Row max, row sum and the top-k count are register-local plus ONE lane^32 exchange.  The K
// rows are fed to the MFMA in a permuted order (bits 2 and 3 of the row index swapped) so that the 8
// accumulator registers of one half-block are 8 CONSECUTIVE keys: the probabilities, split to f16 in
// place, are then already the A operand of the P.V product whose B operand is one ds_read_b128 of V^T.
//
// Top-k: the exact k-th largest logit of a row is found by a per-row bracketing search on the
// threshold value t (count(s >= t) is monotone), see topk_threshold().  Masked softmax over "s >= t"
// equals softmax over the gathered top-k; exact ties at the k-th value are all kept (torch.topk would
// keep an arbitrary subset of them).
#include "common.hpp"

namespace {

constexpr int KROWH = 72;   // K row in LDS, halves: 32 hi | 32 lo | 8 pad (144 B = 9 x 16 B: conflict free)
constexpr int MAXBLK = 16;  // 16 x 32 = 512 keys

struct AttnArgs {
    const _Float16* q16;   // [B][P][4][2][32]   pre-scaled by log2(e)/sqrt(32)
    const _Float16* k16;   // [B][P][4][2][32]
    const _Float16* vt16;  // [B][4][2][32][PP]  keys of frame 0 at columns [0, N), frame 1 at [Npad, Npad + M); pads zero
    float* msg;            // [B][P][128]
    int N, M, Npad, PP, cross, topk;
};

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ int xor32i(int v) { return __shfl_xor(v, 32, 64); }

// x -> (f16 hi, f16 residual), unscaled residual: used for the probabilities, which are carried times
// 2048 so that the residual of every value that matters is a normal f16
__device__ __forceinline__ void split8(const float (&p)[8], f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)p[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = (_Float16)(p[j] - (float)h[j]);
}

template <int NBLK>
__device__ __forceinline__ float topk_threshold(const f32x16 (&S)[NBLK], float m, int k, int nk) {
    const float INF = __builtin_inff();
    float smin = INF;
#pragma unroll
    for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = S[jb][r];
            smin = fminf(smin, s == -INF ? INF : s);
        }
    smin = fminf(smin, xor32(smin));
    auto count_ge = [&](float t) {
        int c = 0;
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) c += (S[jb][r] >= t) ? 1 : 0;
        return c + xor32i(c);
    };
    float thr = -INF;
    float lo = smin, hv = m;
    int clo = nk, chi = count_ge(m);
    bool done = false;
    if (chi >= k) { thr = m; done = true; }          // ties at the maximum (or k == 1)
    if (nk <= k) { thr = smin; done = true; }        // this frame has exactly k keys: keep all
    for (int it = 0; it < 96; ++it) {
        if (__all(done)) break;
        const float mid = 0.5f * lo + 0.5f * hv;
        float t = mid;
        if (!(it & 1)) {
            const float frac = (float)(clo - k) / (float)(clo - chi);
            const float ti = lo + (hv - lo) * frac;
            if (ti > lo && ti < hv) t = ti;
        }
        const bool collapsed = !(t > lo && t < hv);   // no float strictly inside the bracket
        const int c = count_ge(t);
        if (!done) {
            if (collapsed) { thr = lo; done = true; }          // ties at the k-th value: keep them all
            else if (c == k) { thr = t; done = true; }
            else if (c > k) { lo = t; clo = c; }
            else { hv = t; chi = c; }
        }
    }
    if (!done) thr = lo;
    return thr;
}

// NBLK = compile-time bound on the number of 32-key blocks (registers for the row: 16 NBLK)
template <bool TOPK, int NBLK>
__global__ __launch_bounds__(256, NBLK <= 8 ? 2 : 1) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int head = blockIdx.y;
    const int b = blockIdx.z >> 1, side = blockIdx.z & 1;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N;
    const int q_off = side ? a.N : 0;
    const int src = a.cross ? (1 - side) : side;
    const int nk = src ? a.M : a.N;
    const int k_off = src ? a.N : 0;
    const int nblk = (nk + 31) >> 5;
    const int nkp = nblk * 32;
    const int VSTR = nkp + 8;               // V^T row stride in halves: (nkp + 8) / 8 is odd -> conflict free

    _Float16* Ks = smem;                    // [nkp][KROWH]
    _Float16* Vs = smem + nkp * KROWH;      // [2 planes][32 dims][VSTR]

    // ---- stage K (128 contiguous bytes per key) and V^T (nkp contiguous halves per (plane, dim)) ----
    {
        const _Float16* kg = a.k16 + (((size_t)b * P + k_off) * 4 + head) * 64;
        for (int idx = tid; idx < nkp * 8; idx += 256) {
            const int row = idx >> 3, c = idx & 7;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (row < nk) x = *reinterpret_cast<const f32x4*>(kg + (size_t)row * 256 + c * 8);
            *reinterpret_cast<f32x4*>(Ks + row * KROWH + c * 8) = x;
        }
        const _Float16* vg = a.vt16 + ((size_t)b * 4 + head) * 64 * a.PP + (src ? a.Npad : 0);
        const int cpr = nblk * 4;           // 16-byte chunks per row
        for (int idx = tid; idx < 64 * cpr; idx += 256) {
            const int row = idx / cpr, c = idx - row * cpr;
            *reinterpret_cast<f32x4*>(Vs + row * VSTR + c * 8) =
                *reinterpret_cast<const f32x4*>(vg + (size_t)row * a.PP + c * 8);
        }
    }
    __syncthreads();

    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2 <-> 3
    const float NEG_INF = -__builtin_inff();
    // only the last block can be partially valid: register r of half `hi` holds key 16 (r >> 3) + 8 hi + (r & 7)
    const int last_lim = nk - (nblk - 1) * 32 - 8 * hi;
    const bool last_partial = (nk & 31) != 0;

    for (int q0 = blockIdx.x * 128; q0 < nq; q0 += gridDim.x * 128) {
        const int qw = q0 + wave * 32;
        if (qw >= nq) continue;             // wave-uniform; no barrier inside the loop

        // ---- this lane's query fragments: dims 16 t + 8 hi + j of query l31, planes hi / lo ----
        f16x8 qh[2], ql[2];
        {
            const int qrow = min(qw + l31, nq - 1);
            const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * hi;
            qh[0] = *reinterpret_cast<const f16x8*>(p);
            qh[1] = *reinterpret_cast<const f16x8*>(p + 16);
            ql[0] = *reinterpret_cast<const f16x8*>(p + 32);
            ql[1] = *reinterpret_cast<const f16x8*>(p + 48);
        }

        // ---- S^T = K Q^T, whole row resident in registers ----
        f32x16 S[NBLK];
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
            if (jb < nblk) {
                const _Float16* kp = Ks + (jb * 32 + krow) * KROWH + 8 * hi;
                const f16x8 kh0 = *reinterpret_cast<const f16x8*>(kp);
                const f16x8 kh1 = *reinterpret_cast<const f16x8*>(kp + 16);
                const f16x8 kl0 = *reinterpret_cast<const f16x8*>(kp + 32);
                const f16x8 kl1 = *reinterpret_cast<const f16x8*>(kp + 48);
                f32x16 acc, acx;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acx[r] = 0.f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, qh[0], acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, ql[0], acx, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, qh[1], acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, ql[1], acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, qh[0], acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, qh[1], acx, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaf(acx[r], MDGAT_SPLIT_INV, acc[r]);
                if (last_partial && jb == nblk - 1) {   // wave-uniform
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (16 * (r >> 3) + (r & 7) >= last_lim) acc[r] = NEG_INF;
                }
                S[jb] = acc;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[jb][r] = NEG_INF;
            }
        }

        // ---- row max ----
        float m = NEG_INF;
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, S[jb][r]);
        m = fmaxf(m, xor32(m));

        // ---- exact top-k threshold (dynamic layers only) ----
        float thr = NEG_INF;
        if (TOPK) thr = topk_threshold<NBLK>(S, m, a.topk, nk);

        // ---- P' = 2048 exp2(s - m) split to f16 in place, row sum, O = P' V ----
        const float m11 = m - 11.0f;
        float l = 0.f;
        f32x16 Om, Ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Om[r] = 0.f; Ox[r] = 0.f; }
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
            if (jb < nblk) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float p[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float s = S[jb][8 * t + j];
                        float e = __builtin_amdgcn_exp2f(s - m11);
                        if (TOPK) e = (s >= thr) ? e : 0.f;
                        p[j] = e;
                        l += e;
                    }
                    f16x8 ph, pl;
                    split8(p, ph, pl);
                    const _Float16* vp = Vs + l31 * VSTR + jb * 32 + t * 16 + 8 * hi;
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(vp);
                    const f16x8 vl = *reinterpret_cast<const f16x8*>(vp + 32 * VSTR);
                    Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, Om, 0, 0, 0);
                    Ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, Ox, 0, 0, 0);
                    Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, Om, 0, 0, 0);
                }
            }
        }
        l += xor32(l);
        const float inv_l = 1.0f / l;

        // ---- message rows: lane holds column (dim) l31 of queries mfma32_row(r, hi) ----
        float* out = a.msg + ((size_t)b * P + q_off) * 128 + head * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, hi);
            const float inv = __shfl(inv_l, row, 64);
            const int q = qw + row;
            if (q < nq) out[(size_t)q * 128] = fmaf(Ox[r], MDGAT_SPLIT_INV, Om[r]) * inv;
        }
    }
}

// fp32 q/k/v [B][P][3][4][32] -> the split-f16 operand layouts above (per-op entry point and tests; the
// forward's q/k/v projection writes these layouts directly from its epilogue)
__global__ __launch_bounds__(256) void qkv_split_kernel(const float* qkv, _Float16* q16, _Float16* k16, _Float16* vt16,
                                                         int B, int N, int M, int Npad, int PP) {
    const int P = N + M;
    const size_t total = (size_t)B * PP * 128;   // one thread per (b, padded column, head*32 + dim)
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx & 127);
        const size_t bc = idx >> 7;
        const int col = (int)(bc % PP), b = (int)(bc / PP);
        const int head = c >> 5, d = c & 31;
        int p = -1;
        if (col < Npad) { if (col < N) p = col; }
        else if (col - Npad < M) p = N + col - Npad;
        _Float16* vt = vt16 + (((size_t)b * 4 + head) * 2 * 32 + d) * PP + col;
        if (p < 0) { vt[0] = (_Float16)0.f; vt[(size_t)32 * PP] = (_Float16)0.f; continue; }
        const float* src = qkv + ((size_t)b * P + p) * 384 + c;
        const float q = src[0] * (MDGAT_LOG2E * 0.17677669529663687f), k = src[128], v = src[256];
        _Float16 h, l;
        const size_t o = (((size_t)b * P + p) * 4 + head) * 64 + d;
        mdgat_split(q, h, l); q16[o] = h; q16[o + 32] = l;
        mdgat_split(k, h, l); k16[o] = h; k16[o + 32] = l;
        mdgat_split(v, h, l); vt[0] = h; vt[(size_t)32 * PP] = l;
    }
}

}  // namespace

size_t mdgat_qkv16_halves(int B, int N, int M) {
    const int Npad = (N + 31) & ~31, Mpad = (M + 31) & ~31;
    return (size_t)B * (N + M) * 256 * 2 + (size_t)B * 256 * (Npad + Mpad);
}

Qkv16 mdgat_qkv16_carve(_Float16* base, int B, int N, int M) {
    Qkv16 q{};
    q.Npad = (N + 31) & ~31;
    q.PP = q.Npad + ((M + 31) & ~31);
    q.q16 = base;
    q.k16 = base + (size_t)B * (N + M) * 256;
    q.vt16 = q.k16 + (size_t)B * (N + M) * 256;
    return q;
}

int launch_qkv_split(int B, int N, int M, const float* qkv, const Qkv16& o, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    const size_t total = (size_t)B * o.PP * 128;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(qkv_split_kernel, dim3(blocks), dim3(256), 0, s, qkv, o.q16, o.k16, o.vt16, B, N, M, o.Npad, o.PP);
    return mdgat_check_hip(hipGetLastError(), "qkv split launch");
}

int launch_attention(int B, int N, int M, int cross, int topk, const Qkv16& qkv, float* msg, hipStream_t s) {
    if (B <= 0 || N <= 0 || M <= 0) return MDGAT_OK;
    const int nk_max = N > M ? N : M;
    if (nk_max > MAXBLK * 32) {
        mdgat_set_error("attention: %d keys > %d supported by the register-resident kernel", nk_max, MAXBLK * 32);
        return MDGAT_ERR_UNSUPPORTED;
    }
    if (topk > 0) {
        // torch.topk raises when k exceeds the number of keys of either direction (mdgat.py:202)
        const int nk_min = N < M ? N : M;
        if (topk > nk_min) {
            mdgat_set_error("dynamic attention: k=%d exceeds the number of keys (%d)", topk, nk_min);
            return MDGAT_ERR_BAD_ARG;
        }
    }
    AttnArgs a{qkv.q16, qkv.k16, qkv.vt16, msg, N, M, qkv.Npad, qkv.PP, cross, topk};
    const int nkp = ((nk_max + 31) / 32) * 32;
    const size_t lds = ((size_t)nkp * KROWH + (size_t)64 * (nkp + 8)) * sizeof(_Float16);
    // one workgroup per (pair, frame, head) loops over its query tiles; split the tiles over more
    // workgroups only when there are too few (pair, frame, head) units to fill the chip twice
    const int qtiles = (nk_max + 127) / 128;
    int qsplit = (512 + B * 2 * MDGAT_HEADS - 1) / (B * 2 * MDGAT_HEADS);
    if (qsplit > qtiles) qsplit = qtiles;
    if (qsplit < 1) qsplit = 1;
    dim3 grid(qsplit, MDGAT_HEADS, B * 2);
    // k == number of keys on both sides keeps every key: identical to full attention
    const bool dyn = topk > 0 && !(topk == N && topk == M);
    const int nblk = nkp / 32;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    };
    if (dyn) {
        if (nblk <= 4) go(attention_kernel<true, 4>);
        else if (nblk <= 8) go(attention_kernel<true, 8>);
        else go(attention_kernel<true, 16>);
    } else {
        if (nblk <= 4) go(attention_kernel<false, 4>);
        else if (nblk <= 8) go(attention_kernel<false, 8>);
        else go(attention_kernel<false, 16>);
    }
    return mdgat_check_hip(hipGetLastError(), "attention launch");
}
