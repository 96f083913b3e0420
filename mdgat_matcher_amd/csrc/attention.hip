// Multi-head scaled-dot-product attention over keypoints, full (mdgat.py:190-194, attention) and
// dynamic (mdgat.py:196-210, dynamic_attention: keep the k largest logits of every row, softmax over
// those, zero elsewhere).  The N x M logits / probabilities are never written to memory.
//
// gfx950 mapping.  One workgroup = 4 waves = 128 queries of one (pair, frame, head); the head's whole
// K [M][32] and V [M][32] tiles (<= 512 keys) are staged once in LDS (K rows padded to 36 dwords so the
// ds_read_b128 fragment reads are bank-conflict free, V rows dense: its ds_read_b32 fragment reads hit
// 32 consecutive banks).  A wave owns 32 queries and computes S^T = K Q^T with v_mfma_f32_32x32x2_f32
// ("swapped" product): in the C/D fragment layout a lane then holds, for ITS query (lane & 31), the
// logits of 16 keys per 32-key block - the row of a query lives in two lanes (l, l+32) x 16 registers
// per block, i.e. up to 256 registers for 512 keys (one wave per SIMD, 512-register budget).  Row max,
// row sum and the top-k count are therefore register-local plus ONE lane^32 exchange, and the
// probabilities are already in the A-operand layout of the P.V product (k-slot (r, hi) <-> key
// mfma32_row(r, hi); V's B-operand rows are read from LDS in that same order), so P never moves.
//
// Top-k: the exact k-th largest logit of a row is found by a per-row bracketing search on the
// threshold value t (count(s >= t) is monotone): interpolation steps alternate with bisection steps,
// each step is 256 compare+add per lane; it stops when count == k (or when no float lies between the
// bracket ends, i.e. exact ties at the k-th value, which are then all kept - torch.topk would keep an
// arbitrary subset of them).  Masked softmax over "s >= t" equals softmax over the gathered top-k.
#include "common.hpp"

namespace {

constexpr int KROW = 36;   // padded K row (dwords)
constexpr int MAXBLK = 16; // 16 x 32 = 512 keys

struct AttnArgs {
    const float* qkv;   // [B][P][3][4][32]
    float* msg;         // [B][P][128]
    int N, M, cross, topk;
};

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ int xor32i(int v) { return __shfl_xor(v, 32, 64); }

template <bool TOPK>
__global__ __launch_bounds__(256, 1) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
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
    const int q0 = blockIdx.x * 128;
    if (q0 >= nq) return;
    const int nblk = (nk + 31) >> 5;

    float* Ks = smem;                       // [nblk*32][KROW]
    float* Vs = smem + nblk * 32 * KROW;    // [nblk*32][32]

    // ---- stage K and V of this head (coalesced 16-byte loads; rows beyond nk are zero-filled) ----
    {
        const float* kv_base = a.qkv + ((size_t)b * P + k_off) * 384 + head * 32;
        const int nrow = nblk * 32;
        for (int idx = tid; idx < nrow * 8; idx += 256) {
            const int row = idx >> 3, c4 = (idx & 7) * 4;
            f32x4 kx = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
            if (row < nk) {
                const float* p = kv_base + (size_t)row * 384 + c4;
                kx = *reinterpret_cast<const f32x4*>(p + 128);
                vx = *reinterpret_cast<const f32x4*>(p + 256);
            }
            *reinterpret_cast<f32x4*>(Ks + row * KROW + c4) = kx;
            *reinterpret_cast<f32x4*>(Vs + row * 32 + c4) = vx;
        }
    }

    // ---- this lane's query fragment: Q[q][16*hi + t], pre-scaled by log2(e)/sqrt(32) ----
    const int qw = q0 + wave * 32;
    float qf[16];
    {
        const int qrow = min(qw + l31, nq - 1);
        const float* p = a.qkv + ((size_t)b * P + q_off + qrow) * 384 + head * 32 + hi * 16;
        const float sc = MDGAT_LOG2E * 0.17677669529663687f;   // log2(e) / sqrt(32)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * v);
            qf[4 * v + 0] = x[0] * sc; qf[4 * v + 1] = x[1] * sc; qf[4 * v + 2] = x[2] * sc; qf[4 * v + 3] = x[3] * sc;
        }
    }
    __syncthreads();
    if (qw >= nq) return;   // wave-uniform; no barrier follows

    // ---- S^T = K Q^T, whole row resident in registers ----
    f32x16 S[MAXBLK];
    const float NEG_INF = -__builtin_inff();
#pragma unroll
    for (int jb = 0; jb < MAXBLK; ++jb) {
        if (jb < nblk) {
            float kf[16];
            const float* kp = Ks + (jb * 32 + l31) * KROW + hi * 16;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                f32x4 x = *reinterpret_cast<const f32x4*>(kp + 4 * v);
                kf[4 * v + 0] = x[0]; kf[4 * v + 1] = x[1]; kf[4 * v + 2] = x[2]; kf[4 * v + 3] = x[3];
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], acc, 0, 0, 0);
            if (jb * 32 + 32 > nk) {   // partially valid block (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (jb * 32 + mfma32_row(r, hi) >= nk) acc[r] = NEG_INF;
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
    for (int jb = 0; jb < MAXBLK; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, S[jb][r]);
    m = fmaxf(m, xor32(m));

    // ---- exact top-k threshold (dynamic layers only) ----
    float thr = NEG_INF;
    if (TOPK) {
        const int k = a.topk;
        float smin = __builtin_inff();
#pragma unroll
        for (int jb = 0; jb < MAXBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = S[jb][r];
                smin = fminf(smin, s == NEG_INF ? __builtin_inff() : s);
            }
        smin = fminf(smin, xor32(smin));
        auto count_ge = [&](float t) {
            int c = 0;
#pragma unroll
            for (int jb = 0; jb < MAXBLK; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) c += (S[jb][r] >= t) ? 1 : 0;
            return c + xor32i(c);
        };
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
    }

    // ---- softmax numerators, row sum, normalisation (base-2 exponent: logits carry log2(e)) ----
    float l = 0.f;
#pragma unroll
    for (int jb = 0; jb < MAXBLK; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = S[jb][r];
            float p = __builtin_amdgcn_exp2f(s - m);
            if (TOPK) p = (s >= thr) ? p : 0.f;
            S[jb][r] = p;
            l += p;
        }
    l += xor32(l);
    const float inv_l = 1.0f / l;

    // ---- O = P V ----
    f32x16 O;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
    for (int jb = 0; jb < MAXBLK; ++jb) {
        if (jb < nblk) {
            const float* vp = Vs + (jb * 32 + 4 * hi) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vb = vp[((r & 3) + 8 * (r >> 2)) * 32];
                O = __builtin_amdgcn_mfma_f32_32x32x2f32(S[jb][r] * inv_l, vb, O, 0, 0, 0);
            }
        }
    }

    // ---- message rows: lane holds column (dim) l31 of queries mfma32_row(r, hi) ----
    float* out = a.msg + ((size_t)b * P + q_off) * 128 + head * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = qw + mfma32_row(r, hi);
        if (q < nq) out[(size_t)q * 128] = O[r];
    }
}

}  // namespace

int launch_attention(int B, int N, int M, int cross, int topk, const float* qkv, float* msg, hipStream_t s) {
    if (B <= 0 || N <= 0 || M <= 0) return MDGAT_OK;
    const int nk_max = N > M ? N : M;
    if (nk_max > MAXBLK * 32) {
        mdgat_set_error("attention: %d keys > %d supported by the register-resident kernel", nk_max, MAXBLK * 32);
        return MDGAT_ERR_UNSUPPORTED;
    }
    if (topk > 0) {
        // torch.topk raises when k exceeds the number of keys of either direction (mdgat.py:202)
        const int nk_min = cross ? (N < M ? N : M) : (N < M ? N : M);
        if (topk > nk_min) {
            mdgat_set_error("dynamic attention: k=%d exceeds the number of keys (%d)", topk, nk_min);
            return MDGAT_ERR_BAD_ARG;
        }
    }
    AttnArgs a{qkv, msg, N, M, cross, topk};
    const int nblk = (nk_max + 31) / 32;
    const size_t lds = (size_t)nblk * 32 * (KROW + 32) * sizeof(float);
    dim3 grid((nk_max + 127) / 128, MDGAT_HEADS, B * 2);
    // k == number of keys on both sides keeps every key: identical to full attention
    const bool dyn = topk > 0 && !(topk == N && topk == M);
    if (dyn) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), lds, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), lds, s, a);
    }
    return mdgat_check_hip(hipGetLastError(), "attention launch");
}
