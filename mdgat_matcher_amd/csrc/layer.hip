// One attentional-propagation layer after the attention itself, fused with the NEXT layer's q/k/v
// projection, for 128 keypoints per workgroup - the activations never leave the register file:
//
//   phase 1  hid   = relu(W1 [x ; msg] + b1)      mlp.0 + folded BN + ReLU, merge folded in (mdgat.py:237, 247-248)
//   phase 2  x    += W2 hid + b2                  mlp.3 and the residual (mdgat.py:248, 274)
//   phase 3  q|k|v = Wqkv x + bqkv                proj[0..2] of the next layer (mdgat.py:227-232), written in
//                                                 the split-f16 operand layouts of the attention kernel
//            (last layer: mdesc = Wf x + bf, final_proj of mdgat.py:397, instead of q|k|v)
//
// Arithmetic: split-f16 products on the f16 matrix cores (common.hpp: x = hi + lo/2048, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulation) - fp32-class accuracy at 16/3 the f32 MFMA rate.
//
// gfx950 mapping.  A wave owns 32 keypoints for the whole chain and computes every product "swapped"
// (D^T = W X^T): in the 32x32 C/D fragment layout a lane then holds, for ITS keypoint (lane & 31), 16
// output channels per 32-channel row block.  The W rows are fed in a permuted order (bits 2 and 3 of
// the row index swapped) so that the 8 registers of a half block are 8 CONSECUTIVE channels - which is
// exactly the B-operand fragment of the next product (k-slots 8 hi .. 8 hi + 7 of a 16-deep k-step).
// So relu(hid) and the new x are split to f16 in place and feed the next GEMM without touching LDS or
// memory; only x (fp32, for the residual) and the q/k/v operands are written.  The weights (pre-split
// at load time, [row][hi plane | lo plane]) stream through LDS one 32-row block at a time (33 KB,
// double buffered, 24 stages per tile), shared by the 4 waves; rows are padded by 16 bytes so that the
// ds_read_b128 fragment reads are bank-conflict free.  V uses the non-swapped product (the same x
// fragments as A operand) so that a lane holds 4 consecutive keypoints of one dim: 8-byte stores into
// the transposed V^T layout.  One wave per SIMD (about 370 registers), the MFMA pipe is the bound.
#include "common.hpp"
#include "mma_chain.hpp"


namespace {

constexpr int ROWH256 = 520;                 // LDS row (halves) for K = 256: 256 hi | 256 lo | 8 pad
constexpr int STAGE_HALVES = 32 * ROWH256;   // one staging buffer (33280 B)

struct LayerArgs {
    float* x;               // [R][128] descriptors, updated in place by phase 2
    const float* msg;       // [R][128] attention output (head-major channels)
    const _Float16* w1s;    // [256][2][256] split
    const float* b1;        // [256]
    const _Float16* w2s;    // [128][2][256]
    const float* b2;        // [128]
    const _Float16* w3s;    // [384][2][128] (q|k|v of the next layer) or [128][2][128] (final_proj)
    const float* b3;        // [384] or [128]
    _Float16* q16;          // outputs of phase 3 (mode 1)
    _Float16* k16;
    _Float16* vt16;
    float* mdesc;           // [R][128] output of phase 3 (mode 2)
    int R, N, M, Npad, PP;
    int do_mlp;             // 0: phase 3 only (first layer / no layers)
    int mode3;              // 1: q|k|v, 2: final projection
};

// ---- weight staging: a row block = 32 rows x 2K halves, contiguous in memory ----
template <int K>
__device__ __forceinline__ void stage_issue(const _Float16* g, f32x4 (&st)[8], int tid) {
    constexpr int NU = (32 * 2 * K * 2 / 16) / 256;   // 16-byte chunks per thread: 8 (K = 256) or 4 (K = 128)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        st[u] = *reinterpret_cast<const f32x4*>(g + (size_t)(tid + 256 * u) * 8);
    }
}
template <int K>
__device__ __forceinline__ void stage_commit(_Float16* buf, const f32x4 (&st)[8], int tid) {
    constexpr int NU = (32 * 2 * K * 2 / 16) / 256;
    constexpr int CPR = 2 * K * 2 / 16;               // chunks per row
    constexpr int ROWH = 2 * K + 8;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = tid + 256 * u;
        *reinterpret_cast<f32x4*>(buf + (c / CPR) * ROWH + (c % CPR) * 8) = st[u];
    }
}

__global__ __launch_bounds__(256, 1) void layer_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];   // 2 x STAGE_HALVES + 768 floats of biases
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wrow = perm32(l31);
    const int pt_raw = blockIdx.x * 128 + wave * 32 + l31;
    const int pt = min(pt_raw, a.R - 1);               // clamped loads; stores are masked
    _Float16* buf0 = smem;
    _Float16* buf1 = smem + STAGE_HALVES;
    float* bias1 = reinterpret_cast<float*>(smem + 2 * STAGE_HALVES);   // [256]
    float* bias2 = bias1 + 256;                                         // [128]
    float* bias3 = bias2 + 128;                                         // [384]
    f32x4 st[8];

    const int n3 = a.mode3 == 1 ? 12 : 4;    // row blocks of phase 3
    {
        if (a.do_mlp) {
            bias1[tid] = a.b1[tid];
            if (tid < 128) bias2[tid] = a.b2[tid];
        }
        for (int i = tid; i < n3 * 32; i += 256) bias3[i] = a.b3[i];
    }

    f16x8 xnh[8], xnl[8];     // the (new) descriptors of this lane's keypoint as 8 k-step fragments

    // phase-3 epilogue of row block qb
    auto epilogue3 = [&](int qb, const f32x16& o) {
        if (a.mode3 == 1 && qb < 8) {
            // q or k of head qb & 3: [pt][head][plane][32 dims]; this lane's dims 16 t + 8 hi .. + 7
            const float sc = qb < 4 ? MDGAT_LOG2E * 0.17677669529663687f : 1.0f;   // log2(e) / sqrt(32) on q
            _Float16* dst = (qb < 4 ? a.q16 : a.k16) + ((size_t)pt * 4 + (qb & 3)) * 64 + 8 * hi;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float bias[8], v[8];
                load8(bias3 + qb * 32 + 16 * t + 8 * hi, bias);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (o[8 * t + j] + bias[j]) * sc;
                f16x8 h, l;
                split8s(v, h, l);
                // lanes past the end hold a copy of the last keypoint (clamped loads): they rewrite the same values
                *reinterpret_cast<f16x8*>(dst + 16 * t) = h;
                *reinterpret_cast<f16x8*>(dst + 32 + 16 * t) = l;
            }
        } else if (a.mode3 == 1) {
            // v of head qb & 3 (non-swapped product): lane = dim l31, registers = keypoints mfma32_row(r, hi) of this wave
            const int head = qb & 3;
            const float bias = bias3[qb * 32 + l31];
            const int P = a.N + a.M;
            const int wave_pt0 = blockIdx.x * 128 + wave * 32;
            const bool fast = ((a.N | a.M) & 3) == 0;          // 4 consecutive keypoints share frame and pair, 8-byte aligned
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int p0 = wave_pt0 + 8 * g + 4 * hi;
                if (p0 >= a.R) continue;
                _Float16 h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) mdgat_split(o[4 * g + j] + bias, h[j], l[j]);
                if (fast) {
                    const int bb = p0 / P, pp = p0 - bb * P;
                    _Float16* row_h = a.vt16 + (((size_t)bb * 4 + head) * 2 * 32 + l31) * a.PP;
                    const int col = pp < a.N ? pp : a.Npad + pp - a.N;
                    *reinterpret_cast<f16x4*>(row_h + col) = f16x4{h[0], h[1], h[2], h[3]};
                    *reinterpret_cast<f16x4*>(row_h + (size_t)32 * a.PP + col) = f16x4{l[0], l[1], l[2], l[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pj = p0 + j;
                        if (pj >= a.R) break;
                        const int bj = pj / P, qj = pj - bj * P;
                        const int col = qj < a.N ? qj : a.Npad + qj - a.N;
                        _Float16* rh = a.vt16 + (((size_t)bj * 4 + head) * 2 * 32 + l31) * a.PP;
                        rh[col] = h[j];
                        rh[(size_t)32 * a.PP + col] = l[j];
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float bias[8], v[8];
                const int ch = qb * 32 + 16 * t + 8 * hi;
                load8(bias3 + ch, bias);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = o[8 * t + j] + bias[j];
                store8(a.mdesc + (size_t)pt * 128 + ch, v);
            }
        }
    };

    if (a.do_mlp) {
        // ---- fragments of [x ; msg]: k-step ks covers channels 16 ks .. 16 ks + 15, this lane 8 hi .. 8 hi + 7 ----
        f16x8 ah[16], al[16];
        stage_issue<256>(a.w1s, st, tid);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float* src = (ks < 8 ? a.x : a.msg) + (size_t)pt * 128 + 16 * (ks & 7) + 8 * hi;
            float v[8];
            load8(src, v);
            split8s(v, ah[ks], al[ks]);
        }
        stage_commit<256>(buf0, st, tid);
        __syncthreads();

        // ---- phase 1: 8 row blocks of W1 -> hidden fragments (k-steps 2 rb, 2 rb + 1 of phase 2) ----
        f16x8 hh[16], hl[16];
        auto epilogue1 = [&](int rb, const f32x16& o) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float bias[8], v[8];
                load8(bias1 + rb * 32 + 16 * t + 8 * hi, bias);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(o[8 * t + j] + bias[j], 0.f);
                split8s(v, hh[2 * rb + t], hl[2 * rb + t]);
            }
        };
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            _Float16* cur = (rb & 1) ? buf1 : buf0;
            _Float16* nxt = (rb & 1) ? buf0 : buf1;
            if (rb < 7) stage_issue<256>(a.w1s + (size_t)(rb + 1) * 32 * 512, st, tid);
            else stage_issue<256>(a.w2s, st, tid);
            f32x16 o;
            block_mma<16, true>(cur, wrow, hi, ah, al, o);
            stage_commit<256>(nxt, st, tid);   // before the epilogue: its vmcnt wait must not cover this stage's stores
            epilogue1(rb, o);
            __syncthreads();
        }

        // ---- phase 2: 4 row blocks of W2, residual, new x (fp32 to memory, split fragments kept) ----
        auto epilogue2 = [&](int ob, const f32x16& o, const float (&res)[16]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float bias[8], v[8];
                const int ch = ob * 32 + 16 * t + 8 * hi;
                load8(bias2 + ch, bias);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = res[8 * t + j] + (o[8 * t + j] + bias[j]);
                store8(a.x + (size_t)pt * 128 + ch, v);
                split8s(v, xnh[2 * ob + t], xnl[2 * ob + t]);
            }
        };
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            _Float16* cur = (ob & 1) ? buf1 : buf0;
            _Float16* nxt = (ob & 1) ? buf0 : buf1;
            if (ob < 3) stage_issue<256>(a.w2s + (size_t)(ob + 1) * 32 * 512, st, tid);
            else stage_issue<128>(a.w3s, st, tid);
            float res[16];   // residual x of this block, loaded ahead of the MFMAs
            {
                float ra[8], rb8[8];
                const int ch = ob * 32;
                load8(a.x + (size_t)pt * 128 + ch + 8 * hi, ra);
                load8(a.x + (size_t)pt * 128 + ch + 16 + 8 * hi, rb8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { res[j] = ra[j]; res[8 + j] = rb8[j]; }
            }
            f32x16 o;
            block_mma<16, true>(cur, wrow, hi, hh, hl, o);
            if (ob < 3) stage_commit<256>(nxt, st, tid);
            else stage_commit<128>(nxt, st, tid);
            epilogue2(ob, o, res);
            __syncthreads();
        }

    } else {
        stage_issue<128>(a.w3s, st, tid);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float v[8];
            load8(a.x + (size_t)pt * 128 + 16 * ks + 8 * hi, v);
            split8s(v, xnh[ks], xnl[ks]);
        }
        stage_commit<128>(buf0, st, tid);
        __syncthreads();
    }

    // ---- phase 3: q | k | v of the next layer (12 row blocks) or the final projection (4); after an even
    //      number of stages the first block of W3 is in buf0 in both branches ----
#pragma unroll
    for (int qb = 0; qb < 12; ++qb) {
        if (qb < n3) {
            _Float16* cur = (qb & 1) ? buf1 : buf0;
            _Float16* nxt = (qb & 1) ? buf0 : buf1;
            const bool more = qb + 1 < n3;
            if (more) stage_issue<128>(a.w3s + (size_t)(qb + 1) * 32 * 256, st, tid);
            f32x16 o;
            if (a.mode3 == 1 && qb >= 8) block_mma<8, false>(cur, l31, hi, xnh, xnl, o);
            else block_mma<8, true>(cur, wrow, hi, xnh, xnl, o);
            if (more) stage_commit<128>(nxt, st, tid);
            epilogue3(qb, o);
            __syncthreads();
        }
    }
}

// fp32 [rows][K] -> split [rows][2][K] (hi plane | lo plane), once per weight load
__global__ __launch_bounds__(256) void split_rows_kernel(const float* w, _Float16* out, int rows, int K) {
    const size_t total = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / K, c = i - r * K;
        _Float16 h, l;
        mdgat_split(w[i], h, l);
        out[r * 2 * K + c] = h;
        out[r * 2 * K + K + c] = l;
    }
}

}  // namespace

int launch_split_rows(const float* w, _Float16* out, int rows, int K, hipStream_t s) {
    const size_t total = (size_t)rows * K;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(split_rows_kernel, dim3(blocks), dim3(256), 0, s, w, out, rows, K);
    return mdgat_check_hip(hipGetLastError(), "split_rows launch");
}

int launch_layer(const LayerLaunch& p, hipStream_t s) {
    if (p.R <= 0) return MDGAT_OK;
    LayerArgs a{};
    a.x = p.x; a.msg = p.msg;
    a.w1s = p.w1s; a.b1 = p.b1; a.w2s = p.w2s; a.b2 = p.b2; a.w3s = p.w3s; a.b3 = p.b3;
    a.q16 = p.out.q16; a.k16 = p.out.k16; a.vt16 = p.out.vt16; a.mdesc = p.mdesc;
    a.R = p.R; a.N = p.N; a.M = p.M; a.Npad = p.out.Npad; a.PP = p.out.PP;
    a.do_mlp = p.do_mlp; a.mode3 = p.mode3;
    const size_t lds = (size_t)2 * STAGE_HALVES * sizeof(_Float16) + 768 * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (int rc = mdgat_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "layer LDS attribute"))
            return rc;
        attr = true;
    }
    hipLaunchKernelGGL(layer_kernel, dim3((p.R + 127) / 128), dim3(256), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "layer launch");
}
