// The layer tail of layer.hip - [mlp.0 + BN + ReLU -> mlp.3 + residual] of one layer and the q|k|v projection of the next
// (mdgat.py:227-232, 237, 246-248, 274; final_proj of mdgat.py:397 after the last layer) - for launches of a FEW tiles:
// one pair as test.py:132 (batch_size = 1) sends it, small batches.
//
// Why a second kernel.  In layer.hip a wave owns 16 keypoints and ALL output channels: its activations never leave the
// register file, and it walks the whole weight set - a dependent chain of 864 matrix instructions per wave whatever the
// batch (19 us at one wave per SIMD).  A launch of one pair is 8 such workgroups on a 256-CU part: 23.5 us per launch,
// 20 of them in a forward of 0.41 ms.  Here the OUTPUT CHANNELS of a tile are split over the eight waves of a workgroup
// instead: a tile is 32 keypoints, every wave computes a 16- or 32-channel slice of every phase for all of them
// (216 matrix instructions per wave), the activations cross the waves as split-f16 B fragments through LDS (one barrier per
// phase), and a wave's slice of the weights goes from L2 straight into its registers (each fragment is used by one wave
// only: nothing to share through LDS).  Four times the workgroups, a quarter of the chain.
//
// The arithmetic is layer.hip's to the bit: the same split-f16 fragments (common.hpp), the same order of the products
// within every accumulator (hi.lo before lo.hi, k-steps ascending) and the same epilogue operations, so the two kernels are
// interchangeable (tests/test_gpu_ops.py::test_layer_split_equals_layer_kernel compares them bit for bit).
// Not for large launches: every 32-keypoint workgroup pulls the layer's 590 KB of weights through its CU's L1.
#include <utility>
#include <cstdlib>
#include "common.hpp"
#include "layer_image.hpp"
#include "mma_chain.hpp"

namespace {

constexpr int SP_KPB = 2;                    // blocks of 16 keypoints per workgroup
constexpr int SP_PTS = 16 * SP_KPB;          // keypoints per workgroup
constexpr int SP_NW = 8;                     // waves per workgroup
constexpr int SP_TROW = 132;                 // floats per row of the fp32 tiles (128 channels + 16 B pad)
constexpr int SP_FRAG = 512;                 // halves per fragment plane: 64 lanes x 8

typedef f16x8 __attribute__((may_alias)) f16x8_m;
typedef f16x4 __attribute__((may_alias)) f16x4_m;
typedef f32x4 __attribute__((may_alias)) f32x4_m;

// B fragments are stored as the lanes hold them: [k-step][keypoint block][plane hi / lo][lane] x 16 B, read back with
// lane-consecutive ds_read_b128 (conflict free)
struct SpLds {
    float xt[SP_PTS][SP_TROW];               // x rows (fp32): residual of phase 2, new x written in place
    float mt[SP_PTS][SP_TROW];               // msg rows
    _Float16 inF[8][SP_KPB][2][SP_FRAG];     // [x ; msg] fragments: k-steps 0-3 = x, 4-7 = msg
    _Float16 hidF[8][SP_KPB][2][SP_FRAG];    // relu(hid) fragments: k-step = unit of phase 1
    _Float16 xnF[4][SP_KPB][2][SP_FRAG];     // new x fragments
    float bias[768];                         // b1 [256] | b2 [128] | b3 [384]
};

#ifdef SPLIT_TRACE
// s_memtime stamps of waves 0 and 7 of workgroup 0 (tools/ab_build.sh layer_split sptrace -DSPLIT_TRACE; mdgat_split_trace_read)
__device__ long long g_split_trace[2 * 16];
#define SPT(k) do { if (DO_MLP && MODE3 == 1 && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) g_split_trace[(wave == 7) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPT(k) do {} while (0)
#endif

struct Acc2 { f32x4 m, x; };                 // one row block: hi.hi and hi.lo + lo.hi (x 1/2048 when combined)
struct Acc4 { f32x4 pm, px, qm, qx; };       // a 32-channel unit: row blocks P and Q

__device__ __forceinline__ f32x4 mma(const f16x8& a, const f16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int DO_MLP, int MODE3>
__global__ __launch_bounds__(64 * SP_NW) void layer_split_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    SpLds& S = *reinterpret_cast<SpLds*>(sp_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int pt0 = blockIdx.x * SP_PTS;
    float* bias1 = S.bias;
    float* bias2 = S.bias + 256;
    float* bias3 = S.bias + 384;
    constexpr int NB3 = MODE3 == 1 ? 384 : 128;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    SPT(0);
    // fragment (row block rb, k-step ks, plane) of a weight matrix with NK k-steps, as lane (row l15, 16-byte column g) holds it:
    // the fragment-order images (launch_frag_image) keep it as 64 consecutive 16-byte pieces - one contiguous KB per load
    auto wfrag = [&](const _Float16* img, int NK, int rb, int ks, int plane) __attribute__((always_inline)) {
        return *reinterpret_cast<const f16x8*>(img + ((size_t)(rb * NK + ks) * 2 + plane) * SP_FRAG + lane * 8);
    };
    auto frag_ld = [&](const _Float16* p) __attribute__((always_inline)) { return *reinterpret_cast<const f16x8_m*>(p + lane * 8); };
    auto frag_st = [&](_Float16* p, const f16x8& v) __attribute__((always_inline)) { *reinterpret_cast<f16x8_m*>(p + lane * 8) = v; };

    // ---- biases first: vector memory operations return in order, so a bias requested BEHIND the rows and the 32 KB of W1 would
    //      make its LDS store - and with it the first barrier - wait for all of W1 (phase trace: 5 900 of 25 000 ticks) ----
    {
        float bv1 = 0.f, bv2 = 0.f, bv3 = 0.f;
        if (DO_MLP) {
            if (tid < 256) bv1 = a.b1[tid];
            if (tid < 128) bv2 = a.b2[tid];
        }
        if (tid < NB3) bv3 = a.b3[tid];
        __builtin_amdgcn_sched_barrier(0);
        if (DO_MLP) {
            if (tid < 256) bias1[tid] = bv1;
            if (tid < 128) bias2[tid] = bv2;
        }
        if (tid < NB3) bias3[tid] = bv3;
    }
    // ---- input rows: thread t takes row t >> 4, 16-byte pieces (t & 15) and (t & 15) + 16 of it (whole 256-byte half
    //      rows per 16 threads); rows past the end are copies of the last keypoint and are never written back ----
    const int irow = tid >> 4, ic = tid & 15;
    const size_t igp = (size_t)min(pt0 + irow, a.R - 1) * 128;
    const f32x4 tx0 = *reinterpret_cast<const f32x4*>(a.x + igp + ic * 4);
    const f32x4 tx1 = *reinterpret_cast<const f32x4*>(a.x + igp + 64 + ic * 4);
    f32x4 tm0 = zero4, tm1 = zero4;
    if (DO_MLP) {
        tm0 = *reinterpret_cast<const f32x4*>(a.msg + igp + ic * 4);
        tm1 = *reinterpret_cast<const f32x4*>(a.msg + igp + 64 + ic * 4);
    }
    // ---- this wave's slice of W1 (unit `wave`: row blocks P = 2 wave, Q = 2 wave + 1; all 8 k-steps) is requested behind
    //      the rows: it lands while the rows are split ----
    f16x8 w1[8][4];
    if (DO_MLP) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            w1[ks][0] = wfrag(a.w1f, 8, 2 * wave, ks, 0);
            w1[ks][1] = wfrag(a.w1f, 8, 2 * wave, ks, 1);
            w1[ks][2] = wfrag(a.w1f, 8, 2 * wave + 1, ks, 0);
            w1[ks][3] = wfrag(a.w1f, 8, 2 * wave + 1, ks, 1);
        }
    }
    SPT(1);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4_m*>(&S.xt[irow][ic * 4]) = tx0;
    *reinterpret_cast<f32x4_m*>(&S.xt[irow][64 + ic * 4]) = tx1;
    if (DO_MLP) {
        *reinterpret_cast<f32x4_m*>(&S.mt[irow][ic * 4]) = tm0;
        *reinterpret_cast<f32x4_m*>(&S.mt[irow][64 + ic * 4]) = tm1;
    }
    // f16 operand range guard (layer.hip rows_guard): the largest integer image of the rows
    if (a.guard) {
        unsigned gm = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gm = max(gm, max(__builtin_bit_cast(unsigned, tx0[c]) & 0x7fffffffu, __builtin_bit_cast(unsigned, tx1[c]) & 0x7fffffffu));
            gm = max(gm, max(__builtin_bit_cast(unsigned, tm0[c]) & 0x7fffffffu, __builtin_bit_cast(unsigned, tm1[c]) & 0x7fffffffu));
        }
        if (gm >= __builtin_bit_cast(unsigned, MDGAT_F16_GUARD)) __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    SPT(2);
    __syncthreads();
    SPT(3);
    // ---- rows -> split fragments: lane (keypoint l15, g) holds channels 32 ks + 8 g .. + 7 ----
    if (DO_MLP) {
        const float (*src)[SP_TROW] = wave < 4 ? S.xt : S.mt;
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) {
            float v[8];
            f16x8 h, l;
            load8(&src[kpb * 16 + l15][32 * (wave & 3) + 8 * g], v);
            split8s(v, h, l);
            frag_st(S.inF[wave][kpb][0], h);
            frag_st(S.inF[wave][kpb][1], l);
        }
    } else {
        float v[8];
        f16x8 h, l;
        load8(&S.xt[(wave & 1) * 16 + l15][32 * (wave >> 1) + 8 * g], v);
        split8s(v, h, l);
        frag_st(S.xnF[wave >> 1][wave & 1][0], h);
        frag_st(S.xnF[wave >> 1][wave & 1][1], l);
    }
    __syncthreads();
    SPT(4);

    // ---- this wave's slices of the phase-3 weights: q / k unit `wave` (row blocks 2 wave, 2 wave + 1) and row block 16 + wave
    //      of v (mode 1), or row block `wave` of final_proj (mode 2) ----
    f16x8 w3[4][MODE3 == 1 ? 6 : 2];
    auto load_w3 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (MODE3 == 1) {
                w3[ks][0] = wfrag(a.w3f, 4, 2 * wave, ks, 0);
                w3[ks][1] = wfrag(a.w3f, 4, 2 * wave, ks, 1);
                w3[ks][2] = wfrag(a.w3f, 4, 2 * wave + 1, ks, 0);
                w3[ks][3] = wfrag(a.w3f, 4, 2 * wave + 1, ks, 1);
                w3[ks][4] = wfrag(a.w3f, 4, 16 + wave, ks, 0);
                w3[ks][5] = wfrag(a.w3f, 4, 16 + wave, ks, 1);
            } else {
                w3[ks][0] = wfrag(a.w3f, 4, wave, ks, 0);
                w3[ks][1] = wfrag(a.w3f, 4, wave, ks, 1);
            }
        }
    };

    if (DO_MLP) {
        // ---- phase 1: unit `wave` of hid = relu(W1 [x ; msg] + b1) for both keypoint blocks ----
        Acc4 c1[SP_KPB];
        f16x8 w2[8][2];       // the slice of W2 (row block `wave`), requested during the second half of phase 1 (the registers of
                              // the W1 fragments already consumed)
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) { c1[kpb].pm = zero4; c1[kpb].px = zero4; c1[kpb].qm = zero4; c1[kpb].qx = zero4; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks >= 4) {
#pragma unroll
                for (int j = 2 * (ks - 4); j < 2 * (ks - 4) + 2; ++j) {
                    w2[j][0] = wfrag(a.w2f, 8, wave, j, 0);
                    w2[j][1] = wfrag(a.w2f, 8, wave, j, 1);
                }
            }
            f16x8 xh[SP_KPB], xl[SP_KPB];
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { xh[kpb] = frag_ld(S.inF[ks][kpb][0]); xl[kpb] = frag_ld(S.inF[ks][kpb][1]); }
            // per accumulator: hi.lo, (hi.hi), lo.hi in this order - layer.hip block_mma16
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { c1[kpb].px = mma(w1[ks][0], xl[kpb], c1[kpb].px); c1[kpb].qx = mma(w1[ks][2], xl[kpb], c1[kpb].qx); }
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { c1[kpb].pm = mma(w1[ks][0], xh[kpb], c1[kpb].pm); c1[kpb].qm = mma(w1[ks][2], xh[kpb], c1[kpb].qm); }
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { c1[kpb].px = mma(w1[ks][1], xh[kpb], c1[kpb].px); c1[kpb].qx = mma(w1[ks][3], xh[kpb], c1[kpb].qx); }
        }
        SPT(5);
        // the slice of W3 is requested here: it lands behind the epilogue, the barrier and phase 2
        load_w3();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) {
            float pb[8], pv[8];
            load8(bias1 + wave * 32 + 8 * g, pb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = fmaf(c1[kpb].px[r], MDGAT_SPLIT_INV, c1[kpb].pm[r]);
                pv[4 + r] = fmaf(c1[kpb].qx[r], MDGAT_SPLIT_INV, c1[kpb].qm[r]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = fmaxf(pv[j] + pb[j], 0.f);
            f16x8 h, l;
            split8s(pv, h, l);
            frag_st(S.hidF[wave][kpb][0], h);       // lane (n, g): channels 8 g .. 8 g + 7 of unit `wave` = k-step `wave` of phase 2
            frag_st(S.hidF[wave][kpb][1], l);
        }
        SPT(6);
        __syncthreads();
        SPT(7);

        // ---- phase 2: row block `wave` (unit ob = wave >> 1, half P / Q = wave & 1) of x += W2 hid + b2 ----
        Acc2 c2[SP_KPB];
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) { c2[kpb].m = zero4; c2[kpb].x = zero4; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f16x8 hh[SP_KPB], hl[SP_KPB];
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { hh[kpb] = frag_ld(S.hidF[ks][kpb][0]); hl[kpb] = frag_ld(S.hidF[ks][kpb][1]); }
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) c2[kpb].x = mma(w2[ks][0], hl[kpb], c2[kpb].x);
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) c2[kpb].m = mma(w2[ks][0], hh[kpb], c2[kpb].m);
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) c2[kpb].x = mma(w2[ks][1], hh[kpb], c2[kpb].x);
        }
        SPT(8);
        {
            const int ob = wave >> 1, half = wave & 1;
            const int ch = ob * 32 + 8 * g + 4 * half;     // rows 4 g .. 4 g + 3 of block P / Q are channels 8 g (+ 4) .. + 3 of the unit
            const f32x4 pb = *reinterpret_cast<const f32x4_m*>(bias2 + ch);
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) {
                float* xr = &S.xt[kpb * 16 + l15][ch];
                const f32x4 res = *reinterpret_cast<const f32x4_m*>(xr);
                f32x4 pv;
                f16x4 h, l;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = fmaf(c2[kpb].x[r], MDGAT_SPLIT_INV, c2[kpb].m[r]) + pb[r];
                    pv[r] += res[r];
                    h[r] = (_Float16)pv[r];
                    l[r] = (_Float16)((pv[r] - (float)h[r]) * MDGAT_SPLIT_SCALE);
                }
                *reinterpret_cast<f32x4_m*>(xr) = pv;
                *reinterpret_cast<f16x4_m*>(S.xnF[ob][kpb][0] + lane * 8 + 4 * half) = h;
                *reinterpret_cast<f16x4_m*>(S.xnF[ob][kpb][1] + lane * 8 + 4 * half) = l;
            }
        }
        SPT(9);
        __syncthreads();
        SPT(10);
        // new x rows out (the same thread -> row piece map as the input)
        if (pt0 + irow < a.R) {
            *reinterpret_cast<f32x4*>(a.x + igp + ic * 4) = *reinterpret_cast<const f32x4_m*>(&S.xt[irow][ic * 4]);
            *reinterpret_cast<f32x4*>(a.x + igp + 64 + ic * 4) = *reinterpret_cast<const f32x4_m*>(&S.xt[irow][64 + ic * 4]);
        }
    } else {
        load_w3();
    }

    SPT(11);
    // ---- phase 3 ----
    if constexpr (MODE3 == 1) {
        // q / k unit `wave` (head wave & 3; swapped product: lane = keypoint) and row block (head wave >> 1, dims 16 (wave & 1) ..)
        // of v (non-swapped: lane = dim, registers = 4 consecutive keypoints, for the transposed V^T layout)
        Acc4 c3[SP_KPB];
        Acc2 cv[SP_KPB];
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) {
            c3[kpb].pm = zero4; c3[kpb].px = zero4; c3[kpb].qm = zero4; c3[kpb].qx = zero4;
            cv[kpb].m = zero4; cv[kpb].x = zero4;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 xh[SP_KPB], xl[SP_KPB];
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { xh[kpb] = frag_ld(S.xnF[ks][kpb][0]); xl[kpb] = frag_ld(S.xnF[ks][kpb][1]); }
            constexpr int V = 4;
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) {
                c3[kpb].px = mma(w3[ks][0], xl[kpb], c3[kpb].px);
                c3[kpb].qx = mma(w3[ks][2], xl[kpb], c3[kpb].qx);
                cv[kpb].x = mma(xl[kpb], w3[ks][V], cv[kpb].x);
            }
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) {
                c3[kpb].pm = mma(w3[ks][0], xh[kpb], c3[kpb].pm);
                c3[kpb].qm = mma(w3[ks][2], xh[kpb], c3[kpb].qm);
                cv[kpb].m = mma(xh[kpb], w3[ks][V], cv[kpb].m);
            }
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) {
                c3[kpb].px = mma(w3[ks][1], xh[kpb], c3[kpb].px);
                c3[kpb].qx = mma(w3[ks][3], xh[kpb], c3[kpb].qx);
                cv[kpb].x = mma(xh[kpb], w3[ks][V + 1], cv[kpb].x);
            }
        }
        SPT(12);
        // q / k: [pt][head][plane][32 dims]; q pre-scaled by log2(e) / sqrt(32); residual plane unscaled (common.hpp)
        {
            float pb[8];
            load8(bias3 + wave * 32 + 8 * g, pb);
            _Float16* dst0 = (wave < 4 ? a.q16 : a.k16) + (size_t)(wave & 3) * 64 + 8 * g;
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) {
                float pv[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = fmaf(c3[kpb].px[r], MDGAT_SPLIT_INV, c3[kpb].pm[r]);
                    pv[4 + r] = fmaf(c3[kpb].qx[r], MDGAT_SPLIT_INV, c3[kpb].qm[r]);
                }
                f16x8 h, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pv[j] += pb[j];
                    if (wave < 4) pv[j] *= MDGAT_LOG2E * 0.17677669529663687f;
                    h[j] = (_Float16)pv[j];
                    l[j] = (_Float16)(pv[j] - (float)h[j]);
                }
                const int gp = pt0 + kpb * 16 + l15;
                if (gp < a.R) {
                    *reinterpret_cast<f16x8*>(dst0 + (size_t)gp * 256) = h;
                    *reinterpret_cast<f16x8*>(dst0 + (size_t)gp * 256 + 32) = l;
                }
            }
        }
        // v: lane (dim l15 of the half, g) holds keypoints 4 g .. 4 g + 3 of the block: V^T rows [pair][head][plane][dim][PP]
        {
            const int head = wave >> 1, dim = 16 * (wave & 1) + l15;
            const float bias = bias3[(8 + head) * 32 + dim];
            const int P = a.N + a.M;
            const bool fast = ((a.N | a.M) & 3) == 0;      // 4 consecutive keypoints share frame and pair, 8-byte aligned
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) {
                const int p0 = pt0 + kpb * 16 + 4 * g;
                if (p0 >= a.R) continue;
                _Float16 h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) mdgat_split_unscaled(fmaf(cv[kpb].x[j], MDGAT_SPLIT_INV, cv[kpb].m[j]) + bias, h[j], l[j]);
                if (fast) {
                    const int bb = p0 / P, pp = p0 - bb * P;
                    _Float16* row_h = a.vt16 + (((size_t)bb * 4 + head) * 2 * 32 + dim) * a.PP;
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
                        _Float16* rh = a.vt16 + (((size_t)bj * 4 + head) * 2 * 32 + dim) * a.PP;
                        rh[col] = h[j];
                        rh[(size_t)32 * a.PP + col] = l[j];
                    }
                }
            }
        }
        SPT(13);
    } else {
        // final_proj: row block `wave` (unit wave >> 1, half wave & 1) of mdesc = Wf x + bf
        Acc2 c3[SP_KPB];
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) { c3[kpb].m = zero4; c3[kpb].x = zero4; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 xh[SP_KPB], xl[SP_KPB];
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) { xh[kpb] = frag_ld(S.xnF[ks][kpb][0]); xl[kpb] = frag_ld(S.xnF[ks][kpb][1]); }
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) c3[kpb].x = mma(w3[ks][0], xl[kpb], c3[kpb].x);
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) c3[kpb].m = mma(w3[ks][0], xh[kpb], c3[kpb].m);
#pragma unroll
            for (int kpb = 0; kpb < SP_KPB; ++kpb) c3[kpb].x = mma(w3[ks][1], xh[kpb], c3[kpb].x);
        }
        const int ch = (wave >> 1) * 32 + 8 * g + 4 * (wave & 1);
        const f32x4 pb = *reinterpret_cast<const f32x4_m*>(bias3 + ch);
#pragma unroll
        for (int kpb = 0; kpb < SP_KPB; ++kpb) {
            const int gp = pt0 + kpb * 16 + l15;
            f32x4 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = fmaf(c3[kpb].x[r], MDGAT_SPLIT_INV, c3[kpb].m[r]) + pb[r];
            if (gp < a.R) *reinterpret_cast<f32x4*>(a.mdesc + (size_t)gp * 128 + ch) = pv;
        }
    }
}

// row image [rows][rowh] (hi plane | lo plane | pad; layer.hip) -> fragment order: piece (rb, ks, plane, lane) = the 8 halves at
// row 16 rb + (lane & 15), column plane K + 32 ks + 8 (lane >> 4)
__global__ __launch_bounds__(256) void frag_image_kernel(const _Float16* img, _Float16* out, int rows, int K, int rowh) {
    const int NK = K / 32;
    const size_t pieces = (size_t)(rows / 16) * NK * 2 * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (size_t)gridDim.x * 256) {
        const int ln = (int)(i & 63), plane = (int)((i >> 6) & 1);
        const size_t q = i >> 7;
        const int ks = (int)(q % NK), rb = (int)(q / NK);
        *reinterpret_cast<f16x8*>(out + i * 8) =
            *reinterpret_cast<const f16x8*>(img + (size_t)(rb * 16 + (ln & 15)) * rowh + plane * K + 32 * ks + 8 * (ln >> 4));
    }
}

template <int DO_MLP, int MODE3>
int launch_split_t(const LayerArgs& a, hipStream_t s) {
    static std::atomic<unsigned long long> optin;        // (one per template instance)
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(layer_split_kernel<DO_MLP, MODE3>), sizeof(SpLds), optin, "layer (split) LDS attribute")) return rc;
    hipLaunchKernelGGL((layer_split_kernel<DO_MLP, MODE3>), dim3((a.R + SP_PTS - 1) / SP_PTS), dim3(64 * SP_NW), sizeof(SpLds), s, a);
    return mdgat_check_hip(hipGetLastError(), "layer (split) launch");
}

}  // namespace

#ifdef SPLIT_TRACE
extern "C" int mdgat_split_trace_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_split_trace), n * sizeof(long long)); }
#endif

int launch_frag_image(const _Float16* img, _Float16* out, int rows, int K, int rowh, hipStream_t s) {
    const size_t pieces = (size_t)(rows / 16) * (K / 32) * 2 * 64;
    hipLaunchKernelGGL(frag_image_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, img, out, rows, K, rowh);
    return mdgat_check_hip(hipGetLastError(), "frag_image launch");
}

int launch_layer_split(const LayerArgs& a, int do_mlp, int mode3, hipStream_t s) {
    if (do_mlp) return mode3 != 1 ? launch_split_t<1, 2>(a, s) : launch_split_t<1, 1>(a, s);
    return mode3 != 1 ? launch_split_t<0, 2>(a, s) : launch_split_t<0, 1>(a, s);
}
