// EXPERIMENT (round 4): the channel-split layer tail of layer_split.hip shaped for LARGE launches - 64 keypoints per
// 4-wave workgroup, two workgroups per CU (67 KB of LDS, <= 256 registers), so that one workgroup's epilogues and waits run
// beside the other's matrix instructions without any hand-placed interleave.
//
// Against layer.hip (a wave owns 16 keypoints and walks ALL weights through LDS: one 1 KB weight-fragment read per 1.5
// matrix instructions, 9.8 MB of LDS reads per CU and launch) a wave here owns a 32-channel unit of a phase for all 64
// keypoints: its weight fragments come from L2 straight into registers (fragment-order images, 8 steps ahead of their use),
// the activation fragments come from LDS - 8 KB per 24 matrix instructions, a third of the LDS traffic.  The price: every
// 64-keypoint workgroup pulls the layer's 590 KB of weights through its CU's L1 (twice layer.hip's L2 -> CU traffic).
//
// A wave's work is ONE flat sequence of steps (unit, k-step): phase 1 = 16 steps (units w and w + 4 of hid), phase 2 = 8
// (unit w of the new x), phase 3 = 8 (q unit w + v row block w; k unit w + v row block 4 + w) or 4 (final_proj unit w); the
// weights of step s + 8 are requested at step s, across phase boundaries.  The arithmetic is layer.hip's to the bit.
#include <utility>
#include <type_traits>
#include <cstdlib>
#include "common.hpp"
#include "layer_image.hpp"
#include "mma_chain.hpp"

namespace {

constexpr int CS_KPB = 4;                    // blocks of 16 keypoints per workgroup
constexpr int CS_PTS = 16 * CS_KPB;
constexpr int CS_NW = 4;
constexpr int CS_FRAG = 512;                 // halves per fragment plane: 64 lanes x 8
// steps between the request of a step's weight fragments and their use: 6 steps of 24 matrix instructions for the 4-fragment
// steps (96 registers in flight), 3 steps of 36 for the 6-fragment steps of the q|k|v phase (72)
constexpr int CS_A4 = 6, CS_A6 = 3;

typedef f16x8 __attribute__((may_alias)) f16x8_c;
typedef f32x4 __attribute__((may_alias)) f32x4_c;

struct CsLds {
    // one 64 KB region, three lives: [x ; msg] fragments [8 k-steps][4][2][512] -> relu(hid) fragments (same shape) -> new x
    // fragments [4 k-steps][4][2][512]; every change of hands is fenced by a barrier on both sides
    _Float16 F[8][CS_KPB][2][CS_FRAG];
    float bias[768];                         // b1 [256] | b2 [128] | b3 [384]
};

struct CAcc4 { f32x4 pm, px, qm, qx; };
struct CAcc2 { f32x4 m, x; };

// knock-outs (measurement builds: tools/ab_build.sh layer_cs <name> -DCS_KNOCK_MFMA | _FRAG | _WLOAD | _STORE): the kernel
// without its matrix instructions / LDS fragment reads / weight loads beyond the first steps / global stores
__device__ __forceinline__ f32x4 cmma(const f16x8& a, const f16x8& b, const f32x4& c) {
#ifdef CS_KNOCK_MFMA
    asm volatile("" :: "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
#ifdef CS_KNOCK_STORE
#define CS_STORE_OK(a) ((a).R == -12345)
#else
#define CS_STORE_OK(a) true
#endif

template <typename F, int... I>
__device__ __forceinline__ void for_const_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void for_const(F&& f) { for_const_impl(f, std::make_integer_sequence<int, N>{}); }

template <int DO_MLP, int MODE3>
__global__ __launch_bounds__(64 * CS_NW, 2) void layer_cs_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char cs_smem[];
    CsLds& S = *reinterpret_cast<CsLds*>(cs_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int pt0 = blockIdx.x * CS_PTS;
    float* bias1 = S.bias;
    float* bias2 = S.bias + 256;
    float* bias3 = S.bias + 384;
    constexpr int NB3 = MODE3 == 1 ? 384 : 128;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // Two workgroups share a CU and do identical work: started together they stay in lockstep - both in their matrix
    // instructions, then both in their epilogues.  The second resident of a CU (workgroups are dispatched breadth first: the
    // first `stagger_mod` of them one per CU) starts late by `stagger` x ~3.9 us (s_sleep 127), once per launch.
    if (a.stagger > 0 && ((blockIdx.x / a.stagger_mod) & 1)) {
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    constexpr int S1 = DO_MLP ? 16 : 0, S2 = DO_MLP ? 8 : 0, S3 = MODE3 == 1 ? 8 : 4, NS = S1 + S2 + S3;
    constexpr int NF = MODE3 == 1 ? 6 : 4, S12 = S1 + S2, A3 = MODE3 == 1 ? CS_A6 : CS_A4;
    f16x8 W[NS][NF];
    auto wfrag = [&](const _Float16* img, int NK, int rb, int ks, int plane) __attribute__((always_inline)) {
        return *reinterpret_cast<const f16x8*>(img + ((size_t)(rb * NK + ks) * 2 + plane) * CS_FRAG + lane * 8);
    };
    auto wunit = [&](f16x8 (&w)[NF], const _Float16* img, int NK, int unit, int ks) __attribute__((always_inline)) {
        w[0] = wfrag(img, NK, 2 * unit, ks, 0);
        w[1] = wfrag(img, NK, 2 * unit, ks, 1);
        w[2] = wfrag(img, NK, 2 * unit + 1, ks, 0);
        w[3] = wfrag(img, NK, 2 * unit + 1, ks, 1);
    };
    // request the weight fragments of step s (nothing beyond the last step)
    auto wload = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
#ifdef CS_KNOCK_WLOAD
        if constexpr (s < NS) {         // zero weights, no loads (the outputs are the biases: finite)
#pragma unroll
            for (int i = 0; i < NF; ++i) { f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; asm volatile("" : "+v"(z)); W[s][i] = z; }
            return;
        }
#endif
        if constexpr (s < S1) wunit(W[s], a.w1f, 8, s < 8 ? wave : wave + 4, s & 7);
        else if constexpr (s < S1 + S2) wunit(W[s], a.w2f, 8, wave, s - S1);
        else if constexpr (s < NS) {
            constexpr int t = s - S1 - S2;
            if constexpr (MODE3 == 1) {
                constexpr int pass = t >> 2, ks = t & 3;
                wunit(W[s], a.w3f, 4, 4 * pass + wave, ks);
                W[s][NF - 2] = wfrag(a.w3f, 4, 16 + 4 * pass + wave, ks, 0);
                W[s][NF - 1] = wfrag(a.w3f, 4, 16 + 4 * pass + wave, ks, 1);
            } else wunit(W[s], a.w3f, 4, wave, t);
        }
    };
    // the requests that belong to step s: step s + CS_A4 of phases 1 / 2, step s + A3 of phase 3
    auto prefetch = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + CS_A4 < S12) wload(std::integral_constant<int, s + CS_A4>{});
        if constexpr (s + A3 >= S12 && s + A3 < NS) wload(std::integral_constant<int, s + A3>{});
        __builtin_amdgcn_sched_barrier(0);
    };
#ifdef CS_KNOCK_FRAG
    auto frag_ld = [&](const _Float16* p) __attribute__((always_inline)) { f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}; asm volatile("" : "+v"(v)); return v; };
#else
    auto frag_ld = [&](const _Float16* p) __attribute__((always_inline)) { return *reinterpret_cast<const f16x8_c*>(p + lane * 8); };
#endif
    auto frag_st = [&](_Float16* p, const f16x8& v) __attribute__((always_inline)) { *reinterpret_cast<f16x8_c*>(p + lane * 8) = v; };

    // ---- the first steps' weights, then this wave's share of the input rows: lane (keypoint l15, g) of wave w
    //      takes channels 32 w + 8 g .. + 7 of x (k-step w) and of msg (k-step 4 + w) for the four keypoint blocks ----
    for_const<NS>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        if constexpr ((t < S12 ? t - CS_A4 : t - A3) < 0) wload(tc);
    });
    f32x4 rx[CS_KPB][2], rm[CS_KPB][2];
#pragma unroll
    for (int kpb = 0; kpb < CS_KPB; ++kpb) {
        const size_t off = (size_t)min(pt0 + kpb * 16 + l15, a.R - 1) * 128 + 32 * wave + 8 * g;
        rx[kpb][0] = *reinterpret_cast<const f32x4*>(a.x + off);
        rx[kpb][1] = *reinterpret_cast<const f32x4*>(a.x + off + 4);
        if (DO_MLP) {
            rm[kpb][0] = *reinterpret_cast<const f32x4*>(a.msg + off);
            rm[kpb][1] = *reinterpret_cast<const f32x4*>(a.msg + off + 4);
        }
    }
    if (DO_MLP) {
        for (int i = tid; i < 256; i += 64 * CS_NW) bias1[i] = a.b1[i];
        for (int i = tid; i < 128; i += 64 * CS_NW) bias2[i] = a.b2[i];
    }
    for (int i = tid; i < NB3; i += 64 * CS_NW) bias3[i] = a.b3[i];
    __builtin_amdgcn_sched_barrier(0);
    {
        unsigned gm = 0;
#pragma unroll
        for (int kpb = 0; kpb < CS_KPB; ++kpb) {
            float v[8];
            f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = rx[kpb][j >> 2][j & 3]; gm = max(gm, __builtin_bit_cast(unsigned, v[j]) & 0x7fffffffu); }
            split8s(v, h, l);
            // DO_MLP: [x ; msg] fragments of phase 1; else x IS the phase-3 operand (k-step w of the new-x fragments)
            frag_st(S.F[wave][kpb][0], h);
            frag_st(S.F[wave][kpb][1], l);
            if (DO_MLP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = rm[kpb][j >> 2][j & 3]; gm = max(gm, __builtin_bit_cast(unsigned, v[j]) & 0x7fffffffu); }
                split8s(v, h, l);
                frag_st(S.F[4 + wave][kpb][0], h);
                frag_st(S.F[4 + wave][kpb][1], l);
            }
        }
        // f16 operand range guard (layer.hip rows_guard)
        if (a.guard && gm >= __builtin_bit_cast(unsigned, MDGAT_F16_GUARD)) __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();

    // one step of a 32-channel unit against the four keypoint blocks: 24 matrix instructions; per accumulator hi.lo, (hi.hi),
    // lo.hi in this order (layer.hip block_mma16 / unit_mma16)
    auto unit_step = [&](const f16x8 (&w)[NF], const _Float16 (*F)[2][CS_FRAG], CAcc4 (&c)[CS_KPB]) __attribute__((always_inline)) {
#pragma unroll
        for (int hf = 0; hf < CS_KPB; hf += 2) {      // two keypoint blocks at a time: 16 registers of activation fragments
            f16x8 xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { xh[i] = frag_ld(F[hf + i][0]); xl[i] = frag_ld(F[hf + i][1]); }
#pragma unroll
            for (int i = 0; i < 2; ++i) { c[hf + i].px = cmma(w[0], xl[i], c[hf + i].px); c[hf + i].qx = cmma(w[2], xl[i], c[hf + i].qx); }
#pragma unroll
            for (int i = 0; i < 2; ++i) { c[hf + i].pm = cmma(w[0], xh[i], c[hf + i].pm); c[hf + i].qm = cmma(w[2], xh[i], c[hf + i].qm); }
#pragma unroll
            for (int i = 0; i < 2; ++i) { c[hf + i].px = cmma(w[1], xh[i], c[hf + i].px); c[hf + i].qx = cmma(w[3], xh[i], c[hf + i].qx); }
        }
    };
    auto zero_acc = [&](CAcc4 (&c)[CS_KPB]) __attribute__((always_inline)) {
#pragma unroll
        for (int kpb = 0; kpb < CS_KPB; ++kpb) { c[kpb].pm = zero4; c[kpb].px = zero4; c[kpb].qm = zero4; c[kpb].qx = zero4; }
    };
    auto combine8 = [&](const CAcc4& c, float (&pv)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] = fmaf(c.px[r], MDGAT_SPLIT_INV, c.pm[r]);
            pv[4 + r] = fmaf(c.qx[r], MDGAT_SPLIT_INV, c.qm[r]);
        }
    };

    if constexpr (DO_MLP != 0) {
        // ---- phase 1: units w (steps 0-7) and w + 4 (steps 8-15) of hid = relu(W1 [x ; msg] + b1) ----
        f16x8 hidh[2][CS_KPB], hidl[2][CS_KPB];       // the split fragments wait in registers until every wave has read its inputs
        for_const<2>([&](auto uc) __attribute__((always_inline)) {
            constexpr int ui = decltype(uc)::value;
            CAcc4 c[CS_KPB];
            zero_acc(c);
            for_const<8>([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value, s = 8 * ui + ks;
                prefetch(std::integral_constant<int, s>{});
                unit_step(W[s], S.F[ks], c);
            });
            const int unit = ui ? wave + 4 : wave;
            float pb[8];
            load8(bias1 + unit * 32 + 8 * g, pb);
#pragma unroll
            for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                float pv[8];
                combine8(c[kpb], pv);
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = fmaxf(pv[j] + pb[j], 0.f);
                split8s(pv, hidh[ui][kpb], hidl[ui][kpb]);
            }
        });
        __syncthreads();                              // every wave is done with the [x ; msg] fragments
#pragma unroll
        for (int ui = 0; ui < 2; ++ui)
#pragma unroll
            for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                frag_st(S.F[ui ? wave + 4 : wave][kpb][0], hidh[ui][kpb]);     // k-step `unit` of phase 2
                frag_st(S.F[ui ? wave + 4 : wave][kpb][1], hidl[ui][kpb]);
            }
        __syncthreads();

        // ---- phase 2: unit w of x += W2 hid + b2 (steps 16-23); the residual is this lane's own slice of the x rows ----
        CAcc4 c2[CS_KPB];
        f32x4 res[CS_KPB][2];
        zero_acc(c2);
        for_const<8>([&](auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(kc)::value, s = S1 + ks;
            prefetch(std::integral_constant<int, s>{});
            if constexpr (ks == 5) {
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                    const size_t off = (size_t)min(pt0 + kpb * 16 + l15, a.R - 1) * 128 + 32 * wave + 8 * g;
                    res[kpb][0] = *reinterpret_cast<const f32x4*>(a.x + off);
                    res[kpb][1] = *reinterpret_cast<const f32x4*>(a.x + off + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            unit_step(W[s], S.F[ks], c2);
        });
        f16x8 xnh[CS_KPB], xnl[CS_KPB];
        {
            float pb[8];
            load8(bias2 + wave * 32 + 8 * g, pb);
#pragma unroll
            for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                const int gp = pt0 + kpb * 16 + l15;
                float pv[8];
                combine8(c2[kpb], pv);
#pragma unroll
                for (int j = 0; j < 8; ++j) { pv[j] += pb[j]; pv[j] += res[kpb][j >> 2][j & 3]; }
                if (gp < a.R && CS_STORE_OK(a)) {
                    float* xr = a.x + (size_t)gp * 128 + 32 * wave + 8 * g;
                    *reinterpret_cast<f32x4*>(xr) = f32x4{pv[0], pv[1], pv[2], pv[3]};
                    *reinterpret_cast<f32x4*>(xr + 4) = f32x4{pv[4], pv[5], pv[6], pv[7]};
                }
                split8s(pv, xnh[kpb], xnl[kpb]);
            }
        }
        __syncthreads();                              // every wave is done with the hid fragments
#pragma unroll
        for (int kpb = 0; kpb < CS_KPB; ++kpb) {
            frag_st(S.F[wave][kpb][0], xnh[kpb]);     // k-step w of phase 3
            frag_st(S.F[wave][kpb][1], xnl[kpb]);
        }
        __syncthreads();
    }

    // ---- phase 3 ----
    if constexpr (MODE3 == 1) {
        for_const<2>([&](auto pc) __attribute__((always_inline)) {
            // pass 0: q unit w (head w) + v row block w; pass 1: k unit w + v row block 4 + w (head of v: j >> 1, dims 16 (j & 1) ..)
            constexpr int pass = decltype(pc)::value;
            CAcc4 c[CS_KPB];
            CAcc2 cv[CS_KPB];
            zero_acc(c);
#pragma unroll
            for (int kpb = 0; kpb < CS_KPB; ++kpb) { cv[kpb].m = zero4; cv[kpb].x = zero4; }
            for_const<4>([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value, s = S1 + S2 + 4 * pass + ks;
                prefetch(std::integral_constant<int, s>{});
                f16x8 xh[CS_KPB], xl[CS_KPB];
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) { xh[kpb] = frag_ld(S.F[ks][kpb][0]); xl[kpb] = frag_ld(S.F[ks][kpb][1]); }
                const f16x8 (&w)[NF] = W[s];
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                    c[kpb].px = cmma(w[0], xl[kpb], c[kpb].px);
                    c[kpb].qx = cmma(w[2], xl[kpb], c[kpb].qx);
                    cv[kpb].x = cmma(xl[kpb], w[NF - 2], cv[kpb].x);
                }
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                    c[kpb].pm = cmma(w[0], xh[kpb], c[kpb].pm);
                    c[kpb].qm = cmma(w[2], xh[kpb], c[kpb].qm);
                    cv[kpb].m = cmma(xh[kpb], w[NF - 2], cv[kpb].m);
                }
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                    c[kpb].px = cmma(w[1], xh[kpb], c[kpb].px);
                    c[kpb].qx = cmma(w[3], xh[kpb], c[kpb].qx);
                    cv[kpb].x = cmma(xh[kpb], w[NF - 1], cv[kpb].x);
                }
            });
            // q / k: [pt][head][plane][32 dims]; q pre-scaled by log2(e) / sqrt(32); residual plane unscaled (common.hpp)
            {
                float pb[8];
                load8(bias3 + (4 * pass + wave) * 32 + 8 * g, pb);
                _Float16* dst0 = (pass == 0 ? a.q16 : a.k16) + (size_t)wave * 64 + 8 * g;
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                    float pv[8];
                    combine8(c[kpb], pv);
                    f16x8 h, l;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // (no contraction: with `pass` a constant the compiler would fuse the scale into the residual,
                        // fma(pv, c, -h) - the residual of the UNROUNDED product; layer.hip rounds the scaled value first)
#pragma clang fp contract(off)
                        pv[j] += pb[j];
                        if (pass == 0) pv[j] *= MDGAT_LOG2E * 0.17677669529663687f;
                        h[j] = (_Float16)pv[j];
                        l[j] = (_Float16)(pv[j] - (float)h[j]);
                    }
                    const int gp = pt0 + kpb * 16 + l15;
                    if (gp < a.R && CS_STORE_OK(a)) {
                        *reinterpret_cast<f16x8*>(dst0 + (size_t)gp * 256) = h;
                        *reinterpret_cast<f16x8*>(dst0 + (size_t)gp * 256 + 32) = l;
                    }
                }
            }
            // v: lane (dim l15 of the half, g) holds keypoints 4 g .. 4 g + 3 of a block: V^T rows [pair][head][plane][dim][PP]
            {
                const int j = 4 * pass + wave;
                const int head = j >> 1, dim = 16 * (j & 1) + l15;
                const float bias = bias3[(8 + head) * 32 + dim];
                const int P = a.N + a.M;
                const bool fast = ((a.N | a.M) & 3) == 0;      // 4 consecutive keypoints share frame and pair, 8-byte aligned
#pragma unroll
                for (int kpb = 0; kpb < CS_KPB; ++kpb) {
                    const int p0 = pt0 + kpb * 16 + 4 * g;
                    if (p0 >= a.R || !CS_STORE_OK(a)) continue;
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) mdgat_split_unscaled(fmaf(cv[kpb].x[r], MDGAT_SPLIT_INV, cv[kpb].m[r]) + bias, h[r], l[r]);
                    if (fast) {
                        const int bb = p0 / P, pp = p0 - bb * P;
                        _Float16* row_h = a.vt16 + (((size_t)bb * 4 + head) * 2 * 32 + dim) * a.PP;
                        const int col = pp < a.N ? pp : a.Npad + pp - a.N;
                        *reinterpret_cast<f16x4*>(row_h + col) = f16x4{h[0], h[1], h[2], h[3]};
                        *reinterpret_cast<f16x4*>(row_h + (size_t)32 * a.PP + col) = f16x4{l[0], l[1], l[2], l[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int pj = p0 + r;
                            if (pj >= a.R) break;
                            const int bj = pj / P, qj = pj - bj * P;
                            const int col = qj < a.N ? qj : a.Npad + qj - a.N;
                            _Float16* rh = a.vt16 + (((size_t)bj * 4 + head) * 2 * 32 + dim) * a.PP;
                            rh[col] = h[r];
                            rh[(size_t)32 * a.PP + col] = l[r];
                        }
                    }
                }
            }
        });
    } else {
        // final_proj: unit w of mdesc = Wf x + bf
        CAcc4 c[CS_KPB];
        zero_acc(c);
        for_const<4>([&](auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(kc)::value, s = S1 + S2 + ks;
            prefetch(std::integral_constant<int, s>{});
            unit_step(W[s], S.F[ks], c);
        });
        float pb[8];
        load8(bias3 + wave * 32 + 8 * g, pb);
#pragma unroll
        for (int kpb = 0; kpb < CS_KPB; ++kpb) {
            const int gp = pt0 + kpb * 16 + l15;
            float pv[8];
            combine8(c[kpb], pv);
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] += pb[j];
            if (gp < a.R && CS_STORE_OK(a)) {
                float* dst = a.mdesc + (size_t)gp * 128 + 32 * wave + 8 * g;
                *reinterpret_cast<f32x4*>(dst) = f32x4{pv[0], pv[1], pv[2], pv[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{pv[4], pv[5], pv[6], pv[7]};
            }
        }
    }
}

template <int DO_MLP, int MODE3>
int launch_cs_t(const LayerArgs& a, hipStream_t s) {
    static std::atomic<unsigned long long> optin;        // (one per template instance)
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(layer_cs_kernel<DO_MLP, MODE3>), sizeof(CsLds), optin, "layer (cs) LDS attribute")) return rc;
    hipLaunchKernelGGL((layer_cs_kernel<DO_MLP, MODE3>), dim3((a.R + CS_PTS - 1) / CS_PTS), dim3(64 * CS_NW), sizeof(CsLds), s, a);
    return mdgat_check_hip(hipGetLastError(), "layer (cs) launch");
}

}  // namespace

int launch_layer_cs(const LayerArgs& a0, int do_mlp, int mode3, hipStream_t s) {
    LayerArgs a = a0;
    static const int stagger = [] { const char* e = getenv("MDGAT_CS_STAGGER"); return e ? atoi(e) : 0; }();
    static const int stagger_mod = [] { const char* e = getenv("MDGAT_CS_STAGGER_MOD"); return e ? atoi(e) : 256; }();
    a.stagger = stagger; a.stagger_mod = stagger_mod;
    if (do_mlp) return mode3 != 1 ? launch_cs_t<1, 2>(a, s) : launch_cs_t<1, 1>(a, s);
    return mode3 != 1 ? launch_cs_t<0, 2>(a, s) : launch_cs_t<0, 1>(a, s);
}
