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
// So relu(hid) and the new x are split to f16 in place and feed the next GEMM without touching memory.
//
// Weights: pre-split at load time into the LDS image itself ([row][hi plane | lo plane | 16 B pad], the pad
// makes the ds_read_b128 fragment reads conflict free), so a stage (32 rows of K = 256 or 64 rows of K = 128,
// 33 KB) is a flat copy: global_load_lds_dwordx4 moves it L2 -> LDS without passing through registers, issued at
// the start of the stage that precedes its use, double buffered, one barrier per stage (18 stages per tile).
//
// One wave per SIMD (its 32 keypoints need ~430 registers), so nothing but the wave's own instruction stream
// can overlap the matrix pipe with the rest: the epilogue of block n (accumulator combine, bias, ReLU, f16
// split, output staging and stores) is sliced and issued inside the k-loop of block n + 1 (block_mma_il), two
// accumulator sets alternate.
//
// Global traffic is whole rows only: a wave-private LDS tile transposes between "a half wave / 8 lanes / 4
// lanes per contiguous row" (what the memory system wants) and the fragment order (what the MFMAs want).
// V uses the non-swapped product (a lane holds 4 consecutive keypoints of one dim) for the transposed V^T layout.
#include <utility>
#include "common.hpp"

#ifdef LAYER_TRACE
__device__ long long g_dbg[8192];
extern "C" int mdgat_debug_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), n * sizeof(long long)); }
#define TRACE_OFF (2 * 33 * 1024 + (768 + 4 * 4224) * 4)
__device__ __forceinline__ void trace_point(int slot) {
    extern __shared__ __attribute__((aligned(16))) char tsm[];
    if ((threadIdx.x & 63) == 0) {
        long long* tl = reinterpret_cast<long long*>(tsm + TRACE_OFF) + (threadIdx.x >> 6) * 256;
        const int c = (int)tl[255];
        tl[c] = ((long long)slot << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffLL);
        tl[255] = c + 1;
    }
}
#define TR(slot) trace_point(slot)
#define MMA_TR(ks) if ((ks & 1) == 1) trace_point(50 + ks)
#ifndef TRACE_MLP
#define TRACE_MLP 1
#endif
#else
#define TR(slot)
#endif
#include "mma_chain.hpp"
namespace {

constexpr int ROWH256 = 520;                 // LDS / image row (halves) for K = 256: 256 hi | 256 lo | 8 pad
constexpr int ROWH128 = 264;                 // K = 128
constexpr int STAGE_BYTES = 33 * 1024;       // one staging buffer: 33 DMA chunks (32 x 1040 B rounded up, or 64 x 528 B)
constexpr int STAGE_HALVES = STAGE_BYTES / 2;
constexpr int TROW = 132;                    // floats per row of a wave's activation tile (128 channels + 16 B pad)
constexpr int TILE_FLOATS = 32 * TROW;       // [32 keypoints][TROW]: 16896 B per wave
constexpr int QKROW = 72;                    // halves per row of the q/k store tile (32 hi | 32 lo | 16 B pad)
constexpr int VROW = 40;                     // halves per row of the V^T store tile (32 keypoints | 16 B pad)

// the store tiles are written as halves and read back as 16-byte pieces: accesses that must not be reordered
// on the strength of their types
typedef f16x8 __attribute__((may_alias)) f16x8_a;
typedef f16x4 __attribute__((may_alias)) f16x4_a;
typedef f32x4 __attribute__((may_alias)) f32x4_a;

struct LayerArgs {
    float* x;               // [R][128] descriptors, updated in place by phase 2
    const float* msg;       // [R][128] attention output (head-major channels)
    const _Float16* w1s;    // [256][ROWH256] split image
    const float* b1;        // [256]
    const _Float16* w2s;    // [128][ROWH256]
    const float* b2;        // [128]
    const _Float16* w3s;    // [384][ROWH128] (q|k|v of the next layer) or [128][ROWH128] (final_proj)
    const float* b3;        // [384] or [128]
    _Float16* q16;          // outputs of phase 3 (mode 1)
    _Float16* k16;
    _Float16* vt16;
    float* mdesc;           // [R][128] output of phase 3 (mode 2)
    int R, N, M, Npad, PP;
};

// One 33 KB stage: chunk c (1 KB) is moved by wave c & 3; the LDS address comes from M0, the lanes supply
// consecutive 16-byte pieces.  Inline asm: the compiler must not know that LDS is written (it would order every
// later ds_read behind the copy); completion is awaited explicitly (stage_wait) before the stage barrier.
// The 9 copies of a wave are issued one per k-step inside the k-loop of the running block (stage_dma_slice), so
// that their issue cost (address VALU, M0, the VMEM issue itself) also hides behind matrix instructions.
__device__ __forceinline__ void stage_dma_slice(const _Float16* g, unsigned lds_addr, int wave, int lane, int i) {
    const int c = min(wave + 4 * i, 32);      // (waves 1-3 copy the last chunk once more: no branch)
    // scalar base + per-lane 32-bit offset: the address arithmetic stays on the scalar unit
    const char* src = reinterpret_cast<const char*>(g) + c * 1024;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr + c * 1024), "v"(lane * 16), "s"(src) : "memory");
}
__device__ __forceinline__ void stage_dma(const _Float16* g, unsigned lds_addr, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 9; ++i) stage_dma_slice(g, lds_addr, wave, lane, i);
}
// f(0), f(1), ... f(N - 1) with literal arguments (a `#pragma unroll` loop over large inlined bodies is not reliably
// unrolled, and a rolled loop would index the register arrays of the epilogues dynamically)
template <typename F, int... U>
__device__ __forceinline__ void for_each_unit(F&& f, std::integer_sequence<int, U...>) { (f(U), ...); }
template <int N, typename F>
__device__ __forceinline__ void for_units(F&& f) { for_each_unit(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ void stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// DO_MLP 0: phase 3 only (first layer / no layers).  MODE3 1: q|k|v (12 row blocks), 2: final projection (4).
template <int DO_MLP, int MODE3>
__global__ __launch_bounds__(256, 1) void layer_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];   // 2 stages, 768 floats of biases, 4 tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wrow = perm32(l31);
    // (no pointer tables here: a generic pointer loaded from a constant table is taken for a GLOBAL pointer)
    auto bufp = [&](int i) __attribute__((always_inline)) { return smem + (i & 1) * STAGE_HALVES; };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)smem;
    auto ldsb = [&](int i) __attribute__((always_inline)) { return lds0 + (unsigned)(i & 1) * STAGE_BYTES; };
    float* bias1 = reinterpret_cast<float*>(smem + 2 * STAGE_HALVES);   // [256]
    float* bias2 = bias1 + 256;                                         // [128]
    float* bias3 = bias2 + 128;                                         // [384]
    // Wave-private tile.  Same wave, in-order LDS: no barriers.  It holds x (fp32) through phases 1-2 (residual
    // source, new x written in place), then serves as the staging tile of the phase-3 outputs.
    float* tile = bias3 + 384 + wave * TILE_FLOATS;
    _Float16* tile16 = reinterpret_cast<_Float16*>(tile);
    const int wave_pt0 = blockIdx.x * 128 + wave * 32;
    constexpr int NB3 = MODE3 == 1 ? 12 : 4;      // row blocks of phase 3 (two per stage)

#ifdef LAYER_TRACE
    const bool trace_on = (blockIdx.x == 3 || blockIdx.x == gridDim.x - 2) && (wave == 0 || wave == 3) && DO_MLP == TRACE_MLP && MODE3 == 1;
    const int tbase = ((blockIdx.x != 3) * 2 + (wave == 3)) * 256;
    long long* tlds = reinterpret_cast<long long*>(reinterpret_cast<char*>(smem) + TRACE_OFF) + wave * 256;
    if (lane == 0) tlds[255] = 0;
#endif
    TR(0);
    stage_dma(DO_MLP ? a.w1s : a.w3s, ldsb(0), wave, lane);
    if (DO_MLP) {
        bias1[tid] = a.b1[tid];
        if (tid < 128) bias2[tid] = a.b2[tid];
    }
    for (int i = tid; i < NB3 * 32; i += 256) bias3[i] = a.b3[i];

    // [R][128] fp32 rows of this wave's 32 keypoints <-> tile: half a wave per 512-byte row; all 16 loads of a
    // matrix are in flight together
    auto rows_load = [&](const float* src, f32x4 (&t)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gp = min(wave_pt0 + 2 * i + hi, a.R - 1);
            t[i] = *reinterpret_cast<const f32x4*>(src + (size_t)gp * 128 + l31 * 4);
        }
    };
    auto rows_to_tile = [&](const f32x4 (&t)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4_a*>(tile + (2 * i + hi) * TROW + l31 * 4) = t[i];
    };
    auto tile_to_rows = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = 2 * i + hi;
            const f32x4 t = *reinterpret_cast<const f32x4_a*>(tile + p * TROW + l31 * 4);
            // rows past the end hold copies of the last keypoint (clamped loads): they rewrite the same values
            const int gp = min(wave_pt0 + p, a.R - 1);
            *reinterpret_cast<f32x4*>(dst + (size_t)gp * 128 + l31 * 4) = t;
        }
    };

    SplitAcc acc[2];          // alternate between consecutive blocks
    f32x16 o;                 // combined output of the block whose epilogue is in flight
    f16x8 xnh[8], xnl[8];     // the (new) descriptors of this lane's keypoint as 8 k-step fragments

    // ---- epilogues as sequences of small UNITS.  block_mma_il offers three slots per k-step, one behind each
    //      matrix instruction; a unit placed in a slot runs in the shadow of that instruction (32 cycles = 8 VALU
    //      issues of this wave).  With one wave per SIMD nothing hides the latency of a DEPENDENT VALU chain either,
    //      so a unit applies ONE operation to all 16 values of the block (8 independent packed instructions) rather
    //      than all operations to one value; LDS operands are read a few units before their use. ----
    float pbias[16], pv[16], phf[16];  // bias, values in flight, their f16 heads converted back
    float pbias_v = 0.f;
    f16x8 sth[2], stl[2];             // split halves on their way to the store tile
    f32x4 pback[4];                   // store tile read back
    constexpr int E_UNITS = 22;

    auto combine4 = [&](const SplitAcc& c, int i) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) o[r] = fmaf(c.x[r], MDGAT_SPLIT_INV, c.m[r]);
    };
    auto load_bias16 = [&](const float* b) __attribute__((always_inline)) {   // this lane's 2 x 8 channels of a row block
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float b8[8];
            load8(b + 16 * t + 8 * hi, b8);
#pragma unroll
            for (int j = 0; j < 8; ++j) pbias[8 * t + j] = b8[j];
        }
    };
    // the split pv -> (h, l) in five units of eight independent instructions: s = 0 .. 5
    auto split_unit = [&](int s_, f16x8 (&h)[2], f16x8 (&l)[2]) __attribute__((always_inline)) {
        if (s_ == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) h[j >> 3][j & 7] = (_Float16)pv[j];
        } else if (s_ == 1 || s_ == 2) {
#pragma unroll
            for (int j = 8 * (s_ - 1); j < 8 * s_; ++j) phf[j] = (float)h[j >> 3][j & 7];
        } else if (s_ == 3) {
#pragma unroll
            for (int j = 0; j < 16; ++j) pv[j] -= phf[j];
        } else if (s_ == 4) {
#pragma unroll
            for (int j = 0; j < 16; ++j) pv[j] *= MDGAT_SPLIT_SCALE;
        } else if (s_ == 5) {
#pragma unroll
            for (int j = 0; j < 16; ++j) l[j >> 3][j & 7] = (_Float16)pv[j];
        }
    };

    // phase 3, row block qb.  Units: 0 bias read | 1-4 combine | 5 + bias | 6 scale | 7-12 split | 13 tile write |
    //                                14 tile read | 15-18 stores
    auto e3 = [&](int qb, const SplitAcc& c, int u) __attribute__((always_inline)) {
        if (u >= 1 && u <= 4) { combine4(c, u - 1); return; }
        if (MODE3 == 1 && qb < 8) {
            // q or k of head qb & 3: [pt][head][plane][32 dims]; this lane's dims 16 t + 8 hi .. + 7
            if (u == 0) load_bias16(bias3 + qb * 32);
            else if (u == 5) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 6) {
                if (qb < 4) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) pv[j] *= MDGAT_LOG2E * 0.17677669529663687f;   // log2(e) / sqrt(32) on q
                }
            } else if (u <= 12) split_unit(u - 7, sth, stl);
            else if (u == 13) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    *reinterpret_cast<f16x8_a*>(tile16 + l31 * QKROW + 16 * t + 8 * hi) = sth[t];
                    *reinterpret_cast<f16x8_a*>(tile16 + l31 * QKROW + 32 + 16 * t + 8 * hi) = stl[t];
                }
            } else if (u == 14) {
                // 128 contiguous bytes per keypoint: 8 lanes per row, 8 keypoints per store
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    pback[p] = *reinterpret_cast<const f32x4_a*>(tile16 + (8 * p + (lane >> 3)) * QKROW + (lane & 7) * 8);
            } else if (u <= 18) {
                const int p = u - 15;
                _Float16* dst = (qb < 4 ? a.q16 : a.k16) + (size_t)(qb & 3) * 64;
                const int row = 8 * p + (lane >> 3), cc = lane & 7;
                const int gp = min(wave_pt0 + row, a.R - 1);     // rows past the end: copies of the last keypoint
                *reinterpret_cast<f32x4*>(dst + (size_t)gp * 256 + cc * 8) = pback[p];
            }
        } else if (MODE3 == 1) {
            // v of head qb & 3 (non-swapped product): lane = dim l31, registers = keypoints mfma32_row(r, hi) of this wave
            const int head = qb & 3;
            const int P = a.N + a.M;
            if (((a.N | a.M) & 31) == 0) {
                // the 32 keypoints of the wave share frame and pair: 64 contiguous bytes per (plane, dim) row,
                // gathered through the tile so that a lane quad writes one row
                if (wave_pt0 < a.R) {
                    if (u == 0) pbias_v = bias3[qb * 32 + l31];
                    else if (u == 5) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) pv[j] = o[j] + pbias_v;
                    } else if (u >= 7 && u <= 12) split_unit(u - 7, sth, stl);
                    else if (u == 13) {
                        // registers 4 g .. 4 g + 3 = keypoints 8 g + 4 hi .. + 3 of this dim
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int t = g >> 1, e0 = 4 * (g & 1);
                            *reinterpret_cast<f16x4_a*>(tile16 + l31 * VROW + 8 * g + 4 * hi) =
                                f16x4{sth[t][e0], sth[t][e0 + 1], sth[t][e0 + 2], sth[t][e0 + 3]};
                            *reinterpret_cast<f16x4_a*>(tile16 + (32 + l31) * VROW + 8 * g + 4 * hi) =
                                f16x4{stl[t][e0], stl[t][e0 + 1], stl[t][e0 + 2], stl[t][e0 + 3]};
                        }
                    } else if (u == 14) {
#pragma unroll
                        for (int p = 0; p < 4; ++p)     // row = 32 plane + dim
                            pback[p] = *reinterpret_cast<const f32x4_a*>(tile16 + (16 * p + (lane >> 2)) * VROW + (lane & 3) * 8);
                    } else if (u >= 15 && u <= 18) {
                        const int p = u - 15;
                        const int bb = wave_pt0 / P, pp = wave_pt0 - bb * P;
                        const int col0 = pp < a.N ? pp : a.Npad + pp - a.N;
                        _Float16* base = a.vt16 + ((size_t)bb * 4 + head) * 64 * a.PP + col0;
                        const int row = 16 * p + (lane >> 2), cc = lane & 3;
                        *reinterpret_cast<f32x4*>(base + (size_t)row * a.PP + cc * 8) = pback[p];
                    }
                }
            } else if (u == 14) {
                // ragged frames: 4 consecutive keypoints per lane (8-byte stores) or single halves
                const float bias = bias3[qb * 32 + l31];
                const bool fast = ((a.N | a.M) & 3) == 0;      // 4 consecutive keypoints share frame and pair, 8-byte aligned
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
            }
        } else {
            if (u == 0) load_bias16(bias3 + qb * 32);
            else if (u == 5) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 6) {                 // whole rows go out after the last block
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = pv[8 * t + j];
                    store8(tile + l31 * TROW + qb * 32 + 16 * t + 8 * hi, v8);
                }
            }
        }
    };

    if (DO_MLP) {
        // ---- fragments of [x ; msg]: k-step ks covers channels 16 ks .. 16 ks + 15, this lane 8 hi .. 8 hi + 7 ----
        f16x8 ah[16], al[16];
        {
            f32x4 tm[16], tx[16];
            rows_load(a.msg, tm);
            rows_load(a.x, tx);
            rows_to_tile(tm);
#pragma unroll
            for (int ks = 8; ks < 16; ++ks) {
                float v[8];
                load8(tile + l31 * TROW + 16 * (ks & 7) + 8 * hi, v);
                split8s(v, ah[ks], al[ks]);
            }
            rows_to_tile(tx);                    // stays in the tile: residual of phase 2
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float v[8];
            load8(tile + l31 * TROW + 16 * ks + 8 * hi, v);
            split8s(v, ah[ks], al[ks]);
        }
        TR(1);
        stage_wait();
        TR(2);

        // ---- phase 1: 8 row blocks of W1 -> hidden fragments (k-steps 2 rb, 2 rb + 1 of phase 2) ----
        f16x8 hh[16], hl[16];
        // units: 0 bias read | 1-4 combine | 5 + bias | 6, 7 ReLU | 8-13 split
        auto e1 = [&](int rb, const SplitAcc& c, int u) __attribute__((always_inline)) {
            if (u == 0) load_bias16(bias1 + rb * 32);
            else if (u <= 4) combine4(c, u - 1);
            else if (u == 5) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 6 || u == 7) {
#pragma unroll
                for (int j = 8 * (u - 6); j < 8 * (u - 5); ++j) pv[j] = fmaxf(pv[j], 0.f);
            } else if (u <= 13) split_unit(u - 8, *reinterpret_cast<f16x8(*)[2]>(&hh[2 * rb]), *reinterpret_cast<f16x8(*)[2]>(&hl[2 * rb]));
        };
        // K = 256 blocks have 48 slots: a unit in every other one, a stage copy in the slot behind it
        for_units<8>([&](int rb) __attribute__((always_inline)) {
            const _Float16* wnext = rb < 7 ? a.w1s + (size_t)(rb + 1) * 32 * ROWH256 : a.w2s;
            TR(10);
            block_mma_il<16, true, ROWH256>(bufp(rb), wrow, hi, ah, al, acc[rb & 1], [&](int slot) __attribute__((always_inline)) {
                if ((slot & 1) && slot < 18) stage_dma_slice(wnext, ldsb(rb + 1), wave, lane, slot >> 1);
                if (!(slot & 1) && rb > 0) e1(rb - 1, acc[(rb - 1) & 1], slot >> 1);
            });
            TR(11);
            stage_wait();
            TR(12);
        });

        // ---- phase 2: 4 row blocks of W2, residual, new x (fp32 into the tile, split fragments kept) ----
        // units: 0 bias + residual read | 1-4 combine | 5 + bias | 6 + residual | 7 tile write | 8-13 split
        auto e2 = [&](int ob, const SplitAcc& c, int u) __attribute__((always_inline)) {
            if (u == 0) {
                load_bias16(bias2 + ob * 32);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float r8[8];
                    load8(tile + l31 * TROW + ob * 32 + 16 * t + 8 * hi, r8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) phf[8 * t + j] = r8[j];
                }
            } else if (u <= 4) combine4(c, u - 1);
            else if (u == 5) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 6) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] += phf[j];
            } else if (u == 7) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = pv[8 * t + j];
                    store8(tile + l31 * TROW + ob * 32 + 16 * t + 8 * hi, v8);
                }
            } else if (u <= 13) split_unit(u - 8, *reinterpret_cast<f16x8(*)[2]>(&xnh[2 * ob]), *reinterpret_cast<f16x8(*)[2]>(&xnl[2 * ob]));
        };
        for_units<4>([&](int ob) __attribute__((always_inline)) {
            const _Float16* wnext = ob < 3 ? a.w2s + (size_t)(ob + 1) * 32 * ROWH256 : a.w3s;
            TR(20);
            block_mma_il<16, true, ROWH256>(bufp(ob), wrow, hi, hh, hl, acc[ob & 1], [&](int slot) __attribute__((always_inline)) {
                if ((slot & 1) && slot < 18) stage_dma_slice(wnext, ldsb(ob + 1), wave, lane, slot >> 1);
                if (slot & 1) return;
                if (ob == 0) e1(7, acc[1], slot >> 1);
                else e2(ob - 1, acc[(ob - 1) & 1], slot >> 1);
            });
            TR(21);
            stage_wait();
            TR(22);
        });
        // the epilogue of the last block (set 1) is not overlapped: phase 3 needs all of the new x
        for_units<E_UNITS>([&](int u) __attribute__((always_inline)) { e2(3, acc[1], u); });
        TR(23);
        tile_to_rows(a.x);                       // the tile is free afterwards
        TR(24);
    } else {
        {
            f32x4 tx[16];
            rows_load(a.x, tx);
            rows_to_tile(tx);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float v[8];
            load8(tile + l31 * TROW + 16 * ks + 8 * hi, v);
            split8s(v, xnh[ks], xnl[ks]);
        }
        stage_wait();
    }

    // ---- phase 3: q | k | v of the next layer (12 row blocks) or the final projection (4), two row blocks
    //      per stage; after an even number of stages the first stage of W3 is in buffer 0 in both branches.
    //      K = 128 blocks have 24 slots for the 22 units; the stage copies ride in the first slot of a k-step ----
    for_units<NB3 / 2>([&](int j) __attribute__((always_inline)) {
        const bool more = j + 1 < NB3 / 2;
        const _Float16* wnext = a.w3s + (size_t)(j + 1) * 64 * ROWH128;
        const _Float16* cur = bufp(j);
        TR(30);
        const int qa = 2 * j, qb = 2 * j + 1;
        // block A (accumulator set 0); in its shadow: the epilogue of the previous block
        auto inter_a = [&](int slot) __attribute__((always_inline)) {
            if (more && slot % 3 == 0) stage_dma_slice(wnext, ldsb(j + 1), wave, lane, slot / 3);          // chunks 0 .. 7
            if (j > 0) e3(qa - 1, acc[1], slot);
        };
        if (MODE3 == 1 && qa >= 8) block_mma_il<8, false, ROWH128>(cur, l31, hi, xnh, xnl, acc[0], inter_a);
        else block_mma_il<8, true, ROWH128>(cur, wrow, hi, xnh, xnl, acc[0], inter_a);
        TR(31);
        // block B (set 1); in its shadow: the epilogue of block A
        auto inter_b = [&](int slot) __attribute__((always_inline)) {
            if (more && slot == 0) stage_dma_slice(wnext, ldsb(j + 1), wave, lane, 8);
            e3(qa, acc[0], slot);
        };
        if (MODE3 == 1 && qb >= 8) block_mma_il<8, false, ROWH128>(cur + 32 * ROWH128, l31, hi, xnh, xnl, acc[1], inter_b);
        else block_mma_il<8, true, ROWH128>(cur + 32 * ROWH128, wrow, hi, xnh, xnl, acc[1], inter_b);
        TR(32);
        if (more) stage_wait();
        TR(33);
    });
    for_units<E_UNITS>([&](int u) __attribute__((always_inline)) { e3(NB3 - 1, acc[1], u); });
    if (MODE3 != 1) tile_to_rows(a.mdesc);
    TR(40);
#ifdef LAYER_TRACE
    if (trace_on && lane == 0) for (int i = 0; i < (int)tlds[255]; ++i) g_dbg[tbase + i] = tlds[i];
#endif
}

// fp32 [rows][K] -> split image [rows][rowh] (hi plane | lo plane | pad), once per weight load
__global__ __launch_bounds__(256) void split_rows_kernel(const float* w, _Float16* out, int rows, int K, int rowh) {
    const size_t total = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / K, c = i - r * K;
        _Float16 h, l;
        mdgat_split(w[i], h, l);
        out[r * rowh + c] = h;
        out[r * rowh + K + c] = l;
    }
}

template <int DO_MLP, int MODE3>
int launch_layer_t(const LayerArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * STAGE_BYTES + (768 + 4 * TILE_FLOATS) * sizeof(float)
#ifdef LAYER_TRACE
        + 4 * 256 * 8
#endif
        ;
    static bool attr = false;
    if (!attr) {
        if (int rc = mdgat_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_kernel<DO_MLP, MODE3>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "layer LDS attribute"))
            return rc;
        attr = true;
    }
    hipLaunchKernelGGL((layer_kernel<DO_MLP, MODE3>), dim3((a.R + 127) / 128), dim3(256), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "layer launch");
}

}  // namespace

int launch_split_rows(const float* w, _Float16* out, int rows, int K, int rowh, hipStream_t s) {
    const size_t total = (size_t)rows * K;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(split_rows_kernel, dim3(blocks), dim3(256), 0, s, w, out, rows, K, rowh);
    return mdgat_check_hip(hipGetLastError(), "split_rows launch");
}

int launch_layer(const LayerLaunch& p, hipStream_t s) {
    if (p.R <= 0) return MDGAT_OK;
    LayerArgs a{};
    a.x = p.x; a.msg = p.msg;
    a.w1s = p.w1s; a.b1 = p.b1; a.w2s = p.w2s; a.b2 = p.b2; a.w3s = p.w3s; a.b3 = p.b3;
    a.q16 = p.out.q16; a.k16 = p.out.k16; a.vt16 = p.out.vt16; a.mdesc = p.mdesc;
    a.R = p.R; a.N = p.N; a.M = p.M; a.Npad = p.out.Npad; a.PP = p.out.PP;
    if (p.do_mlp) return p.mode3 == 1 ? launch_layer_t<1, 1>(a, s) : launch_layer_t<1, 2>(a, s);
    return p.mode3 == 1 ? launch_layer_t<0, 1>(a, s) : launch_layer_t<0, 2>(a, s);
}
