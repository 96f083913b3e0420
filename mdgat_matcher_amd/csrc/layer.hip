// One attentional-propagation layer after the attention itself, fused with the NEXT layer's q/k/v
// projection, for 128 keypoints per workgroup - the activations never leave the register file:
//
//   phase 1  hid   = relu(W1 [x ; msg] + b1)      mlp.0 + folded BN + ReLU, merge folded in (mdgat.py:237, 247-248)
//   phase 2  x    += W2 hid + b2                  mlp.3 and the residual (mdgat.py:248, 274)
//   phase 3  q|k|v = Wqkv x + bqkv                proj[0..2] of the next layer (mdgat.py:227-232), written in
//                                                 the split-f16 operand layouts of the attention kernel
//            (last layer: mdesc = Wf x + bf, final_proj of mdgat.py:397, instead of q|k|v)
//
// Arithmetic: split-f16 products on the f16 matrix cores (common.hpp: x = hi + lo/2048, three MFMAs per
// product, fp32 accumulation) - fp32-class accuracy at 16/3 the f32 MFMA rate.
//
// gfx950 mapping.  Measured (tools/ubench/mfma_valu.hip): a wave cannot overlap its OWN vector instructions with
// its own matrix instructions - behind an 8-pass MFMA it issues next to nothing - while two waves of a SIMD overlap
// perfectly.  So the chain is cut to fit two waves per SIMD (<= 256 registers each): a wave owns 16 keypoints and
// works on v_mfma_f32_16x16x32_f16 tiles; 8 waves = 128 keypoints per workgroup, one workgroup per CU.
//
// Every product is computed "swapped" (D^T = W X^T): the B operand is the activation fragment (lane (n = keypoint,
// g) holds channels 8 g .. 8 g + 7 of a 32-deep k-step), the D fragment gives lane (n, g) rows 4 g .. 4 g + 3 of a
// 16-row block.  A 32-channel UNIT is two row blocks P and Q whose weight rows are ordered (at weight-split time,
// in the image itself) so that P row i is channel 8 (i >> 2) + (i & 3) and Q row i is channel 8 (i >> 2) + 4 + (i & 3):
// lane (n, g) then holds channels 8 g .. 8 g + 7 of the unit - exactly the B fragment of k-step `unit` of the next
// product.  relu(hid) and the new x are split to f16 in place and feed the next GEMM without touching memory.
//
// Weights: pre-split at load time into the LDS image itself ([row][hi plane | lo plane | 16 B pad], rows in
// P/Q order, the pad makes the ds_read_b128 fragment reads conflict free), so a stage (one unit of K = 256 or two
// units of K = 128, 33 KB) is a flat copy: global_load_lds_dwordx4 moves it L2 -> LDS without passing through
// registers, issued three stages ahead into a ring of five slots, one barrier per pair of stages (18 per tile).
//
// The epilogue of unit n (accumulator combine, bias, ReLU, f16 split, output staging and stores) is cut into
// small steps that are issued between the matrix instructions of unit n + 1 (unit_mma16: six slots per k-step),
// two accumulator sets alternate; the other wave of the SIMD fills the matrix pipe meanwhile.
//
// Global traffic is whole rows only: a wave-private LDS tile transposes between "a half wave / 8 lanes per
// contiguous row" (what the memory system wants) and the fragment order (what the MFMAs want).
// V uses the non-swapped product (a lane holds 4 consecutive keypoints of one dim) for the transposed V^T layout.
#include <utility>
#include <cstdlib>
#include "common.hpp"
#include "layer_image.hpp"
#ifdef LAYER_TRACE
__device__ long long g_dbg[8192];
extern "C" int mdgat_debug_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), n * sizeof(long long)); }
#define TRACE_OFF (5 * 17 * 1024 + (768 + 8 * 2112) * 4)
__device__ __forceinline__ void trace_point(int slot) {
    extern __shared__ __attribute__((aligned(16))) char tsm[];
    if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3)) {     // waves 0 and 3 only (LDS is nearly full)
        long long* tl = reinterpret_cast<long long*>(tsm + TRACE_OFF) + ((threadIdx.x >> 6) == 3) * 256;
        const int c = (int)tl[255];
        tl[c] = ((long long)slot << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffLL);
        tl[255] = c + 1;
    }
}
#define TR(slot) trace_point(slot)
#ifndef TRACE_MLP
#define TRACE_MLP 1
#endif
#else
#define TR(slot)
#endif
#include "mma_chain.hpp"
namespace {

// (image row pitches ROWH256 / ROWH128 and the kernel arguments: layer_image.hpp)
// A stage = 16 image rows of K = 256 (one row block, 16896 B) or 32 rows of K = 128 (one unit, 17408 B): 17 copies
// of 1 KB.  Ring of NSLOT slots; stages are consumed in pairs (the two row blocks of a K = 256 unit, two K = 128 units)
// with one barrier per pair; the copy of stage h + LOOKAHEAD is issued during stage h (the stage copies take
// about three stage times to land: every CU asks L2 for the same lines at the same time).
constexpr int SLOT_CHUNKS = 17;
constexpr int SLOT_BYTES = SLOT_CHUNKS * 1024;
constexpr int SLOT_HALVES = SLOT_BYTES / 2;
constexpr int NSLOT = 5, LOOKAHEAD = 3;
// waves per workgroup (16 keypoints each), template parameter NW: 8 (128 keypoints, two waves per SIMD: the throughput shape) or 4
// (64 keypoints, one wave per SIMD: small batches, where a launch has fewer 128-keypoint tiles than half the CUs - a wave that has
// its SIMD's matrix pipe to itself finishes its chain of 864 products sooner than two waves sharing it)
constexpr int WPTS = 16;                     // keypoints per wave
constexpr int TROW = 132;                    // floats per row of a wave's activation tile (128 channels + 16 B pad)
constexpr int TILE_FLOATS = WPTS * TROW;     // [16 keypoints][TROW]: 8448 B per wave
constexpr int QKROW = 72;                    // halves per row of the q/k store tile (32 hi | 32 lo | 16 B pad)
constexpr int VROW = 24;                     // halves per row of the V^T store tile (16 keypoints | 16 B pad)

// the store tiles are written as halves and read back as 16-byte pieces: accesses that must not be reordered
// on the strength of their types
typedef f16x8 __attribute__((may_alias)) f16x8_a;
typedef f16x4 __attribute__((may_alias)) f16x4_a;
typedef f32x4 __attribute__((may_alias)) f32x4_a;

// One stage: chunk c (1 KB) is moved by wave c & 7; the LDS address comes from M0, the lanes supply consecutive
// 16-byte pieces.  Inline asm: the compiler must not know that LDS is written (it would order every later ds_read
// behind the copy); completion is awaited explicitly (stage_wait) before the stage barrier.  The 3 copies of a
// wave are issued one at a time between the matrix instructions of the running stage.
template <int NW>
__device__ __forceinline__ void stage_dma_slice(const _Float16* g, unsigned lds_addr, int wave, int lane, int i) {
    const int c = min(wave + NW * i, SLOT_CHUNKS - 1);  // (the last chunk is copied more than once: no branch)
    // scalar base + per-lane 32-bit offset: the address arithmetic stays on the scalar unit
    const char* src = reinterpret_cast<const char*>(g) + c * 1024;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr + c * 1024), "v"(lane * 16), "s"(src) : "memory");
}
template <int NW> constexpr int dma_slices() { return (SLOT_CHUNKS + NW - 1) / NW; }
template <int NW>
__device__ __forceinline__ void stage_dma(const _Float16* g, unsigned lds_addr, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < dma_slices<NW>(); ++i) stage_dma_slice<NW>(g, lds_addr, wave, lane, i);
}
// f(0), f(1), ... f(N - 1) with literal arguments (a `#pragma unroll` loop over large inlined bodies is not reliably
// unrolled, and a rolled loop would index the register arrays of the epilogues dynamically)
template <typename F, int... U>
__device__ __forceinline__ void for_each_unit(F&& f, std::integer_sequence<int, U...>) { (f(U), ...); }
template <int N, typename F>
__device__ __forceinline__ void for_units(F&& f) { for_each_unit(f, std::make_integer_sequence<int, N>{}); }

// End of a stage: the copy of the NEXT stage must have landed; the copies issued after it (YOUNGER instructions:
// DMA_SLICES per later stage already under way) may stay in flight.  Vector memory operations retire in order, so
// "at most YOUNGER outstanding" says exactly that (stores issued in between only make the wait stricter).
template <int YOUNGER>
__device__ __forceinline__ void stage_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER) : "memory");
    __syncthreads();
}
// end of a PAIR of stages (last stage h1) that share one barrier: stages h1 + 1 and h1 + 2 must have landed, the copy of
// stage h1 + 3 (issued during h1) may be in flight.  Needs NSLOT >= 5: stage h + 3 and h + 4 land in the slots of the pair before.
template <int NW>
__device__ __forceinline__ void end_of_pair(int h1, int nstage) {
    if (h1 + 3 < nstage) stage_wait<dma_slices<NW>()>();
    else stage_wait<0>();
}

// Accumulators of one 32-channel unit: row blocks P and Q, each m = hi.hi and x = hi.lo + lo.hi (to be scaled by
// 1/2048 when combined).
struct UnitAcc { f32x4 pm, px, qm, qx; };

// One unit (32 image rows at `buf`) times NK32 k-steps of 32.  SWAP: W is the A operand, X (fragments xh / xl) the
// B operand; else X is A and W is B.  The W fragments are read AHEAD k-steps ahead; `inter(slot)` is issued behind
// each of the six matrix instructions of a k-step (slots 6 ks .. 6 ks + 5).
template <int NK32, bool SWAP, int ROWH, typename Inter>
__device__ __forceinline__ void unit_mma16(const _Float16* buf, int l15, int g, const f16x8* xh, const f16x8* xl,
                                           UnitAcc& acc, Inter&& inter) {
    constexpr int K = NK32 * 32, AHEAD = 2;
    const _Float16* wp = buf + l15 * ROWH + 8 * g;
    const _Float16* wq = wp + 16 * ROWH;
    acc.pm = f32x4{0.f, 0.f, 0.f, 0.f}; acc.px = acc.pm; acc.qm = acc.pm; acc.qx = acc.pm;
    f16x8 ph[NK32], pl[NK32], qh[NK32], ql[NK32];
#pragma unroll
    for (int ks = 0; ks < AHEAD; ++ks) {
        ph[ks] = *reinterpret_cast<const f16x8*>(wp + 32 * ks);
        pl[ks] = *reinterpret_cast<const f16x8*>(wp + K + 32 * ks);
        qh[ks] = *reinterpret_cast<const f16x8*>(wq + 32 * ks);
        ql[ks] = *reinterpret_cast<const f16x8*>(wq + K + 32 * ks);
    }
    auto mm = [&](const f16x8& w, const f16x8& x, const f32x4& c) __attribute__((always_inline)) {
        return SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_16x16x32_f16(x, w, c, 0, 0, 0);
    };
#pragma unroll
    for (int ks = 0; ks < NK32; ++ks) {
        if (ks + AHEAD < NK32) {
            ph[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wp + 32 * (ks + AHEAD));
            pl[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wp + K + 32 * (ks + AHEAD));
            qh[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wq + 32 * (ks + AHEAD));
            ql[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wq + K + 32 * (ks + AHEAD));
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads of k-step ks + AHEAD ahead of the MFMAs of k-step ks
        acc.px = mm(ph[ks], xl[ks], acc.px); inter(6 * ks); __builtin_amdgcn_sched_barrier(0);
        acc.qx = mm(qh[ks], xl[ks], acc.qx); inter(6 * ks + 1); __builtin_amdgcn_sched_barrier(0);
        acc.pm = mm(ph[ks], xh[ks], acc.pm); inter(6 * ks + 2); __builtin_amdgcn_sched_barrier(0);
        acc.qm = mm(qh[ks], xh[ks], acc.qm); inter(6 * ks + 3); __builtin_amdgcn_sched_barrier(0);
        acc.px = mm(pl[ks], xh[ks], acc.px); inter(6 * ks + 4); __builtin_amdgcn_sched_barrier(0);
        acc.qx = mm(ql[ks], xh[ks], acc.qx); inter(6 * ks + 5); __builtin_amdgcn_sched_barrier(0);
    }
}

// One row block (16 image rows at `buf`) of a K = 256 unit: accumulators m / x of that block, three slots per k-step.
template <int NK32, int ROWH, typename Inter>
__device__ __forceinline__ void block_mma16(const _Float16* buf, int l15, int g, const f16x8* xh, const f16x8* xl,
                                            f32x4& m, f32x4& x, Inter&& inter) {
    constexpr int K = NK32 * 32, AHEAD = 3;
    const _Float16* wp = buf + l15 * ROWH + 8 * g;
    m = f32x4{0.f, 0.f, 0.f, 0.f}; x = m;
    f16x8 ph[NK32], pl[NK32];
#pragma unroll
    for (int ks = 0; ks < AHEAD; ++ks) {
        ph[ks] = *reinterpret_cast<const f16x8*>(wp + 32 * ks);
        pl[ks] = *reinterpret_cast<const f16x8*>(wp + K + 32 * ks);
    }
#pragma unroll
    for (int ks = 0; ks < NK32; ++ks) {
        if (ks + AHEAD < NK32) {
            ph[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wp + 32 * (ks + AHEAD));
            pl[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wp + K + 32 * (ks + AHEAD));
        }
        __builtin_amdgcn_sched_barrier(0);
        x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[ks], xl[ks], x, 0, 0, 0); inter(3 * ks); __builtin_amdgcn_sched_barrier(0);
        m = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[ks], xh[ks], m, 0, 0, 0); inter(3 * ks + 1); __builtin_amdgcn_sched_barrier(0);
        x = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[ks], xh[ks], x, 0, 0, 0); inter(3 * ks + 2); __builtin_amdgcn_sched_barrier(0);
    }
}

// DO_MLP 0: phase 3 only (first layer / no layers).  MODE3 1: q|k|v (12 units), 2: final projection (4).
// VW 1 (MODE3 1, frames that are multiples of 128 keypoints): the V^T row pieces of the whole workgroup are gathered
// in LDS, so that each (plane, dim) row goes out as 256 contiguous bytes instead of eight 32-byte pieces.
template <int DO_MLP, int MODE3, int VW, int NW>
__global__ __launch_bounds__(64 * NW) void layer_kernel(LayerArgs a) {
    constexpr int NWAVE = NW, DMA_SLICES = dma_slices<NW>();
    static_assert(!VW || NW == 8, "the V^T gather buffers are laid out for 128-keypoint workgroups");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];   // 4 stage slots, 768 floats of biases, 8 tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // (no pointer tables here: a generic pointer loaded from a constant table is taken for a GLOBAL pointer)
    auto bufp = [&](int i) __attribute__((always_inline)) { return smem + (i % NSLOT) * SLOT_HALVES; };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)smem;
    auto ldsb = [&](int i) __attribute__((always_inline)) { return lds0 + (unsigned)(i % NSLOT) * SLOT_BYTES; };
    float* bias1 = reinterpret_cast<float*>(smem + NSLOT * SLOT_HALVES);   // [256]
    float* bias2 = bias1 + 256;                                         // [128]
    float* bias3 = bias2 + 128;                                         // [384]
    // Wave-private tile.  Same wave, in-order LDS: no barriers.  It holds x (fp32) through phases 1-2 (residual
    // source, new x written in place), then serves as the staging tile of the phase-3 outputs.
    float* tile = bias3 + 384 + wave * TILE_FLOATS;
    _Float16* tile16 = reinterpret_cast<_Float16*>(tile);
    constexpr int NT = 64 * NWAVE, TILE_PTS = WPTS * NWAVE;
    const int wave_pt0 = blockIdx.x * TILE_PTS + wave * WPTS;
    constexpr int NB3 = MODE3 == 1 ? 12 : 4;      // units of phase 3 (two per stage)

#ifdef LAYER_TRACE
    const bool trace_on = (blockIdx.x == 3 || blockIdx.x == gridDim.x - 2) && (wave == 0 || wave == 3) && DO_MLP == TRACE_MLP && MODE3 == 1;
    const int tbase = ((blockIdx.x != 3) * 2 + (wave == 3)) * 256;
    long long* tlds = reinterpret_cast<long long*>(reinterpret_cast<char*>(smem) + TRACE_OFF) + (wave == 3) * 256;
    if (lane == 0 && (wave == 0 || wave == 3)) tlds[255] = 0;
#endif
    // stages of the tile: 16 row blocks of W1, 8 of W2, then the units of W3
    constexpr int NSTAGE = (DO_MLP ? 24 : 0) + NB3;
    auto stage_src = [&](int h) __attribute__((always_inline)) -> const _Float16* {
        if (DO_MLP && h < 16) return a.w1s + (size_t)h * 16 * ROWH256;
        if (DO_MLP && h < 24) return a.w2s + (size_t)(h - 16) * 16 * ROWH256;
        return a.w3s + (size_t)(h - (DO_MLP ? 24 : 0)) * 32 * ROWH128;
    };
    // copy of stage h + LOOKAHEAD, slice i (issued during stage h)
    auto copy_ahead = [&](int h, int i) __attribute__((always_inline)) {
        if (h + LOOKAHEAD < NSTAGE) stage_dma_slice<NW>(stage_src(h + LOOKAHEAD), ldsb(h + LOOKAHEAD), wave, lane, i);
    };
    TR(0);
    // (unrolled: as run-time loops the 64-keypoint variant's two rounds of b3 went load / wait / store, one round trip each)
    if (DO_MLP) {
#pragma unroll
        for (int i = tid; i < 256; i += NT) bias1[i] = a.b1[i];
#pragma unroll
        for (int i = tid; i < 128; i += NT) bias2[i] = a.b2[i];
    }
#pragma unroll
    for (int i = tid; i < NB3 * 32; i += NT) bias3[i] = a.b3[i];

    // [R][128] fp32 rows of this wave's 16 keypoints <-> tile: half a wave per 512-byte row; all 8 loads of a
    // matrix are in flight together
    const int l31 = lane & 31, hi = lane >> 5;
    auto rows_load = [&](const float* src, f32x4 (&t)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gp = min(wave_pt0 + 2 * i + hi, a.R - 1);
            t[i] = *reinterpret_cast<const f32x4*>(src + (size_t)gp * 128 + l31 * 4);
        }
    };
    // f16 operand range guard: every activation becomes an f16 head + residual; a head beyond 65504 is
    // inf and everything after it NaN - which a ReLU (v_max) can swallow again.  The rows of x and msg pass through here
    // on their way to the split: the largest magnitude of the 64 values of a lane, compared as an unsigned integer (NaN and
    // inf have the largest images), flags the handle's status word.  msg carries what the attention kernels made of q, k, v
    // (an overflow there is inf - inf = NaN in the softmax), x the residual stream and, in the first launch, the encoders.
    auto rows_guard = [&](const f32x4 (&t)[8], unsigned acc) __attribute__((always_inline)) -> unsigned {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c < 4; c += 2)
                acc = max(acc, max(__builtin_bit_cast(unsigned, t[i][c]) & 0x7fffffffu, __builtin_bit_cast(unsigned, t[i][c + 1]) & 0x7fffffffu));
        return acc;
    };
    auto guard_report = [&](unsigned acc) __attribute__((always_inline)) {
        if (a.guard && acc >= __builtin_bit_cast(unsigned, MDGAT_F16_GUARD))
            __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto rows_to_tile = [&](const f32x4 (&t)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4_a*>(tile + (2 * i + hi) * TROW + l31 * 4) = t[i];
    };
    auto tile_to_rows = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 2 * i + hi;
            const f32x4 t = *reinterpret_cast<const f32x4_a*>(tile + p * TROW + l31 * 4);
            // rows past the end hold copies of the last keypoint (clamped loads): they rewrite the same values
            const int gp = min(wave_pt0 + p, a.R - 1);
            *reinterpret_cast<f32x4*>(dst + (size_t)gp * 128 + l31 * 4) = t;
        }
    };
    // fragment of k-step ks (channels 32 ks .. 32 ks + 31) of the fp32 rows in the tile
    auto tile_fragment = [&](int ks, f16x8& h, f16x8& l) __attribute__((always_inline)) {
        float v[8];
        load8(tile + l15 * TROW + 32 * ks + 8 * g, v);
        split8s(v, h, l);
    };

    UnitAcc acc[2];           // alternate between consecutive units
    float o[8];               // combined output of the unit whose epilogue is in flight: channels 8 g .. 8 g + 7
    f16x8 xnh[4], xnl[4];     // the (new) descriptors of this lane's keypoint as 4 k-step fragments

    // ---- epilogues as sequences of small STEPS placed in the slots of the next unit.  A step applies ONE
    //      operation to all 8 values of the unit (independent instructions); LDS operands are read a few steps
    //      before their use. ----
    float pbias[8], pv[8], phf[8];    // bias, values in flight, their f16 heads converted back
    float pbias_v[2] = {0.f, 0.f};
    f16x8 sth, stl;                   // split halves on their way to the store tile
    f32x4 pback[2];                   // store tile read back
    constexpr int E_STEPS = 16;

    auto combine = [&](const UnitAcc& c, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[4 * half + r] = half == 0 ? fmaf(c.px[r], MDGAT_SPLIT_INV, c.pm[r]) : fmaf(c.qx[r], MDGAT_SPLIT_INV, c.qm[r]);
    };
    auto load_bias8 = [&](const float* b) __attribute__((always_inline)) { load8(b + 8 * g, pbias); };
    // the split pv -> (h, l) in four steps (a fifth slot stays empty)
    // SCALED: the residual plane carries the factor 2048 (operands of this kernel's own products); the q / k / v planes
    // written for the attention kernels are UNSCALED (lo = f16(x - hi), f16 denormals included: the matrix cores honour
    // them), so that all three products of a contraction there go into one accumulator
    auto split_step = [&](int s_, f16x8& h, f16x8& l, bool SCALED = true) __attribute__((always_inline)) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if (s_ == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)pv[j];
        } else if (s_ == 1) {
            // residual pv - (float)h straight from the packed halves (one v_fma_mix_f32 per value)
            const u32x4 hp = __builtin_bit_cast(u32x4, h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(pv[2 * j]) : "v"(pv[2 * j]), "v"(hp[j]));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(pv[2 * j + 1]) : "v"(pv[2 * j + 1]), "v"(hp[j]));
            }
        } else if (s_ == 2) {
            if (SCALED) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] *= MDGAT_SPLIT_SCALE;
            }
        } else if (s_ == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j] = (_Float16)pv[j];
        }
    };

    // VW: element (row, col) of gather buffer b, and the store of this wave's 8 rows (piece p: rows 4 p .. 4 p + 3,
    // 16 lanes x 16 bytes per row) of unit qb
    constexpr int VSH_OFF = 16 * QKROW, VSHROW = 136, VSH_BUF = 8 * VSHROW;
    _Float16* tiles16 = reinterpret_cast<_Float16*>(bias3 + 384);
    auto vsh = [&](int b, int row, int col) __attribute__((always_inline)) {
        return tiles16 + (row >> 3) * (2 * TILE_FLOATS) + VSH_OFF + b * VSH_BUF + (row & 7) * VSHROW + col;
    };
    auto vstore = [&](int qb, int u) __attribute__((always_inline)) {
        const int p = u >> 1, rl = 4 * p + g;
        if ((u & 1) == 0) pback[p] = *reinterpret_cast<const f32x4_a*>(tile16 + VSH_OFF + (qb & 1) * VSH_BUF + rl * VSHROW + l15 * 8);
        else {
            const int P = a.N + a.M;
            const int pt0 = blockIdx.x * (WPTS * NWAVE);
            const int bb = pt0 / P, pp = pt0 - bb * P;
            const int col0 = pp < a.N ? pp : a.Npad + pp - a.N;
            _Float16* base = a.vt16 + (((size_t)bb * 4 + (qb & 3)) * 64 + 8 * wave + rl) * a.PP + col0;
            *reinterpret_cast<f32x4*>(base + l15 * 8) = pback[p];
        }
    };

    // phase 3, unit qb.  Steps: 0 bias read | 1, 2 combine | 3 + bias | 4 scale | 5-9 split | 10 tile write |
    //                           11 tile read | 12, 13 stores
    auto e3 = [&](int qb, const UnitAcc& c, int u) __attribute__((always_inline)) {
        if (u == 1 || u == 2) { combine(c, u - 1); return; }
        if (MODE3 == 1 && qb < 8) {
            // q or k of head qb & 3: [pt][head][plane][32 dims]; this lane's dims 8 g .. 8 g + 7
            if (u == 0) load_bias8(bias3 + qb * 32);
            else if (u == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 4) {
                if (qb < 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] *= MDGAT_LOG2E * 0.17677669529663687f;   // log2(e) / sqrt(32) on q
                }
            } else if (u <= 9) split_step(u - 5, sth, stl, false);
            else if (u == 10) {
                *reinterpret_cast<f16x8_a*>(tile16 + l15 * QKROW + 8 * g) = sth;
                *reinterpret_cast<f16x8_a*>(tile16 + l15 * QKROW + 32 + 8 * g) = stl;
            } else if (u == 11) {
                // 128 contiguous bytes per keypoint: 8 lanes per row, 8 keypoints per store
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    pback[p] = *reinterpret_cast<const f32x4_a*>(tile16 + (8 * p + (lane >> 3)) * QKROW + (lane & 7) * 8);
            } else if (u <= 13) {
                const int p = u - 12;
                _Float16* dst = (qb < 4 ? a.q16 : a.k16) + (size_t)(qb & 3) * 64;
                const int row = 8 * p + (lane >> 3), cc = lane & 7;
                const int gp = min(wave_pt0 + row, a.R - 1);     // rows past the end: copies of the last keypoint
                *reinterpret_cast<f32x4*>(dst + (size_t)gp * 256 + cc * 8) = pback[p];
            }
        } else if (MODE3 == 1) {
            // v of head qb & 3 (non-swapped product): lane (n, g) = dims n (block P) and 16 + n (block Q), keypoints
            // 4 g .. 4 g + 3 of this wave: o[0..3] / o[4..7]
            const int head = qb & 3;
            const int P = a.N + a.M;
            if (VW) {
                // shared gather: row 32 plane + dim of the workgroup lives in the tile of wave row >> 3, above the
                // q/k staging rows; two buffers (qb & 1).  This wave fills columns 16 wave + 4 g .. + 3 of all 64 rows.
                if (u == 0) { pbias_v[0] = bias3[qb * 32 + l15]; pbias_v[1] = bias3[qb * 32 + 16 + l15]; }
                else if (u == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = o[j] + pbias_v[j >> 2];
                } else if (u >= 5 && u <= 9) split_step(u - 5, sth, stl, false);
                else if (u == 10) {
                    _Float16* c0 = vsh(qb & 1, l15, 16 * wave + 4 * g);
                    *reinterpret_cast<f16x4_a*>(c0) = f16x4{sth[0], sth[1], sth[2], sth[3]};
                    *reinterpret_cast<f16x4_a*>(c0 + 2 * 2 * TILE_FLOATS) = f16x4{sth[4], sth[5], sth[6], sth[7]};
                    *reinterpret_cast<f16x4_a*>(c0 + 4 * 2 * TILE_FLOATS) = f16x4{stl[0], stl[1], stl[2], stl[3]};
                    *reinterpret_cast<f16x4_a*>(c0 + 6 * 2 * TILE_FLOATS) = f16x4{stl[4], stl[5], stl[6], stl[7]};
                }
            } else if (((a.N | a.M) & 15) == 0) {
                // the 16 keypoints of the wave share frame and pair: 32 contiguous bytes per (plane, dim) row,
                // gathered through the tile so that a lane writes one whole row
                if (wave_pt0 < a.R) {
                    if (u == 0) { pbias_v[0] = bias3[qb * 32 + l15]; pbias_v[1] = bias3[qb * 32 + 16 + l15]; }
                    else if (u == 3) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) pv[j] = o[j] + pbias_v[j >> 2];
                    } else if (u >= 5 && u <= 9) split_step(u - 5, sth, stl, false);
                    else if (u == 10) {
                        // tile rows: 32 plane + dim; 16 keypoints (halves) per row
                        *reinterpret_cast<f16x4_a*>(tile16 + l15 * VROW + 4 * g) = f16x4{sth[0], sth[1], sth[2], sth[3]};
                        *reinterpret_cast<f16x4_a*>(tile16 + (16 + l15) * VROW + 4 * g) = f16x4{sth[4], sth[5], sth[6], sth[7]};
                        *reinterpret_cast<f16x4_a*>(tile16 + (32 + l15) * VROW + 4 * g) = f16x4{stl[0], stl[1], stl[2], stl[3]};
                        *reinterpret_cast<f16x4_a*>(tile16 + (48 + l15) * VROW + 4 * g) = f16x4{stl[4], stl[5], stl[6], stl[7]};
                    } else if (u == 11) {
#pragma unroll
                        for (int p = 0; p < 2; ++p)     // lane = row (32 plane + dim), two 16-byte halves of it
                            pback[p] = *reinterpret_cast<const f32x4_a*>(tile16 + lane * VROW + p * 8);
                    } else if (u == 12 || u == 13) {
                        const int p = u - 12;
                        const int bb = wave_pt0 / P, pp = wave_pt0 - bb * P;
                        const int col0 = pp < a.N ? pp : a.Npad + pp - a.N;
                        _Float16* base = a.vt16 + ((size_t)bb * 4 + head) * 64 * a.PP + col0;
                        *reinterpret_cast<f32x4*>(base + (size_t)lane * a.PP + p * 8) = pback[p];
                    }
                }
            } else if (u == 11) {
                // ragged frames: 4 consecutive keypoints per lane (8-byte stores) or single halves
                const bool fast = ((a.N | a.M) & 3) == 0;      // 4 consecutive keypoints share frame and pair, 8-byte aligned
                const int p0 = wave_pt0 + 4 * g;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    if (p0 >= a.R) continue;
                    const int dim = 16 * blk + l15;
                    const float bias = bias3[qb * 32 + dim];
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) mdgat_split_unscaled(o[4 * blk + j] + bias, h[j], l[j]);
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
        } else {
            if (u == 0) load_bias8(bias3 + qb * 32);
            else if (u == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 4) store8(tile + l15 * TROW + qb * 32 + 8 * g, pv);   // whole rows go out after the last unit
        }
    };

    if (DO_MLP) {
        // ---- fragments of [x ; msg]: k-step ks covers channels 32 ks .. 32 ks + 31, this lane 8 g .. 8 g + 7 ----
        f16x8 ah[8], al[8];
        {
            f32x4 tm[8], tx[8];
            rows_load(a.msg, tm);
            rows_load(a.x, tx);
            for_units<LOOKAHEAD>([&](int h) __attribute__((always_inline)) { stage_dma<NW>(stage_src(h), ldsb(h), wave, lane); });
            rows_to_tile(tm);
            const unsigned gm = rows_guard(tm, 0u);
            for_units<4>([&](int ks) __attribute__((always_inline)) { tile_fragment(ks, ah[4 + ks], al[4 + ks]); });
            rows_to_tile(tx);                    // stays in the tile: residual of phase 2
            guard_report(rows_guard(tx, gm));
        }
        for_units<4>([&](int ks) __attribute__((always_inline)) { tile_fragment(ks, ah[ks], al[ks]); });
        TR(1);
        stage_wait<DMA_SLICES>();               // stages 0 and 1 have landed, stage 2 may be in flight
        TR(2);

        // ---- phase 1: 8 units of W1 -> hidden fragments (k-step rb of phase 2) ----
        f16x8 hh[8], hl[8];
        // steps: 0 bias read | 1, 2 combine | 3 + bias | 4 ReLU | 5-9 split
        auto e1 = [&](int rb, const UnitAcc& c, int u) __attribute__((always_inline)) {
            if (u == 0) load_bias8(bias1 + rb * 32);
            else if (u <= 2) combine(c, u - 1);
            else if (u == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = fmaxf(pv[j], 0.f);
            } else if (u <= 9) split_step(u - 5, hh[rb], hl[rb]);
        };
        // K = 256 units have 48 slots: a step in every other one, the stage copies in slots 6 i + 1
        // end of stage h: the copies of stages h + 2 and h + 3 may still be in flight
        for_units<8>([&](int rb) __attribute__((always_inline)) {
            TR(10);
            // slots 0 .. 23: row block P, 24 .. 47: row block Q; the pending epilogue in every other slot
            auto inter = [&](int h, int base) __attribute__((always_inline)) {
                return [&, h, base](int slot) __attribute__((always_inline)) {
                    if (slot % 4 == 1 && slot / 4 < DMA_SLICES) copy_ahead(h, slot / 4);
                    if (slot % 2 == 0 && rb > 0) e1(rb - 1, acc[(rb - 1) & 1], (base + slot) / 2);
                };
            };
            block_mma16<8, ROWH256>(bufp(2 * rb), l15, g, ah, al, acc[rb & 1].pm, acc[rb & 1].px, inter(2 * rb, 0));
            block_mma16<8, ROWH256>(bufp(2 * rb + 1), l15, g, ah, al, acc[rb & 1].qm, acc[rb & 1].qx, inter(2 * rb + 1, 24));
            TR(11);
            end_of_pair<NW>(2 * rb + 1, NSTAGE);
            TR(12);
        });

        // ---- phase 2: 4 units of W2, residual, new x (fp32 into the tile, split fragments kept) ----
        // steps: 0 bias + residual read | 1, 2 combine | 3 + bias | 4 + residual | 5 tile write | 6-10 split
        auto e2 = [&](int ob, const UnitAcc& c, int u) __attribute__((always_inline)) {
            if (u == 0) {
                load_bias8(bias2 + ob * 32);
                load8(tile + l15 * TROW + ob * 32 + 8 * g, phf);
            } else if (u <= 2) combine(c, u - 1);
            else if (u == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = o[j] + pbias[j];
            } else if (u == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] += phf[j];
            } else if (u == 5) store8(tile + l15 * TROW + ob * 32 + 8 * g, pv);
            else if (u <= 10) split_step(u - 6, xnh[ob], xnl[ob]);
        };
        for_units<4>([&](int ob) __attribute__((always_inline)) {
            constexpr int H0 = 16;
            TR(20);
            auto inter = [&](int h, int base) __attribute__((always_inline)) {
                return [&, h, base](int slot) __attribute__((always_inline)) {
                    if (slot % 4 == 1 && slot / 4 < DMA_SLICES) copy_ahead(h, slot / 4);
                    if (slot % 2 != 0) return;
                    if (ob == 0) e1(7, acc[1], (base + slot) / 2);
                    else e2(ob - 1, acc[(ob - 1) & 1], (base + slot) / 2);
                };
            };
            block_mma16<8, ROWH256>(bufp(H0 + 2 * ob), l15, g, hh, hl, acc[ob & 1].pm, acc[ob & 1].px, inter(H0 + 2 * ob, 0));
            block_mma16<8, ROWH256>(bufp(H0 + 2 * ob + 1), l15, g, hh, hl, acc[ob & 1].qm, acc[ob & 1].qx, inter(H0 + 2 * ob + 1, 24));
            TR(21);
            end_of_pair<NW>(H0 + 2 * ob + 1, NSTAGE);
            TR(22);
        });
        // the epilogue of the last unit (set 1) is not overlapped: phase 3 needs all of the new x
        for_units<E_STEPS>([&](int u) __attribute__((always_inline)) { e2(3, acc[1], u); });
        TR(23);
        tile_to_rows(a.x);                       // the tile is free afterwards
        TR(24);
    } else {
        {
            f32x4 tx[8];
            rows_load(a.x, tx);
            for_units<LOOKAHEAD>([&](int h) __attribute__((always_inline)) { stage_dma<NW>(stage_src(h), ldsb(h), wave, lane); });
            rows_to_tile(tx);
            guard_report(rows_guard(tx, 0u));
        }
        for_units<4>([&](int ks) __attribute__((always_inline)) { tile_fragment(ks, xnh[ks], xnl[ks]); });
        stage_wait<DMA_SLICES>();
    }

    // ---- phase 3: q | k | v of the next layer (12 units) or the final projection (4), one unit per stage.
    //      K = 128 units have 24 slots: a step of the previous unit's epilogue in every slot up to E_STEPS ----
    for_units<NB3>([&](int q) __attribute__((always_inline)) {
        constexpr int H0 = DO_MLP ? 24 : 0;
        const _Float16* cur = bufp(H0 + q);
        TR(30);
        auto inter = [&](int slot) __attribute__((always_inline)) {
            if (slot % 4 == 1 && slot / 4 < DMA_SLICES) copy_ahead(H0 + q, slot / 4);
            if (q > 0) e3(q - 1, acc[(q - 1) & 1], slot);
            if (VW && q >= 10 && slot >= 18 && slot < 22) vstore(q - 2, slot - 18);
        };
        if (MODE3 == 1 && q >= 8) unit_mma16<4, false, ROWH128>(cur, l15, g, xnh, xnl, acc[q & 1], inter);
        else unit_mma16<4, true, ROWH128>(cur, l15, g, xnh, xnl, acc[q & 1], inter);
        TR(32);
        if ((q & 1) && q + 1 < NB3) end_of_pair<NW>(H0 + q, NSTAGE);
        else if (VW && q >= 10) __syncthreads();       // the gather buffers change hands every unit from here on
        TR(33);
    });
    if (VW) for_units<4>([&](int u) __attribute__((always_inline)) { vstore(NB3 - 2, u); });
    for_units<E_STEPS>([&](int u) __attribute__((always_inline)) { e3(NB3 - 1, acc[(NB3 - 1) & 1], u); });
    if (VW) {
        __syncthreads();
        for_units<4>([&](int u) __attribute__((always_inline)) { vstore(NB3 - 1, u); });
    }
    if (MODE3 != 1) tile_to_rows(a.mdesc);
    TR(40);
#ifdef LAYER_TRACE
    if (trace_on && lane == 0) for (int i = 0; i < (int)tlds[255]; ++i) g_dbg[tbase + i] = tlds[i];
#endif
}

// fp32 [rows][K] -> split image [rows][rowh] (hi plane | lo plane | pad), once per weight load.  The first `nperm`
// rows (a multiple of 32) go out in the P/Q order of layer_kernel: within a unit of 32 channels, image row j < 16
// holds channel 8 (j >> 2) + (j & 3), image row 16 + j holds channel 8 (j >> 2) + 4 + (j & 3).
__global__ __launch_bounds__(256) void split_rows_kernel(const float* w, _Float16* out, int rows, int K, int rowh, int nperm) {
    const size_t total = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / K), c = (int)(i - (size_t)r * K);      // image row, column
        int src = r;
        if (r < nperm) {
            const int j = r & 15, q = (r >> 4) & 1;
            src = (r & ~31) + 8 * (j >> 2) + 4 * q + (j & 3);
        }
        _Float16 h, l;
        mdgat_split(w[(size_t)src * K + c], h, l);
        out[(size_t)r * rowh + c] = h;
        out[(size_t)r * rowh + K + c] = l;
    }
}

template <int DO_MLP, int MODE3, int VW, int NW>
int launch_layer_t(const LayerArgs& a, hipStream_t s) {
    constexpr int NWAVE = NW;
    const size_t lds = (size_t)NSLOT * SLOT_BYTES + (768 + NWAVE * TILE_FLOATS) * sizeof(float)
#ifdef LAYER_TRACE
        + 2 * 256 * 8
#endif
        ;
    static std::atomic<unsigned long long> optin;        // (one per template instance)
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(layer_kernel<DO_MLP, MODE3, VW, NW>), lds, optin, "layer LDS attribute")) return rc;
    constexpr int TILE_PTS = WPTS * NWAVE;
    hipLaunchKernelGGL((layer_kernel<DO_MLP, MODE3, VW, NW>), dim3((a.R + TILE_PTS - 1) / TILE_PTS), dim3(64 * NWAVE), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "layer launch");
}

}  // namespace

int launch_split_rows(const float* w, _Float16* out, int rows, int K, int rowh, int nperm, hipStream_t s) {
    const size_t total = (size_t)rows * K;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(split_rows_kernel, dim3(blocks), dim3(256), 0, s, w, out, rows, K, rowh, nperm);
    return mdgat_check_hip(hipGetLastError(), "split_rows launch");
}

// launches of at most this many 128-keypoint tiles run layer_split.hip (MDGAT_LAYER_SPLIT_TILES; 0: never)
static std::atomic<int> g_split_tiles{[] { const char* e = getenv("MDGAT_LAYER_SPLIT_TILES"); return e ? atoi(e) : MDGAT_LAYER_SPLIT_TILES_DEFAULT; }()};
extern "C" int mdgat_set_layer_split_tiles(int tiles) { return g_split_tiles.exchange(tiles < 0 ? MDGAT_LAYER_SPLIT_TILES_DEFAULT : tiles); }

int launch_layer(const LayerLaunch& p, hipStream_t s) {
    if (p.R <= 0) return MDGAT_OK;
    LayerArgs a{};
    a.x = p.x; a.msg = p.msg;
    a.w1s = p.w1s; a.b1 = p.b1; a.w2s = p.w2s; a.b2 = p.b2; a.w3s = p.w3s; a.b3 = p.b3;
    a.w1f = p.w1f; a.w2f = p.w2f; a.w3f = p.w3f;
    a.q16 = p.out.q16; a.k16 = p.out.k16; a.vt16 = p.out.vt16; a.mdesc = p.mdesc;
    a.R = p.R; a.N = p.N; a.M = p.M; a.Npad = p.out.Npad; a.PP = p.out.PP; a.guard = p.guard;
    // launches of a few tiles (one pair, small batches): the channel-split kernel of layer_split.hip - 32-keypoint workgroups
    // whose eight waves share the output channels; bit-identical results (mdgat_set_layer_split_tiles: tuning / A-B hook)
    if ((p.R + 127) / 128 <= g_split_tiles.load(std::memory_order_relaxed)) return launch_layer_split(a, p.do_mlp, p.mode3, s);
    // small launches (fewer 128-keypoint tiles than half the CUs of the part): 64-keypoint workgroups, one wave per SIMD
    static const int small_tiles = [] { const char* e = getenv("MDGAT_LAYER_SMALL_TILES"); return e ? atoi(e) : 128; }();
    if ((p.R + 127) / 128 <= small_tiles) {
        if (p.do_mlp) return p.mode3 != 1 ? launch_layer_t<1, 2, 0, 4>(a, s) : launch_layer_t<1, 1, 0, 4>(a, s);
        return p.mode3 != 1 ? launch_layer_t<0, 2, 0, 4>(a, s) : launch_layer_t<0, 1, 0, 4>(a, s);
    }
    const bool vw = p.mode3 == 1 && ((p.N | p.M) & 127) == 0;
    if (p.do_mlp) return p.mode3 != 1 ? launch_layer_t<1, 2, 0, 8>(a, s) : vw ? launch_layer_t<1, 1, 1, 8>(a, s) : launch_layer_t<1, 1, 0, 8>(a, s);
    return p.mode3 != 1 ? launch_layer_t<0, 2, 0, 8>(a, s) : vw ? launch_layer_t<0, 1, 1, 8>(a, s) : launch_layer_t<0, 1, 0, 8>(a, s);
}
