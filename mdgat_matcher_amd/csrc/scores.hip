// Score matrix of the optimal-transport layer: scores[b][i][j] = <mdesc0[b][i], mdesc1[b][j]> / sqrt(128)
// (mdgat.py:430-431, einsum 'bdn,bdm->bnm' over the 128 projected channels).
//
// Split-f16 products like the rest of the network (common.hpp): a 128 x 128 tile per workgroup, both operand
// tiles are split to f16 (hi | lo) on the way into LDS, 8 waves (2 per SIMD) of 64 x 32 outputs each, three
// v_mfma_f32_32x32x16_f16 per product.  12.9 GFLOP executed per launch at B = 64, N = M = 512.
#include "common.hpp"

namespace {

constexpr int SROW = 264;     // LDS row (halves): 128 hi | 128 lo | 8 pad (528 B: conflict-free ds_read_b128 fragments)

struct ScoreArgs {
    const float* A;           // [B] x [N][128] rows, batch stride sA floats
    const float* Bm;          // [B] x [M][128] rows, batch stride sB floats
    size_t sA, sB;
    float* scores;            // [B][N][M] = scale <A_i, B_j> - col_bias[b][j]
    int N, M;
    float scale;
    const float* col_bias;    // [B][M] or NULL (the kNN helper: |s_j|^2, pointops.hip)
    f32x4* zero;              // optional: zero_n 16-byte units cleared on the side (the exchange slots of the Sinkhorn kernel that
    size_t zero_n;            // runs next: spares the forward a memset launch)
    int B, tx, ty;            // pairs; 128-wide tiles per pair along columns / rows
    unsigned* guard;          // optional, host-mapped: set when an operand is outside the f16 operand range or not finite
};

// BIAS: subtract col_bias[b][j] (the kNN helper; kept out of the score-matrix instance, whose epilogue is store-bound)
template <bool BIAS>
__global__ __launch_bounds__(512) void scores_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];   // As[128][SROW] | Bs[128][SROW]
    _Float16* As = smem;
    _Float16* Bs = smem + 128 * SROW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    if (a.zero) {
        const size_t wg = blockIdx.x, nwg = gridDim.x;
        for (size_t i = wg * 512 + tid; i < a.zero_n; i += nwg * 512) a.zero[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // XCD-aware tile order: workgroup i runs on XCD i % 8 and every XCD has its own L2.  All tiles of a pair go to ONE XCD
    // (pair = 8 slot + i % 8), so each row of the two operand blocks is fetched from HBM once and re-read (tx / ty times)
    // from that L2 - with the plain (x, y, pair) order the tiles of a pair were spread over all XCDs and PMC FETCH_SIZE
    // showed 98 MB per launch for 33.5 MB of operands.  The grid is padded to 8 pairs per slot.
    const int tiles = a.tx * a.ty;
    const int q = blockIdx.x >> 3, b = (q / tiles) * 8 + (blockIdx.x & 7), t = q % tiles;
    if (b >= a.B) return;
    const int i0 = (t / a.tx) * 128, j0 = (t % a.tx) * 128;
    const float* A = a.A + (size_t)b * a.sA;
    const float* Bm = a.Bm + (size_t)b * a.sB;

    // ---- both operand tiles: fp32 rows -> (hi | lo) halves in LDS; 8 x 16-byte loads per thread and operand ----
    // f16 operand range guard: the largest integer image of the operands on their way to the split (NaN / inf on top).  The
    // final projection's output is checked HERE - whatever overflowed in the last layer arrives as inf / NaN - for every shape,
    // the streaming Sinkhorn path (frames beyond 2048 keypoints, which has no guard of its own) included.
    unsigned gmax = 0u;
    auto stage = [&](const float* src, int r0, int nrows, _Float16* dst) {
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = u * 512 + tid, row = idx >> 5, c = idx & 31;
            x[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(r0 + row, nrows - 1) * 128 + c * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) gmax = max(gmax, __builtin_bit_cast(unsigned, x[u][j]) & 0x7fffffffu);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = u * 512 + tid, row = idx >> 5, c = idx & 31;
            _Float16 h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mdgat_split(x[u][j], h[j], l[j]);
            *reinterpret_cast<f16x4*>(dst + row * SROW + c * 4) = f16x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<f16x4*>(dst + row * SROW + 128 + c * 4) = f16x4{l[0], l[1], l[2], l[3]};
        }
    };
    stage(A, i0, a.N, As);
    stage(Bm, j0, a.M, Bs);
    if (a.guard && gmax >= __builtin_bit_cast(unsigned, MDGAT_F16_GUARD)) __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();

    // ---- wave (wr, wc): rows 64 wr .. + 63, columns 32 wc .. + 31 of the tile ----
    const int wr = wave >> 2, wc = wave & 3;
    f32x16 acc[2], acx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acx[t][r] = 0.f; }
    const _Float16* ap = As + (64 * wr + l31) * SROW + 8 * hi;
    const _Float16* bp = Bs + (32 * wc + l31) * SROW + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + 16 * ks);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + 128 + 16 * ks);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap + 32 * t * SROW + 16 * ks);
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap + 32 * t * SROW + 128 + 16 * ks);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            acx[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acx[t], 0, 0, 0);
            acx[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acx[t], 0, 0, 0);
        }
    }
    // ---- D fragment: lane (column l31, hi) holds rows mfma32_row(r, hi); a half wave writes 128 contiguous bytes ----
    const int j = j0 + 32 * wc + l31;
    if (j < a.M) {
        float* out = a.scores + ((size_t)b * a.N) * a.M + j;
        const float bias = BIAS ? a.col_bias[(size_t)b * a.M + j] : 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + 64 * wr + 32 * t + mfma32_row(r, hi);
                if (i < a.N) out[(size_t)i * a.M] = BIAS ? fmaf(acx[t][r], MDGAT_SPLIT_INV, acc[t][r]) * a.scale - bias
                                                        : fmaf(acx[t][r], MDGAT_SPLIT_INV, acc[t][r]) * a.scale;
            }
    }
}

}  // namespace

int launch_dots(int B, int N, int M, const float* A, size_t strideA, const float* Bm, size_t strideB, float* out, float scale,
                const float* col_bias, hipStream_t s, void* zero, size_t zero_bytes, unsigned* guard) {
    if (B <= 0 || N <= 0 || M <= 0) return MDGAT_OK;
    const int tx = (M + 127) / 128, ty = (N + 127) / 128;
    ScoreArgs a{A, Bm, strideA, strideB, out, N, M, scale, col_bias, static_cast<f32x4*>(zero), zero ? zero_bytes / 16 : 0, B, tx, ty, guard};
    const unsigned grid = (unsigned)(((B + 7) / 8) * 8 * tx * ty);
    const size_t lds = (size_t)2 * 128 * SROW * sizeof(_Float16);
    static std::atomic<unsigned long long> optin[2];
    const void* kern = col_bias ? reinterpret_cast<const void*>(scores_kernel<true>) : reinterpret_cast<const void*>(scores_kernel<false>);
    if (int rc = mdgat_lds_optin(kern, lds, optin[col_bias != nullptr], "scores LDS attribute")) return rc;
    if (col_bias) hipLaunchKernelGGL(scores_kernel<true>, dim3(grid), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(scores_kernel<false>, dim3(grid), dim3(512), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "scores launch");
}

int launch_scores(int B, int N, int M, const float* mdesc, float* scores, float scale, hipStream_t s, void* zero, size_t zero_bytes, unsigned* guard) {
    const size_t P = (size_t)(N + M) * 128;
    return launch_dots(B, N, M, mdesc, P, mdesc + (size_t)N * 128, P, scores, scale, nullptr, s, zero, zero_bytes, guard);
}
