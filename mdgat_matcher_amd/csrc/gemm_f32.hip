// Per-point linear layers (Conv1d k=1 with folded BatchNorm) and the score contraction as an fp32
// MFMA GEMM:  C[m][n] = act(scale * sum_k A[m][k] * W[n][k] + bias[n]) (+ R[m][n]).
//
// Replaces: MLP / Conv1d(k=1) stacks of mdgat.py:34-46 (after BN folding), proj[i] of 227-232,
// mlp of 247-248 with the residual of 274, final_proj of 397 and the einsum of 430-431.
//
// gfx950 mapping: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A workgroup is 4 waves
// in a 2x2 arrangement computing a 128x128 tile; each wave owns 64x64 = 2x2 MFMA tiles (64
// accumulator registers).  Both operands are K-contiguous ("NT" GEMM), so a lane's 16 operand values
// for one 32-deep K chunk are 64 contiguous bytes of one row: they are fetched straight into
// registers with four global_load_dwordx4 (each 128-byte line is consumed whole by the lane pair
// (l, l+32)), no LDS and no barrier.  The MFMA k-slot (step t, half hi) is mapped to column
// kc + 16*hi + t for both operands - any bijection works as long as A and W agree.  The next chunk's
// fragments are loaded while the current chunk's 64 MFMAs (4096 cycles) run.
#include "common.hpp"

namespace {

struct Frag {
    float a[2][16];
    float w[2][16];
};

__device__ __forceinline__ void load_frag(Frag& f, const float* const (&pa)[2], const float* const (&pw)[2], int koff_a,
                                          int koff_w) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x4 x = *reinterpret_cast<const f32x4*>(pa[i] + koff_a + 4 * v);
            f.a[i][4 * v + 0] = x[0]; f.a[i][4 * v + 1] = x[1]; f.a[i][4 * v + 2] = x[2]; f.a[i][4 * v + 3] = x[3];
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x4 x = *reinterpret_cast<const f32x4*>(pw[j] + koff_w + 4 * v);
            f.w[j][4 * v + 0] = x[0]; f.w[j][4 * v + 1] = x[1]; f.w[j][4 * v + 2] = x[2]; f.w[j][4 * v + 3] = x[3];
        }
    }
}

__device__ __forceinline__ void mma_chunk(const Frag& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][t], f.w[j][t], acc[i][j], 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.y * 128 + (wave & 1) * 64;
    if (m0 >= g.M || n0 >= g.N) return;   // wave-uniform
    const long long z = blockIdx.z;
    const float* A0 = g.A0 + z * g.sA;
    const float* W = g.W + z * g.sW;
    float* C = g.C + z * g.sC;

    int ra[2], rw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ra[i] = min(m0 + i * 32 + l31, g.M - 1);
        rw[i] = min(n0 + i * 32 + l31, g.N - 1);
    }
    const float* pa0[2] = {A0 + (size_t)ra[0] * g.lda0 + hi * 16, A0 + (size_t)ra[1] * g.lda0 + hi * 16};
    const float* pa1[2] = {g.A1 ? g.A1 + (size_t)ra[0] * g.lda1 + hi * 16 : nullptr,
                           g.A1 ? g.A1 + (size_t)ra[1] * g.lda1 + hi * 16 : nullptr};
    const float* pw[2] = {W + (size_t)rw[0] * g.ldw + hi * 16, W + (size_t)rw[1] * g.ldw + hi * 16};

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int K = g.K, K0 = g.K0;
    Frag fa, fb;
    auto load = [&](Frag& f, int kc) {
        if (kc < K0) load_frag(f, pa0, pw, kc, kc);
        else load_frag(f, pa1, pw, kc - K0, kc);
    };
    load(fa, 0);
    for (int kc = 0; kc < K; kc += 64) {
        const bool has_b = kc + 32 < K;
        if (has_b) load(fb, kc + 32);
        mma_chunk(fa, acc);
        if (kc + 64 < K) load(fa, kc + 64);
        if (has_b) mma_chunk(fb, acc);
    }

    // epilogue in the C/D fragment layout: lane holds column n0 + 32 j + l31, rows mfma32_row(r, hi)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 32 + l31;
        const bool col_ok = col < g.N;
        const float b = (g.bias && col_ok) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + mfma32_row(r, hi);
                if (col_ok && row < g.M) {
                    float v = acc[i][j][r] * g.scale + b;
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (g.R) v += g.R[(size_t)row * g.ldr + col];
                    C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return MDGAT_OK;
    if (a.K % 32 != 0 || a.K0 % 32 != 0 || a.K0 > a.K || (a.K0 < a.K && !a.A1) || (a.lda0 % 4) || (a.ldw % 4) ||
        (a.A1 && (a.lda1 % 4))) {
        mdgat_set_error("gemm: unsupported K=%d K0=%d lda=%d ldw=%d", a.K, a.K0, a.lda0, a.ldw);
        return MDGAT_ERR_UNSUPPORTED;
    }
    dim3 grid((a.M + 127) / 128, (a.N + 127) / 128, a.batch);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, a);
    return mdgat_check_hip(hipGetLastError(), "gemm launch");
}
