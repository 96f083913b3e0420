// Register-resident GEMM chain on the f16 matrix cores (layer.hip: a wave owns 16 keypoints, 16x16x32 MFMAs;
// encoder.hip: 32 keypoints, 32x32x16): split-f16 operands, products computed "swapped" so that the output fragment is the next product's B operand.
#pragma once
#include "common.hpp"

namespace {

__device__ __forceinline__ int perm32(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__device__ __forceinline__ void split8s(const float (&v)[8], f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = (_Float16)((v[j] - (float)h[j]) * MDGAT_SPLIT_SCALE);
}

typedef f32x4 __attribute__((may_alias)) f32x4_alias;   // LDS tiles are reused with other element types
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4_alias*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4_alias*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4_alias*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4_alias*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// D^T block (32 channels x 32 keypoints) = W block (LDS) . X^T (register fragments), NK k-steps of 16.
// SWAP: W is the A operand (row = channel perm), X the B operand; else X is A and W is B.
// The W fragments are read two k-steps ahead of the MFMAs that consume them (one wave per SIMD: nothing
// else hides the LDS latency); sched_barrier keeps the compiler from sinking the reads back to their use.
template <int NK, bool SWAP>
__device__ __forceinline__ void block_mma(const _Float16* buf, int wrow, int hi, const f16x8* xh, const f16x8* xl,
                                          f32x16& out) {
    constexpr int K = NK * 16, ROWH = 2 * K + 8;
    const _Float16* wp = buf + wrow * ROWH + 8 * hi;
    auto kcol = [&](int ks) { return 16 * ks; };
    f32x16 acc, aca, acb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; aca[r] = 0.f; acb[r] = 0.f; }
    f16x8 wh[NK], wl[NK];
    wh[0] = *reinterpret_cast<const f16x8*>(wp + kcol(0));
    wl[0] = *reinterpret_cast<const f16x8*>(wp + K + kcol(0));
    wh[1] = *reinterpret_cast<const f16x8*>(wp + kcol(1));
    wl[1] = *reinterpret_cast<const f16x8*>(wp + K + kcol(1));
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + 2 < NK) {
            wh[ks + 2] = *reinterpret_cast<const f16x8*>(wp + kcol(ks + 2));
            wl[ks + 2] = *reinterpret_cast<const f16x8*>(wp + K + kcol(ks + 2));
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads of k-step ks + 2 ahead of the MFMAs of k-step ks
        if (SWAP) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xh[ks], acc, 0, 0, 0);
            aca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xl[ks], aca, 0, 0, 0);
            acb = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], xh[ks], acb, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[ks], wh[ks], acc, 0, 0, 0);
            aca = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[ks], wh[ks], aca, 0, 0, 0);
            acb = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[ks], wl[ks], acb, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = fmaf(aca[r] + acb[r], MDGAT_SPLIT_INV, acc[r]);
}

}  // namespace
