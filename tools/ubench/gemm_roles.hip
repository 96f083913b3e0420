// Microbenchmark: which operand of the split-f16 GEMM chain should live in LDS?
//   A (layer.hip today): weight fragments from LDS, activations in registers - a wave owns 16 keypoints and walks
//     ALL output rows: per k-step and 16-row block 2 ds_read_b128 feed 3 MFMAs (16x16x32).
//   B (candidate): activations of 128 keypoints in LDS, weights straight from L2 into registers - a wave owns 32 output
//     rows and walks ALL 8 keypoint tiles: per k-step 16 ds_read_b128 + 4 global loads feed 48 MFMAs.
// Both do the 256 x 256 product for 128 keypoints per workgroup (3072 MFMAs, ideal 12288 cycles on 4 SIMDs), REPS times.
//   hipcc --offload-arch=gfx950 -O3 gemm_roles.hip -o gemm_roles.bin && ./gemm_roles.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int ROWH = 528;      // halves per row: 256 hi | 256 lo | 16 pad

// ---- A: weights in LDS (64 image rows resident, reused for the 16 row blocks), activations in registers
__global__ __launch_bounds__(512) void roles_a(const _Float16* w, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 64 * ROWH / 8; i += 512) reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(w)[i];
    __syncthreads();
    f16x8 xh[8], xl[8];
    for (int ks = 0; ks < 8; ++ks)
        for (int j = 0; j < 8; ++j) { xh[ks][j] = (_Float16)(0.01f * (tid + j + ks)); xl[ks][j] = (_Float16)(0.02f * (tid - j + ks)); }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int blk = 0; blk < 16; ++blk) {
            const _Float16* wp = smem + ((blk & 3) * 16 + l15) * ROWH + 8 * g;
            f32x4 m = {0.f, 0.f, 0.f, 0.f}, x = m;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f16x8 ph = *reinterpret_cast<const f16x8*>(wp + 32 * ks);
                const f16x8 pl = *reinterpret_cast<const f16x8*>(wp + 256 + 32 * ks);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, xl[ks], x, 0, 0, 0);
                m = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, xh[ks], m, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, xh[ks], x, 0, 0, 0);
            }
            sum += m + x * 0.00048828125f;
        }
    }
    if (sum[0] == 123.456f) out[tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

// ---- B: activations in LDS, weights from global memory (L2) into registers, AHEAD k-steps ahead
template <int AHEAD>
__global__ __launch_bounds__(512) void roles_b(const _Float16* w, const _Float16* act, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];     // [128 keypoints][ROWH]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * ROWH / 8; i += 512) reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(act)[i];
    __syncthreads();
    const _Float16* wrow0 = w + (size_t)(wave * 32 + l15) * ROWH + 8 * g;      // row block P; block Q is 16 rows further
    const _Float16* bp = smem + l15 * ROWH + 8 * g;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        f32x4 m[2][8], x[2][8];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < 8; ++t) { m[p][t] = f32x4{0.f, 0.f, 0.f, 0.f}; x[p][t] = m[p][t]; }
        f16x8 ah[2][8], al[2][8];
#pragma unroll
        for (int ks = 0; ks < AHEAD; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ah[p][ks] = *reinterpret_cast<const f16x8*>(wrow0 + p * 16 * ROWH + 32 * ks);
                al[p][ks] = *reinterpret_cast<const f16x8*>(wrow0 + p * 16 * ROWH + 256 + 32 * ks);
            }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + AHEAD < 8) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    ah[p][ks + AHEAD] = *reinterpret_cast<const f16x8*>(wrow0 + p * 16 * ROWH + 32 * (ks + AHEAD));
                    al[p][ks + AHEAD] = *reinterpret_cast<const f16x8*>(wrow0 + p * 16 * ROWH + 256 + 32 * (ks + AHEAD));
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + t * 16 * ROWH + 32 * ks);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + t * 16 * ROWH + 256 + 32 * ks);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    x[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p][ks], bl, x[p][t], 0, 0, 0);
                    m[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p][ks], bh, m[p][t], 0, 0, 0);
                    x[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p][ks], bh, x[p][t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < 8; ++t) sum += m[p][t] + x[p][t] * 0.00048828125f;
    }
    if (sum[0] == 123.456f) out[tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <typename F>
float timed(F&& launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    _Float16 *w, *act; float* out;
    (void)hipMalloc(&w, 256 * ROWH * 2); (void)hipMalloc(&act, 128 * ROWH * 2); (void)hipMalloc(&out, 4096);
    (void)hipMemset(w, 0, 256 * ROWH * 2); (void)hipMemset(act, 0, 128 * ROWH * 2);
    const int reps = 200, blocks = 256;
    const double ideal_us = 3072.0 * 16 / 4 / 2400.0;     // 16 cycles per 16x16x32 MFMA, 4 SIMDs, 2.4 GHz
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(roles_a), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * ROWH * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(roles_b<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * ROWH * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(roles_b<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * ROWH * 2);
    float ms = timed([&] { hipLaunchKernelGGL(roles_a, dim3(blocks), dim3(512), 64 * ROWH * 2, 0, w, out, reps); });
    printf("A weights in LDS, activations in registers:           %.2f us per 256x256x128 tile (MFMA floor %.2f us): %.0f %% of the matrix peak\n", ms * 1e3 / reps, ideal_us, 100 * ideal_us / (ms * 1e3 / reps));
    ms = timed([&] { hipLaunchKernelGGL(roles_b<2>, dim3(blocks), dim3(512), 128 * ROWH * 2, 0, w, act, out, reps); });
    printf("B activations in LDS, weights from L2 (2 k-steps ahead): %.2f us per tile: %.0f %%\n", ms * 1e3 / reps, 100 * ideal_us / (ms * 1e3 / reps));
    ms = timed([&] { hipLaunchKernelGGL(roles_b<4>, dim3(blocks), dim3(512), 128 * ROWH * 2, 0, w, act, out, reps); });
    printf("B activations in LDS, weights from L2 (4 k-steps ahead): %.2f us per tile: %.0f %%\n", ms * 1e3 / reps, 100 * ideal_us / (ms * 1e3 / reps));
    return 0;
}
