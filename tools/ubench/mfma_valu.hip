// Dev microbenchmark: can one wave's VALU instructions issue in the shadow of its own MFMAs?
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu.bin mfma_valu.hip && ./mfma_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef float f32x4v __attribute__((ext_vector_type(4)));
// two waves per SIMD (512 threads), 16x16x32 MFMAs (KIND16) or 32x32x16
template <int NV, int K16>
__global__ __launch_bounds__(512, 1) void k2(float* out, long long* cyc, int iters) {
    f32x16 a0 = {}, a1 = {};
    f32x4v b0 = {}, b1 = {};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x + j;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (K16) b0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, b0, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 15] = fmaf(v[j & 15], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
        if (K16) b1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, b1, 0, 0, 0);
        else a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 15] = fmaf(v[j & 15], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + a0[j] + a1[j];
    for (int j = 0; j < 4; ++j) s += b0[j] + b1[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, int K16>
void run2(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k2<NV, K16>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k2<NV, K16>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s NV=%d: %.1f ticks per (MFMA + %d VALU) per wave, 2 waves/SIMD\n", name, NV, c / (2.0 * iters), NV);
}

template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    f32x16 a0 = {}, a1 = {};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x + j;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) v[j & 15] = fmaf(v[j & 15], 1.0001f, 0.5f);                       // independent chains (16)
            else if (KIND == 1) v[0] = fmaf(v[0], 1.0001f, 0.5f);                           // one dependent chain
            else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[j & 15]) : "v"(v[(j + 1) & 15]), "v"(v[(j + 2) & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) v[j & 15] = fmaf(v[j & 15], 1.0001f, 0.5f);
            else if (KIND == 1) v[0] = fmaf(v[0], 1.0001f, 0.5f);
            else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[j & 15]) : "v"(v[(j + 1) & 15]), "v"(v[(j + 2) & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + a0[j] + a1[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int KIND>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s NV=%d: %.1f ticks per (MFMA + %d VALU)\n", name, NV, c / (2.0 * iters), NV);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    run<0, 0>("indep", out, cyc); run<4, 0>("indep", out, cyc); run<8, 0>("indep", out, cyc); run<12, 0>("indep", out, cyc); run<16, 0>("indep", out, cyc);
    run<4, 1>("chain", out, cyc); run<8, 1>("chain", out, cyc);
    run<4, 2>("cvtpk", out, cyc); run<8, 2>("cvtpk", out, cyc);
    run2<0, 0>("32x32x16", out, cyc); run2<4, 0>("32x32x16", out, cyc); run2<8, 0>("32x32x16", out, cyc); run2<16, 0>("32x32x16", out, cyc);
    run2<0, 1>("16x16x32", out, cyc); run2<2, 1>("16x16x32", out, cyc); run2<4, 1>("16x16x32", out, cyc); run2<8, 1>("16x16x32", out, cyc);
    return 0;
}
