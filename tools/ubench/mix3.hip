// Dev microbenchmark: which vector instructions run in the shadow of MFMAs with two waves per SIMD?
// hipcc --offload-arch=gfx950 -O3 -o mix3.bin mix3.hip && ./mix3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND, int NM>
__global__ __launch_bounds__(512, 1) void k2(float* out, long long* cyc, int iters) {
    f32x16 a0 = {}, a1 = {};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j * 0.01f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (NM) { if (h == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0); else a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int d = j & 7, s1 = 8 + (j & 7), s2 = 8 + ((j + 1) & 7);
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[d]) : "v"(v[s1]), "v"(v[s2]));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(v[d]) : "v"(v[s1]));
                else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(f32x2*)&v[2 * (j & 3)]) : "v"(*(f32x2*)&v[8 + 2 * (j & 3)]), "v"(*(f32x2*)&v[8 + 2 * ((j + 1) & 3)]));
                else if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[d]) : "v"(v[s1]), "v"(v[s2]));
                else if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(v[d]) : "v"(v[s1]), "v"(v[s2]));
                else if (KIND == 5) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(v[d]) : "v"(v[s1]), "v"(v[s2]));
                else if (KIND == 6) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[d]) : "v"(v[s1]), "v"(v[s2]));
                else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(f32x2*)&v[2 * (j & 3)]) : "v"(*(f32x2*)&v[8 + 2 * (j & 3)]), "v"(*(f32x2*)&v[8 + 2 * ((j + 1) & 3)]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + a0[j] + a1[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, int KIND, int NM>
void run2(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k2<NV, KIND, NM>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k2<NV, KIND, NM>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k2<NV, KIND, NM>), dim3(1), dim3(512), 0, 0, out, cyc, iters * 50);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-8s mfma=%d NV=%2d: %6.1f ticks = %6.1f ns per (MFMA + NV ops) per wave; 2 waves/SIMD\n", name, NM, NV, c / (100.0 * iters), ms * 1e6 / (100.0 * iters));
}
template <int KIND>
void runs(const char* name, float* out, long long* cyc) {
    run2<8, KIND, 0>(name, out, cyc); run2<16, KIND, 0>(name, out, cyc);
    run2<4, KIND, 1>(name, out, cyc); run2<8, KIND, 1>(name, out, cyc); run2<16, KIND, 1>(name, out, cyc);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    run2<0, 0, 1>("none", out, cyc);
    runs<0>("fma", out, cyc); runs<6>("add", out, cyc); runs<1>("exp", out, cyc); runs<2>("pk_add", out, cyc); runs<7>("pk_fma", out, cyc);
    runs<3>("cvt_pk", out, cyc); runs<4>("fma_mix", out, cyc); runs<5>("max3", out, cyc);
    return 0;
}
