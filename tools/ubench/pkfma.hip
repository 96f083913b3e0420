// Microbenchmark: issue rate of v_pk_fma_f32 against v_fma_f32 (two waves per SIMD, every CU busy, no other work).
//   hipcc --offload-arch=gfx950 -O3 pkfma.hip -o pkfma.bin && ./pkfma.bin
// Work per thread and rep: 128 fused multiply-adds on 128 independent accumulators, as 128 v_fma_f32 or 64 v_pk_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(512) void k(float* out, int reps, float x) {
    f32x2 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x2{(float)i, (float)threadIdx.x};
    const f32x2 m = {x, x * 0.5f};
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(m));
            else {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i][0]) : "v"(m[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i][1]) : "v"(m[1]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][1];
    if (s == 123.456f) out[threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 2000;
    for (int pk = 0; pk < 2; ++pk)
        for (int blocks : {256, 512}) {
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(a);
                if (pk) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(512), 0, 0, out, reps, 0.999f);
                else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(512), 0, 0, out, reps, 0.999f);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            const double fma = (double)blocks * 512 * 128 * reps;
            printf("%s blocks=%d: %.3f ms, %.1f TFLOP/s fp32, %.2f ns per 128 FMAs per thread\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", blocks, ms,
                   2 * fma / ms * 1e-9, ms * 1e6 / reps / (blocks / 256.0));
        }
    return 0;
}
