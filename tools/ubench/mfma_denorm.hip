// Does v_mfma_f32_16x16x32_f16 / 32x32x16_f16 honour f16 denormal operands?  (An unscaled residual plane lo = x - f16(x)
// of operands of magnitude < 0.25 lies in the f16 denormal range.)
//   hipcc --offload-arch=gfx950 -O3 mfma_denorm.hip -o mfma_denorm.bin && ./mfma_denorm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float av, float bv) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)av; b[j] = (_Float16)bv; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    f32x16 d = {};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; }
}
int main() {
    float* o; (void)hipMalloc(&o, 8);
    const float as[] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 3.0517578125e-05f /* 2^-15 */};
    for (float a : as) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, a, 1.0f);
        float h[2]; (void)hipMemcpy(h, o, 8, hipMemcpyDeviceToHost);
        printf("a = %.10g (f16 denormal) x b = 1, K = 32 / 16: 16x16x32 -> %.10g (exact %.10g), 32x32x16 -> %.10g (exact %.10g)\n", a, h[0], 32.0 * a, h[1], 16.0 * a);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 1.0f, a);
        (void)hipMemcpy(h, o, 8, hipMemcpyDeviceToHost);
        printf("   swapped:                                        16x16x32 -> %.10g, 32x32x16 -> %.10g\n", h[0], h[1]);
    }
    return 0;
}
