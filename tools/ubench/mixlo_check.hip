// Check: v_fma_mixlo_f16 / v_fma_mixhi_f16 against v_fma_mix_f32 + conversion (residual of an f16 split), and the
// clamp of v_pk_fma_f32 (the ">= threshold" indicator of attention.hip).
//   hipcc --offload-arch=gfx950 -O3 mixlo_check.hip -o mixlo_check.bin && ./mixlo_check.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* p, unsigned* out_old, unsigned* out_new, float* ind, float t) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    const float a = p[2 * i], b = p[2 * i + 1];
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h = {(_Float16)a, (_Float16)b};
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a), "v"(hp));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(b), "v"(hp));
    h2 l = {(_Float16)r0, (_Float16)r1};
    out_old[i] = __builtin_bit_cast(unsigned, l);
    unsigned lp;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(a), "v"(hp));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(b), "v"(hp));
    out_new[i] = lp;
    f32x2 d;
    const int tb = __builtin_bit_cast(int, t);
    const int bp = t > 0.f ? tb - 1 : (tb | (int)0x80000000) + 1;
    const float tp = fminf(__builtin_bit_cast(float, bp), t - 0x1p-90f);
    const float c = -tp * 1.2676506002282294e30f;
    const f32x2 s = {a, b}, c2 = {c, c}, big = {1.2676506002282294e30f, 1.2676506002282294e30f};
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(s), "v"(big), "v"(c2));
    ind[2 * i] = d[0]; ind[2 * i + 1] = d[1];
}
int main() {
    const int n = 1 << 16;
    float* hp = new float[2 * n];
    unsigned s = 1u;
    const float t = 3.14159f;
    for (int i = 0; i < 2 * n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (s >> 8) * (1.0f / 16777216.0f);
        hp[i] = (i % 3 == 0) ? 2048.0f * exp2f(-20.0f * u) : (i % 3 == 1 ? t + (u - 0.5f) * 1e-5f : (u - 0.5f) * 20.0f);
        if (i % 1001 == 0) hp[i] = t;
        if (i % 1003 == 0) hp[i] = nextafterf(t, 0.f);
        if (i % 1007 == 0) hp[i] = -INFINITY;
    }
    float *dp, *dind; unsigned *d0, *d1;
    hipMalloc(&dp, 2 * n * 4); hipMalloc(&dind, 2 * n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
    hipMemcpy(dp, hp, 2 * n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, d0, d1, dind, t);
    unsigned* h0 = new unsigned[n]; unsigned* h1 = new unsigned[n]; float* hi = new float[2 * n];
    hipMemcpy(h0, d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hi, dind, 2 * n * 4, hipMemcpyDeviceToHost);
    int bad = 0, badi = 0;
    for (int i = 0; i < n; ++i) if (h0[i] != h1[i]) { if (bad < 5) printf("split %d: p %.9g %.9g old %08x new %08x\n", i, hp[2 * i], hp[2 * i + 1], h0[i], h1[i]); ++bad; }
    for (int i = 0; i < 2 * n; ++i) { const float want = hp[i] >= t ? 1.f : 0.f; if (hi[i] != want) { if (badi < 5) printf("ind %d: s %.9g got %g want %g\n", i, hp[i], hi[i], want); ++badi; } }
    printf("split mismatches %d of %d, indicator mismatches %d of %d\n", bad, n, badi, 2 * n);
    return 0;
}
