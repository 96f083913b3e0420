// Microbenchmark: issue time of the vector instructions the softmax / top-k code is made of (two waves per SIMD, every CU busy).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates.bin && ./valu_rates.bin
// Each kernel repeats one instruction 128 times per loop trip on independent registers; the figure is nanoseconds per
// instruction per wave with two waves sharing a SIMD (so 1 / (2 x that) is the SIMD's issue rate for the instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL(NAME, ASM, OPS)                                                                  \
    __global__ __launch_bounds__(512) void NAME(float* out, int reps, float x) {                \
        f32x2 acc[64];                                                                          \
        _Pragma("unroll") for (int i = 0; i < 64; ++i) acc[i] = f32x2{x + i, x * threadIdx.x};   \
        const f32x2 m = {x, x * 0.5f};                                                          \
        _Pragma("unroll 1") for (int r = 0; r < reps; ++r) {                                     \
            _Pragma("unroll") for (int i = 0; i < 64; ++i) { OPS }                               \
        }                                                                                       \
        float s = 0.f;                                                                          \
        _Pragma("unroll") for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][1];               \
        if (s == 123.456f) out[threadIdx.x] = s;                                                \
    }
#define TWO(ASM) asm volatile(ASM : "+v"(acc[i][0]) : "v"(m[0])); asm volatile(ASM : "+v"(acc[i][1]) : "v"(m[1]));
KERNEL(k_fma, "", TWO("v_fma_f32 %0, %0, %1, %1"))
KERNEL(k_add, "", TWO("v_add_f32 %0, %0, %1"))
KERNEL(k_max, "", TWO("v_max_f32 %0, %0, %1"))
KERNEL(k_max3, "", TWO("v_max3_f32 %0, %0, %1, %1"))
KERNEL(k_exp, "", TWO("v_exp_f32 %0, %1"))
KERNEL(k_log, "", TWO("v_log_f32 %0, %1"))
KERNEL(k_rcp, "", TWO("v_rcp_f32 %0, %1"))
KERNEL(k_cvt, "", TWO("v_cvt_f16_f32 %0, %1"))
KERNEL(k_cvtpk, "", TWO("v_cvt_pk_f16_f32 %0, %0, %1"))
KERNEL(k_cvtback, "", TWO("v_cvt_f32_f16 %0, %1"))
KERNEL(k_fmamix, "", TWO("v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[0,1,0]"))
KERNEL(k_cndmask, "", TWO("v_cndmask_b32 %0, %0, %1, vcc"))
KERNEL(k_cmp, "", asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(acc[i][0]), "v"(m[0]) : "vcc"); asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(acc[i][1]), "v"(m[1]) : "vcc");)
KERNEL(k_cndmask64, "", asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(acc[i][0]) : "v"(m[0]) : "s20", "s21"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(acc[i][1]) : "v"(m[1]) : "s20", "s21");)
KERNEL(k_cndmask_mix, "", asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i][0]) : "v"(m[0])); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i][1]) : "v"(m[1]));)
KERNEL(k_bfi, "", TWO("v_bfi_b32 %0, %1, %0, %1"))
KERNEL(k_med3, "", TWO("v_med3_f32 %0, %0, %1, %1"))
KERNEL(k_and, "", TWO("v_and_b32 %0, %0, %1"))
KERNEL(k_readlane, "", asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 5" :: "v"(acc[i][0]), "v"(acc[i][1]) : "s20", "s21");)
KERNEL(k_alignbit, "", TWO("v_alignbit_b32 %0, %0, %1, 31"))
KERNEL(k_mov_dpp, "", TWO("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL(k_pkfma, "", asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(m)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(m));)
KERNEL(k_pkadd, "", asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m));)
KERNEL(k_pkmul, "", asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m));)
KERNEL(k_pkfma_clamp, "", asm volatile("v_pk_fma_f32 %0, %0, %1, %1 clamp" : "+v"(acc[i]) : "v"(m)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1 clamp" : "+v"(acc[i]) : "v"(m));)
KERNEL(k_pkadd_f16, "", TWO("v_pk_add_f16 %0, %0, %1"))
KERNEL(k_pkfma_f16, "", TWO("v_pk_fma_f16 %0, %0, %1, %1"))
KERNEL(k_dot2, "", TWO("v_dot2_f32_f16 %0, %1, %1, %0"))
KERNEL(k_exp_f16, "", TWO("v_exp_f16 %0, %1"))
typedef void (*kern_t)(float*, int, float);
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int reps = 1000;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_max_f32", k_max}, {"v_max3_f32", k_max3}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log},
        {"v_rcp_f32", k_rcp}, {"v_cvt_f16_f32", k_cvt}, {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_cvt_f32_f16", k_cvtback}, {"v_fma_mix_f32", k_fmamix},
        {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32_e64 sgpr", k_cndmask64}, {"cndmask + fma pairs", k_cndmask_mix}, {"v_bfi_b32", k_bfi}, {"v_med3_f32", k_med3}, {"v_and_b32", k_and}, {"v_readlane_b32", k_readlane}, {"v_cmp_ge_f32", k_cmp}, {"v_alignbit_b32", k_alignbit}, {"v_mov_b32_dpp", k_mov_dpp},
        {"v_pk_fma_f32", k_pkfma}, {"v_pk_add_f32", k_pkadd}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32 clamp", k_pkfma_clamp},
        {"v_pk_add_f16", k_pkadd_f16}, {"v_pk_fma_f16", k_pkfma_f16}, {"v_dot2_f32_f16", k_dot2}, {"v_exp_f16", k_exp_f16}};
    for (auto& e : ks) {
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(512), 0, 0, out, reps, 0.999f);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        printf("%-22s %.3f ms  -> %.2f ns per instruction per wave (two waves per SIMD)\n", e.name, ms, ms * 1e6 / reps / 128);
    }
    return 0;
}
