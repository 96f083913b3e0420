// Device-side helpers shared by the fp64 kernels of the reference-exact mode (f64.hip, layer_f64.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// v_mfma_f64_16x16x4_f64: A 16x4, B 4x16 one double per lane (row / col = lane & 15, k = lane >> 4); C/D col = lane & 15,
// row = (lane >> 4) + 4 reg (cdna_hip_programming.md section 3)
__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// Range guard of the exact mode, BETWEEN the layers: every output of an fp64 product (q | k | v, the hidden layer before its ReLU,
// the residual stream, the encoder stages) and every message row is tested for "not finite, or |v| >= 2^500" by its exponent field,
// and the call is refused (MDGAT_STATUS_RANGE -> mdgat_async_status / MDGAT.check raise) instead of returning plausible numbers.
// Why here: these files are compiled with -fno-honor-nans, and the reference's NaN propagation (mdgat.py:192-193: a NaN logit makes
// the whole row NaN) is not what the hardware does with one - max(NaN, 0) is 0 in a ReLU, v_max_f64 drops a NaN logit from the row
// maximum and exp_neg clamps it to exp(-745) = 0, so a NaN produced mid-stack (inf - inf in a product or in the online softmax)
// could come out as a finite, wrong message.  With every q, k, v below 2^500 a logit is a sum of 32 products below 2^1000: finite; the
// softmax statistics and P.V of finite logits and values are finite.  So the test on the GEMM outputs (before the ReLU, which
// could swallow a NaN, and after the residual is added) is sufficient, and it sits in the epilogues: two integer instructions per
// output element, none in the product loops.  (The asm hides the value's floating-point origin: a mask test the compiler can trace
// back to a double is recognised as a class test and, under the flag, reduced to "is infinite" - a NaN would pass.)
__device__ __forceinline__ bool f64_out_of_range(double v) {
    unsigned hi = (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32);
    asm("" : "+v"(hi));
    return (hi & 0x7ff00000u) >= 0x5f300000u;          // biased exponent >= 1523: |v| >= 2^500, inf, NaN
}
__device__ __forceinline__ void f64_raise(unsigned* guard) { if (guard) __hip_atomic_store(guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
