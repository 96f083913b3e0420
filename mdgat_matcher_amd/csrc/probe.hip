// Measurement only (bench.py, roofline.sustained_*): what the f16 matrix cores of THIS device sustain on non-trivial data.
// The dense peak of the guide (2.5 PFLOP/s) assumes 2.4 GHz; under a dense MFMA load with random operands the chip clocks
// to its power budget instead (measured on the boxes of this pool: ~1.6 GHz, ~1.5 PFLOP/s; zero-filled operands flatter
// the clock by ~20 %).  The probe runs the inner loop of layer.hip stripped to its MFMAs - v_mfma_f32_16x16x32_f16, two
// waves per SIMD, two accumulator chains per wave, operands resident in registers, nothing else - and reports time and
// shader cycles (s_memtime), so that the kernels' MFMA rates can be read against a ceiling that exists.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(const _Float16* src, float* sink, long long* ticks, int reps) {
    const int tid = threadIdx.x;
    f16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)tid * 16 + i) * 8);
        b[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)tid * 16 + 8 + i) * 8);
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        // one "block" of the layer kernel: fresh accumulators, 8 k-steps of three products, combined into a running sum
        // (accumulators that only ever grow toggle fewer bits: the same loop without the reset runs 25 % faster)
        f32x4 m = {0.f, 0.f, 0.f, 0.f}, x = m;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("" : "+v"(a[i]));          // (keeps the loop body from being hoisted: the operands never change)
            x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[i], x, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(i + 1) & 7], m, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + 3) & 7], b[i], x, 0, 0, 0);
        }
        sum += m + x * MDGAT_SPLIT_INV;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const float s = sum[0] + sum[1] + sum[2] + sum[3];
    if (s == 123.456f) sink[tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ void mfma_probe_fill(_Float16* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned s = (unsigned)i * 1664525u + 1013904223u;
        s = s * 1664525u + 1013904223u;
        p[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f);       // values in [-1, 1], all bit patterns busy
    }
}

}  // namespace

extern "C" int mdgat_mfma_probe(int reps, void* workspace, size_t workspace_bytes, float* ms_out, double* flops_out,
                                long long* ticks_out, void* stream) {
    const size_t halves = 512 * 16 * 8, need = halves * 2 + 512 * 4 + 256;
    if (reps <= 0 || !workspace || workspace_bytes < need || !ms_out || !flops_out || !ticks_out) {
        mdgat_set_error("mdgat_mfma_probe: bad argument (workspace >= %zu bytes)", need);
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int dev = 0, num_cu = 0;
    if (int rc = mdgat_check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (int rc = mdgat_check_hip(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev), "CU count")) return rc;
    _Float16* src = static_cast<_Float16*>(workspace);
    float* sink = reinterpret_cast<float*>(src + halves);
    long long* ticks = reinterpret_cast<long long*>(reinterpret_cast<char*>(sink) + 512 * 4 + ((256 - (512 * 4) % 256) % 256));
    hipLaunchKernelGGL(mfma_probe_fill, dim3((halves + 255) / 256), dim3(256), 0, s, src, (int)halves);
    hipEvent_t e0, e1;
    if (int rc = mdgat_check_hip(hipEventCreate(&e0), "event")) return rc;
    if (int rc = mdgat_check_hip(hipEventCreate(&e1), "event")) { (void)hipEventDestroy(e0); return rc; }
    // the clock follows the power drawn over the last milliseconds: 20 launches bring the chip to the state a long run of
    // matrix-bound kernels leaves it in (a single cold launch measures 20-30 % more), then 5 are timed
    constexpr int WARM = 20, TIMED = 5;
    for (int i = 0; i < WARM; ++i) hipLaunchKernelGGL(mfma_probe_kernel, dim3(num_cu), dim3(512), 0, s, src, sink, ticks, reps);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < TIMED; ++i) hipLaunchKernelGGL(mfma_probe_kernel, dim3(num_cu), dim3(512), 0, s, src, sink, ticks, reps);
    (void)hipEventRecord(e1, s);
    int rc = mdgat_check_hip(hipEventSynchronize(e1), "mfma probe");
    float ms = 0.f;
    if (!rc) rc = mdgat_check_hip(hipEventElapsedTime(&ms, e0, e1), "mfma probe time");
    if (!rc) rc = mdgat_check_hip(hipMemcpy(ticks_out, ticks, sizeof(long long), hipMemcpyDeviceToHost), "mfma probe ticks");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    *ms_out = ms / TIMED;
    *flops_out = 2.0 * 16 * 16 * 32 * 24.0 * reps * 8.0 * num_cu;     // 24 MFMAs per wave and rep, 8 waves per workgroup
    return MDGAT_OK;
}
