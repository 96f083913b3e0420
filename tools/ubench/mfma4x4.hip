// Dev microbenchmark: v_mfma_f32_4x4x4_16B_f16 - operand / result layout and issue rate (is it a usable shape for rows of
// 2048 logits: 4 queries per wave, a query's logits in 16 lanes x 128 registers?).
// hipcc --offload-arch=gfx950 -O3 -o mfma4x4.bin mfma4x4.hip && ./mfma4x4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// layout probe: A[b][i][k], B[b][k][j] for 16 blocks; lane l supplies a = A operand, b = B operand as 4 halves
__global__ void layout(const float* Ain, const float* Bin, float* D) {
    const int l = threadIdx.x;
    f16x4 a, b;
    for (int k = 0; k < 4; ++k) { a[k] = (_Float16)Ain[l * 4 + k]; b[k] = (_Float16)Bin[l * 4 + k]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

__global__ __launch_bounds__(512, 1) void rate(float* out, long long* cyc, int iters) {
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f16x4 x, y;
    for (int j = 0; j < 4; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x4f16(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x4f16(x, y, c3, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float hA[256], hB[256], hD[256];
    for (int i = 0; i < 256; ++i) { hA[i] = (float)((i * 7) % 13) - 6.f; hB[i] = (float)((i * 5) % 11) - 5.f; }
    float *dA, *dB, *dD; long long* dc;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096); hipMalloc(&dc, 8);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    // hypothesis H1: block b = lane >> 2, i = lane & 3; A operand of lane (b, i) = row i of A_b (over k), B operand = column i of B_b (over k);
    // D register r of lane (b, j) = D_b[r][j]
    int ok1 = 0, ok2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int b = l >> 2, j = l & 3;
        float d1 = 0, d2 = 0;
        for (int k = 0; k < 4; ++k) {
            d1 += hA[(b * 4 + r) * 4 + k] * hB[(b * 4 + j) * 4 + k];       // D_b[r][j] = sum_k A_b[r][k] B_b[k][j]
            d2 += hA[(b * 4 + j) * 4 + k] * hB[(b * 4 + r) * 4 + k];       // transposed hypothesis
        }
        ok1 += fabsf(d1 - hD[l * 4 + r]) < 1e-3f; ok2 += fabsf(d2 - hD[l * 4 + r]) < 1e-3f;
    }
    printf("layout H1 (D reg r of lane (b, j) = sum_k A[b][r][k] B[b][j][k]): %d / 256; H2 (transposed): %d / 256\n", ok1, ok2);
    // other hypothesis: block index = lane & 15, i = lane >> 4
    int ok3 = 0, ok4 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int b = l & 15, j = l >> 4;
        float d1 = 0, d2 = 0;
        for (int k = 0; k < 4; ++k) {
            d1 += hA[((r * 16 + b)) * 4 + k] * hB[((j * 16 + b)) * 4 + k];
            d2 += hA[((j * 16 + b)) * 4 + k] * hB[((r * 16 + b)) * 4 + k];
        }
        ok3 += fabsf(d1 - hD[l * 4 + r]) < 1e-3f; ok4 += fabsf(d2 - hD[l * 4 + r]) < 1e-3f;
    }
    printf("layout H3 (block = lane & 15, i = lane >> 4): %d / 256; H4: %d / 256\n", ok3, ok4);
    float* dout; hipMalloc(&dout, 4096);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(rate, dim3(1), dim3(512), 0, 0, dout, dc, 4000);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("rate: %.2f ticks per 4x4x4 MFMA per wave (2 waves / SIMD, 4 independent accumulators) -> %.2f per SIMD\n", c / (4.0 * 4000), c / (8.0 * 4000));
    return 0;
}
