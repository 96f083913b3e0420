// The kNN graph helper.
#include "common.hpp"

namespace {

// knn() of mdgat.py:8-15 (+ get_graph_feature's adjacency, 17-32).  One wave per query point: the
// M negative squared distances -|x|^2 + 2 x.s - |s|^2 go to LDS, then k rounds of wave arg-max pick
// the neighbours nearest-first (the order torch.topk returns).
__global__ __launch_bounds__(256) void knn_kernel(int C, int N, int M, int k, const float* __restrict__ x,
                                                  const float* __restrict__ src, int64_t* __restrict__ idx,
                                                  int64_t* __restrict__ adj) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float* d = smem + (size_t)wave * M;
    const float* xp = x + ((size_t)b * N + n) * C;
    float xx = 0.f;
    for (int c = 0; c < C; ++c) xx = fmaf(xp[c], xp[c], xx);
    for (int m = lane; m < M; m += 64) {
        const float* sp = src + ((size_t)b * M + m) * C;
        float dot = 0.f, ss = 0.f;
        for (int c = 0; c < C; ++c) { dot = fmaf(xp[c], sp[c], dot); ss = fmaf(sp[c], sp[c], ss); }
        d[m] = -xx + 2.f * dot - ss;
    }
    if (adj) for (int m = lane; m < M; m += 64) adj[((size_t)b * N + n) * M + m] = 0;
    for (int r = 0; r < k; ++r) {
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int m = lane; m < M; m += 64) {
            const float v = d[m];
            if (v > bv) { bv = v; bi = m; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            idx[((size_t)b * N + n) * k + r] = bi;
            if (adj) adj[((size_t)b * N + n) * M + bi] = 1;
            d[bi] = -__builtin_inff();
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

int launch_knn(int B, int C, int N, int M, int k, const float* x, const float* src, int64_t* idx, int64_t* adj,
               hipStream_t s) {
    if (B <= 0 || N <= 0) return MDGAT_OK;
    if (k <= 0 || k > M) { mdgat_set_error("knn: k=%d out of range for %d source points", k, M); return MDGAT_ERR_BAD_ARG; }
    if ((size_t)M * 4 * sizeof(float) > 64 * 1024) { mdgat_set_error("knn: M=%d too large", M); return MDGAT_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(knn_kernel, dim3((N + 3) / 4, B), dim3(256), (size_t)M * 4 * sizeof(float), s, C, N, M, k, x, src, idx, adj);
    return mdgat_check_hip(hipGetLastError(), "knn launch");
}
