// Microbenchmark: per-CU global->LDS streaming rate of an L2-resident buffer that every workgroup re-reads
// (the weight stream of layer.hip).  hipcc --offload-arch=gfx950 -O3 l2stream.hip -o l2stream && ./l2stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NLOAD, int THREADS>
__global__ __launch_bounds__(THREADS) void stream_kernel(const float* __restrict__ w, size_t bytes, int reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const size_t nchunk = bytes / 16;                       // 16-byte chunks
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (size_t base = 0; base < nchunk; base += (size_t)THREADS * NLOAD) {
            f32x4 x[NLOAD];
#pragma unroll
            for (int u = 0; u < NLOAD; ++u) x[u] = *reinterpret_cast<const f32x4*>(w + (base + tid + (size_t)u * THREADS) * 4);
#pragma unroll
            for (int u = 0; u < NLOAD; ++u) *reinterpret_cast<f32x4*>(lds + ((tid + u * THREADS) & 2047) * 4) = x[u];
            __syncthreads();
            acc += lds[(tid * 7) & 8191];
            __syncthreads();
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int NLOAD, int THREADS>
void run(const float* w, size_t bytes, float* out, int blocks) {
    const int reps = 20;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((stream_kernel<NLOAD, THREADS>), dim3(blocks), dim3(THREADS), 32768, 0, w, bytes, 2, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((stream_kernel<NLOAD, THREADS>), dim3(blocks), dim3(THREADS), 32768, 0, w, bytes, reps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double total = (double)bytes * reps * blocks;
    printf("threads %4d loads/thread %2d blocks %4d: %.3f ms  %.2f TB/s aggregate  %.1f GB/s per block\n", THREADS, NLOAD, blocks, ms,
           total / ms / 1e9, total / blocks / ms / 1e6);
}

int main() {
    const size_t bytes = 576 * 1024;
    float *w, *out;
    hipMalloc(&w, bytes); hipMalloc(&out, 64);
    hipMemset(w, 0, bytes);
    for (int blocks : {256, 512}) {
        run<4, 256>(w, bytes, out, blocks);
        run<8, 256>(w, bytes, out, blocks);
        run<16, 256>(w, bytes, out, blocks);
        run<8, 512>(w, bytes, out, blocks);
        run<8, 1024>(w, bytes, out, blocks);
    }
    return 0;
}
