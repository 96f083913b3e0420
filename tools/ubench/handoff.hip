// Microbenchmark: the per-iteration hand-off of sinkhorn_scaling_kernel in isolation.  Groups of 4 workgroups (512
// threads, one per CU, partners on one XCD like the kernel places them) exchange one 8-byte {tag, value} granule per
// thread and iteration: store own, poll the three partners', sum.  Variants of the memory instructions and of the poll
// loop, with and without the two workgroup barriers an iteration of the kernel has around the exchange.
//   hipcc --offload-arch=gfx950 -O3 handoff.hip -o handoff && ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

enum { NT_SLEEP = 0, NT_NOSLEEP = 1, ATOMIC_SLEEP = 2, ATOMIC_NOSLEEP = 3, NT_TWO_IN_FLIGHT = 4 };

template <int MODE>
__device__ __forceinline__ u64 xload(u64* p) {
    if (MODE == NT_SLEEP || MODE == NT_NOSLEEP || MODE == NT_TWO_IN_FLIGHT) {
        u64 v;
        asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ void xstore(u64* p, u64 v) {
    if (MODE == NT_SLEEP || MODE == NT_NOSLEEP || MODE == NT_TWO_IN_FLIGHT) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE, bool BARRIERS>
__global__ __launch_bounds__(512) void handoff_kernel(u64* slots, int iters, float* out, int work) {
    __shared__ float lds[1024];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int w = q & 3, group = (q >> 2) * 8 + xcd;          // partners share blockIdx % 8
    u64* base = slots + (size_t)group * 2 * 4 * 520;
    float acc = (float)tid;
    for (int it = 1; it <= iters; ++it) {
        // stand-in for the arithmetic of an iteration (dependent FMAs)
        for (int i = 0; i < work; ++i) acc = fmaf(acc, 1.0000001f, 0.5f);
        if (BARRIERS) { lds[tid] = acc; __syncthreads(); acc += lds[(tid + 64) & 511]; }
        u64* b = base + (size_t)(it & 1) * 4 * 520 + tid;
        xstore<MODE>(b + (size_t)w * 520, ((u64)it << 32) | __builtin_bit_cast(unsigned, acc));
        float tot = acc;
        unsigned pending = 0xfu & ~(1u << w);
        while (pending) {
            u64 x[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) if (pending & (1u << p)) x[p] = xload<MODE>(b + (size_t)p * 520);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if ((pending & (1u << p)) && (unsigned)(x[p] >> 32) == (unsigned)it) { tot += __builtin_bit_cast(float, (unsigned)x[p]); pending &= ~(1u << p); }
            if (pending && (MODE == NT_SLEEP || MODE == ATOMIC_SLEEP)) __builtin_amdgcn_s_sleep(1);
        }
        acc = tot * 0.25f;
        if (BARRIERS) { lds[512 + tid] = acc; __syncthreads(); acc = lds[512 + ((tid + 8) & 511)]; }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, bool BARRIERS>
void run(const char* name, u64* slots, float* out, int work) {
    const int iters = 400, blocks = 256;
    void* args[] = {&slots, (void*)&iters, &out, &work};
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(slots, 0, (size_t)64 * 2 * 4 * 520 * 8);
    int warm = 10;
    void* wargs[] = {&slots, &warm, &out, &work};
    hipLaunchCooperativeKernel(reinterpret_cast<void*>(handoff_kernel<MODE, BARRIERS>), dim3(blocks), dim3(512), wargs, 0, 0);
    hipDeviceSynchronize();
    hipMemset(slots, 0, (size_t)64 * 2 * 4 * 520 * 8);
    hipEventRecord(a);
    hipLaunchCooperativeKernel(reinterpret_cast<void*>(handoff_kernel<MODE, BARRIERS>), dim3(blocks), dim3(512), args, 0, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-44s barriers %d work %4d FMAs: %.3f us per iteration\n", name, (int)BARRIERS, work, ms * 1e3 / iters);
}

int main() {
    u64* slots; float* out;
    hipMalloc(&slots, (size_t)64 * 2 * 4 * 520 * 8); hipMalloc(&out, 64);
    for (int work : {0, 300}) {
        run<NT_SLEEP, false>("nt store/load, s_sleep(1) (as shipped)", slots, out, work);
        run<NT_NOSLEEP, false>("nt store/load, no sleep", slots, out, work);
        run<ATOMIC_SLEEP, false>("agent-scope atomics, s_sleep(1)", slots, out, work);
        run<ATOMIC_NOSLEEP, false>("agent-scope atomics, no sleep", slots, out, work);
        run<NT_SLEEP, true>("nt store/load, s_sleep(1) (as shipped)", slots, out, work);
        run<NT_NOSLEEP, true>("nt store/load, no sleep", slots, out, work);
    }
    return 0;
}
