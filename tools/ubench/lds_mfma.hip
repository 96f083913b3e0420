// Microbenchmark: what is the ceiling of the layer kernel's inner loop (weight fragments streamed from LDS,
// activations resident in registers, split-f16 = 3 MFMAs per 2 fragment reads)?
//   SHAPE 16: v_mfma_f32_16x16x32_f16, a wave owns 16 keypoints (layer.hip today)
//   SHAPE 32: v_mfma_f32_32x32x16_f16, a wave owns 32 keypoints (each fragment read feeds twice the MFMA cycles)
//   READS  : ds_read_b128 per k-step (2 = hi and lo plane, 1 = hi only (lo reuses it), 0 = none: MFMA-only ceiling)
//   NV     : independent v_fma_f32 issued behind every MFMA (stands for the epilogue steps)
//   THREADS: 512 = two waves per SIMD, 256 = one
// Work per workgroup and rep: a 256 x 256 weight matrix times (THREADS / 64 * SHAPE) keypoints.
//   hipcc --offload-arch=gfx950 -O3 lds_mfma.hip -o lds_mfma.bin && ./lds_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int ROWH = 528;      // halves per LDS row: 256 hi | 256 lo | 16 pad (pitch = 32 B mod 256 B)

template <int NV>
__device__ __forceinline__ void filler(float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 3]) : "v"(v[4 + (j & 3)]), "v"(v[4 + ((j + 1) & 3)]));
}

// ---- SHAPE 16: block = 16 weight rows, 8 k-steps of 32; fragments AHEAD k-steps ahead, across block boundaries
template <int READS, int NV, int THREADS, int AHEAD>
__global__ __launch_bounds__(THREADS) void k16(const _Float16* w, const _Float16* act, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const long long t_start = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 64 * ROWH / 8; i += THREADS) reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(w)[i];
    f16x8 xh[8], xl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        xh[ks] = *reinterpret_cast<const f16x8*>(act + (size_t)(tid * 16 + ks) * 8);
        xl[ks] = *reinterpret_cast<const f16x8*>(act + (size_t)(tid * 16 + 8 + ks) * 8);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)xh[0][j] * 1e-3f;
    __syncthreads();
    const _Float16* wp0 = smem + l15 * ROWH + 8 * g;
    f16x8 fh[8], fl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { fh[ks] = *reinterpret_cast<const f16x8*>(wp0 + 32 * ks); fl[ks] = *reinterpret_cast<const f16x8*>(wp0 + 256 + 32 * ks); }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int nblk = 16 * reps;
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const _Float16* wcur = wp0 + ((blk + b2) & 3) * 16 * ROWH;
            const _Float16* wnext = wp0 + ((blk + b2 + 1) & 3) * 16 * ROWH;
            f32x4 m = {0.f, 0.f, 0.f, 0.f}, x = m;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (READS >= 1) {
                    const int kn = ks + AHEAD;
                    const _Float16* src = kn < 8 ? wcur + 32 * kn : wnext + 32 * (kn - 8);
                    // fragment kn & 7 was consumed AHEAD... steps ago only if AHEAD < 8: read into the slot used 8 - AHEAD steps later
                    fh[kn & 7] = *reinterpret_cast<const f16x8*>(src);
                    if (READS >= 2) fl[kn & 7] = *reinterpret_cast<const f16x8*>(src + 256);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (READS == 0) asm volatile("" : "+v"(fh[ks]));      // (keeps the block from being hoisted out of the loop)
                const f16x8 ph = fh[ks], pl = READS >= 2 ? fl[ks] : fh[ks];
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, xl[ks], x, 0, 0, 0); filler<NV>(v); __builtin_amdgcn_sched_barrier(0);
                m = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, xh[ks], m, 0, 0, 0); filler<NV>(v); __builtin_amdgcn_sched_barrier(0);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, xh[ks], x, 0, 0, 0); filler<NV>(v); __builtin_amdgcn_sched_barrier(0);
            }
            sum += m + x * 0.00048828125f;
        }
    }
    float s = sum[0] + sum[1] + sum[2] + sum[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out)[64] = __builtin_amdgcn_s_memtime() - t_start;
}

// ---- SHAPE 32: block = 32 weight rows, 16 k-steps of 16 (fragment = 32 rows x 16 K: lane (row l31, half hi) reads 8 halves)
template <int READS, int NV, int THREADS, int AHEAD>
__global__ __launch_bounds__(THREADS) void k32(const _Float16* w, const _Float16* act, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 64 * ROWH / 8; i += THREADS) reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(w)[i];
    f16x8 xh[16], xl[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        xh[ks] = *reinterpret_cast<const f16x8*>(act + (size_t)(tid * 32 + ks) * 8);
        xl[ks] = *reinterpret_cast<const f16x8*>(act + (size_t)(tid * 32 + 16 + ks) * 8);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)xh[0][j] * 1e-3f;
    __syncthreads();
    const _Float16* wp0 = smem + l31 * ROWH + 8 * hi;
    f16x8 fh[8], fl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { fh[ks] = *reinterpret_cast<const f16x8*>(wp0 + 16 * ks); fl[ks] = *reinterpret_cast<const f16x8*>(wp0 + 256 + 16 * ks); }
    f32x16 sum = {};
    const int nblk = 8 * reps;
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const _Float16* wcur = wp0 + ((blk + b2) & 1) * 32 * ROWH;
            const _Float16* wnext = wp0 + ((blk + b2 + 1) & 1) * 32 * ROWH;
            f32x16 m = {}, x = {};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (READS >= 1) {
                    const int kn = ks + AHEAD;
                    const _Float16* src = kn < 16 ? wcur + 16 * kn : wnext + 16 * (kn - 16);
                    fh[kn & 7] = *reinterpret_cast<const f16x8*>(src);
                    if (READS >= 2) fl[kn & 7] = *reinterpret_cast<const f16x8*>(src + 256);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (READS == 0) asm volatile("" : "+v"(fh[ks & 7]));
                const f16x8 ph = fh[ks & 7], pl = READS >= 2 ? fl[ks & 7] : fh[ks & 7];
                x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, xl[ks], x, 0, 0, 0); filler<2 * NV>(v); __builtin_amdgcn_sched_barrier(0);
                m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, xh[ks], m, 0, 0, 0); filler<2 * NV>(v); __builtin_amdgcn_sched_barrier(0);
                x = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, xh[ks], x, 0, 0, 0); filler<2 * NV>(v); __builtin_amdgcn_sched_barrier(0);
            }
            sum += m + x * 0.00048828125f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += sum[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[tid] = s;
}

template <typename F>
float timed(F&& launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

_Float16 *w, *act; float* out;
template <int SHAPE, int READS, int NV, int THREADS, int AHEAD>
void run() {
    const int reps = 200, blocks = 256;
    const size_t lds = 100 * 1024;          // more than half of the LDS: one workgroup per CU
    auto kern = SHAPE == 16 ? k16<READS, NV, THREADS, AHEAD> : k32<READS, NV, THREADS, AHEAD>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float ms = timed([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), lds, 0, w, act, out, reps); });
    const double kp = THREADS / 64 * SHAPE;
    const double flop = 2.0 * 3 * 256 * 256 * kp * reps * blocks;
    const double us_tile128 = ms * 1e3 / reps * 128.0 / kp;
    long long ticks = 0;
    (void)hipMemcpy(&ticks, reinterpret_cast<char*>(out) + 512, 8, hipMemcpyDeviceToHost);
    printf("shape %2d reads %d nv %d threads %3d ahead %d: %7.2f us per 256x256x128-keypoint tile, %6.1f TFLOP/s = %4.1f %% of 2.5 PF",
           SHAPE, READS, NV, THREADS, AHEAD, us_tile128, flop / (ms * 1e-3) * 1e-12, flop / (ms * 1e-3) / 2.5e15 * 100);
    if (SHAPE == 16) printf("   s_memtime: %.0f ticks per us, %.1f ticks per MFMA and wave", ticks / (ms * 1e3), (double)ticks / (3.0 * 8 * 16 * reps));
    printf("\n");
}


// ---- SHAPE 16 with the weights streamed L2 -> LDS like layer.hip: ring of NSLOT stages of 16 rows (17 chunks of 1 KB,
//      chunk c copied by wave c & 7 with global_load_lds_dwordx4), copy of stage h + LOOK issued during stage h, one
//      barrier per PAIR (1 or 2) stages.  DUP 1: every wave issues 3 slices (chunk 16 eight times, as layer.hip does).
template <int NV, int LOOK, int NSLOT, int PAIR, int DUP>
__global__ __launch_bounds__(512) void k16s(const _Float16* w, const _Float16* act, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    constexpr int SLOT_BYTES = 17 * 1024, SLOT_HALVES = SLOT_BYTES / 2;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)smem;
    f16x8 xh[8], xl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        xh[ks] = *reinterpret_cast<const f16x8*>(act + (size_t)(tid * 16 + ks) * 8);
        xl[ks] = *reinterpret_cast<const f16x8*>(act + (size_t)(tid * 16 + 8 + ks) * 8);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)xh[0][j] * 1e-3f;
    auto dma_slice = [&](int stage, int i) __attribute__((always_inline)) {
        // stage = running stage number; source block stage & 15
        int c = wave + 8 * i;
        if (DUP) c = min(c, 16); else if (c > 16) return;
        const char* src = reinterpret_cast<const char*>(w) + (size_t)(stage & 15) * 16 * ROWH * 2 + c * 1024;
        const unsigned dst = lds0 + (unsigned)(stage % NSLOT) * SLOT_BYTES + c * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(lane * 16), "s"(src) : "memory");
    };
    for (int h = 0; h < LOOK; ++h) for (int i = 0; i < 3; ++i) dma_slice(h, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int nblk = 16 * reps;
    constexpr int AHEAD = 3;
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int st = blk + b2;
            const _Float16* wp = smem + (st % NSLOT) * SLOT_HALVES + l15 * ROWH + 8 * g;
            f32x4 m = {0.f, 0.f, 0.f, 0.f}, x = m;
            f16x8 fh[8], fl[8];
#pragma unroll
            for (int ks = 0; ks < AHEAD; ++ks) { fh[ks] = *reinterpret_cast<const f16x8*>(wp + 32 * ks); fl[ks] = *reinterpret_cast<const f16x8*>(wp + 256 + 32 * ks); }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + AHEAD < 8) {
                    fh[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wp + 32 * (ks + AHEAD));
                    fl[ks + AHEAD] = *reinterpret_cast<const f16x8*>(wp + 256 + 32 * (ks + AHEAD));
                }
                __builtin_amdgcn_sched_barrier(0);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks], xl[ks], x, 0, 0, 0); filler<NV>(v); __builtin_amdgcn_sched_barrier(0);
                if (ks < 3) dma_slice(st + LOOK, ks);
                m = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks], xh[ks], m, 0, 0, 0); filler<NV>(v); __builtin_amdgcn_sched_barrier(0);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[ks], xh[ks], x, 0, 0, 0); filler<NV>(v); __builtin_amdgcn_sched_barrier(0);
            }
            sum += m + x * 0.00048828125f;
            if (PAIR == 1 || b2 == 1) {
                // stages st + 1 .. st + PAIR must have landed; younger copies may stay in flight
                constexpr int SL = DUP ? 3 : 3;   // slices issued per stage by a wave (wave 0: 3, others 2 without DUP - conservative: count 2)
                constexpr int YOUNG = (LOOK - PAIR) * (DUP ? 3 : 2);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNG < 0 ? 0 : YOUNG) : "memory");
                (void)SL;
                __syncthreads();
            }
        }
    }
    float s = sum[0] + sum[1] + sum[2] + sum[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[tid] = s;
}
template <int NV, int LOOK, int NSLOT, int PAIR, int DUP>
void run_s() {
    const int reps = 200, blocks = 256;
    const size_t lds = NSLOT * 17 * 1024 > 84 * 1024 ? NSLOT * 17 * 1024 : 84 * 1024;
    auto kern = k16s<NV, LOOK, NSLOT, PAIR, DUP>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float ms = timed([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, w, act, out, reps); });
    const double flop = 2.0 * 3 * 256 * 256 * 128 * reps * blocks;
    printf("staged: nv %d look %d nslot %d pair %d dup %d: %7.2f us per 256x256x128-keypoint tile, %4.1f %% of 2.5 PF\n",
           NV, LOOK, NSLOT, PAIR, DUP, ms * 1e3 / reps, flop / (ms * 1e-3) / 2.5e15 * 100);
}

int main() {
    (void)hipMalloc(&w, 256 * ROWH * 2); (void)hipMalloc(&act, 512 * 32 * 8 * 2 + 4096); (void)hipMalloc(&out, 4096);
    {   // random f16 bit patterns of moderate magnitude (zeros would flatter the power-limited clock)
        const size_t nw = 256 * ROWH, na = 512 * 32 * 8 + 2048;
        _Float16* hw = new _Float16[nw > na ? nw : na];
        unsigned s = 12345u;
        for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; hw[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
        (void)hipMemcpy(w, hw, nw * 2, hipMemcpyHostToDevice);
        for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; hw[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
        (void)hipMemcpy(act, hw, na * 2, hipMemcpyHostToDevice);
        delete[] hw;
    }
    run<16, 0, 0, 512, 3>(); run<32, 0, 0, 512, 3>(); run<16, 0, 0, 256, 3>(); run<32, 0, 0, 256, 3>();
    run<16, 1, 0, 512, 3>(); run<16, 2, 0, 512, 3>(); run<16, 2, 0, 256, 3>();
    run<16, 2, 1, 512, 3>(); run<16, 2, 2, 512, 3>(); run<16, 2, 3, 512, 3>();
    run<32, 2, 0, 512, 3>(); run<32, 2, 0, 256, 3>(); run<32, 2, 2, 256, 3>(); run<32, 2, 2, 512, 3>();
    run_s<0, 3, 5, 2, 1>(); run_s<2, 3, 5, 2, 1>(); run_s<0, 3, 5, 2, 0>(); run_s<0, 3, 4, 1, 1>(); run_s<0, 2, 3, 1, 1>();
    run_s<0, 4, 6, 2, 1>(); run_s<0, 6, 8, 2, 1>(); run_s<0, 6, 8, 2, 0>();
    return 0;
}
