// Both encoders and their sum (mdgat.py:392-393) in one kernel, 128 keypoints per workgroup:
//
//   KeypointEncoder   (mdgat.py:176-188)  [x, y, z, saliency] -> 32 -> 64 -> 128 -> (128)      conv + folded BN + ReLU
//   DescriptorEncoder (mdgat.py:144-155)  33-D FPFH           -> 64 -> 128 -> (128)
//   x = W [denc hidden ; kenc hidden] + b                       the two last convs summed = one 256 -> 128 product
//
// Optionally the inputs are the loader's raw frame records (load_data.py:152-165: 37 float32 per keypoint =
// xyz | saliency | FPFH) and the FPFH part is L2-normalised on the fly (load_data.py:290-292).
//
// Same register-resident chain as layer.hip (mma_chain.hpp): a wave owns 32 keypoints, every product is
// computed swapped on the f16 matrix cores with split operands, the ReLU output of one layer, split in place,
// is the B operand of the next.  The 4 -> 32 conv is plain VALU (K = 4); the 33 -> 64 conv pads K to 48.
// All weights but the last product's sit in LDS at once (92 KB); the 256 -> 128 product streams its four
// 32-row blocks through two 33 KB buffers.
#include "common.hpp"
#include "mma_chain.hpp"

namespace {

struct EncArgs {
    // inputs: separate arrays per frame, or raw records (rec0/rec1 != NULL)
    const float *kpts0, *sigma0, *fpfh0, *kpts1, *sigma1, *fpfh1;
    const float *rec0, *rec1;     // [B][N][37], [B][M][37]
    int normalize;                // records: L2-normalise the FPFH part
    const float* w;               // fp32 blob (kenc0 weights / biases live there)
    size_t kenc0_w, kenc0_b, denc0_b, kenc1_b, kenc2_b, denc1_b, encl_b;
    const _Float16 *k1s, *k2s, *d0s, *d1s, *els;   // split weights [rows][2][K]: 64x32, 128x64, 64x48, 128x64, 128x256
    float* x;                     // [R][128]
    int B, N, M, R;
};

// LDS map (halves)
constexpr int RH32 = 72, RH48 = 104, RH64 = 136, RH256 = 520;
constexpr int OFF_K1 = 0;                          // 64 rows x RH32
constexpr int OFF_K2 = OFF_K1 + 64 * RH32;         // 128 x RH64
constexpr int OFF_D0 = OFF_K2 + 128 * RH64;        // 64 x RH48
constexpr int OFF_D1 = OFF_D0 + 64 * RH48;         // 128 x RH64
constexpr int OFF_EL = OFF_D1 + 128 * RH64;        // 2 x 32 x RH256
constexpr int OFF_END = OFF_EL + 2 * 32 * RH256;   // then fp32: kenc0 w [32][4], b [32], biases

// Eight waves = 256 keypoints per workgroup, one workgroup per CU (the weights fill the LDS): two waves per SIMD, so that one
// wave's matrix instructions run beside the other's loads, splits and 32-byte row stores (four waves: 64 -> see DESIGN.md)
constexpr int ENC_THREADS = 512;
// copy ROWS rows of 2K halves (contiguous in memory) into padded LDS rows, in two steps: every 16-byte piece of a thread is
// REQUESTED first (copy_issue) and stored once all of them are on their way (copy_commit).  As one loop with a run-time trip
// count - as until round 4 - the compiler emitted load / wait / store per iteration: the five weight matrices arrived in ~19
// consecutive L2 round trips at the top of every launch.
template <int K, int ROWS> struct CopyRows {
    static constexpr int CPR = 2 * K * 2 / 16, ROWH = 2 * K + 8, N = ROWS * CPR, IT = (N + ENC_THREADS - 1) / ENC_THREADS;
    f32x4 x[IT];
    __device__ __forceinline__ void issue(const _Float16* g, int tid) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = tid + i * ENC_THREADS;
            if (N % ENC_THREADS == 0 || c < N) x[i] = *reinterpret_cast<const f32x4*>(g + (size_t)c * 8);
        }
    }
    __device__ __forceinline__ void commit(_Float16* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = tid + i * ENC_THREADS;
            if (N % ENC_THREADS == 0 || c < N) *reinterpret_cast<f32x4*>(dst + (c / CPR) * ROWH + (c % CPR) * 8) = x[i];
        }
    }
};

__global__ __launch_bounds__(ENC_THREADS, 2) void encoder_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wrow = perm32(l31);
    const int pt = min(blockIdx.x * (ENC_THREADS / 2) + wave * 32 + l31, a.R - 1);
    float* fl = reinterpret_cast<float*>(smem + OFF_END);
    float* k0w = fl;            // [32][4]
    float* k0b = fl + 128;      // [32]
    float* bk1 = fl + 160;      // [64]
    float* bk2 = fl + 224;      // [128]
    float* bd0 = fl + 352;      // [64]
    float* bd1 = fl + 416;      // [128]
    float* bel = fl + 544;      // [128]

    // ---- stage every weight but the last product's, the first two row blocks of the last one, the biases ----
    {
        CopyRows<32, 64> c1; CopyRows<64, 128> c2; CopyRows<48, 64> c3; CopyRows<64, 128> c4; CopyRows<256, 64> c5;
        c1.issue(a.k1s, tid); c2.issue(a.k2s, tid); c3.issue(a.d0s, tid); c4.issue(a.d1s, tid); c5.issue(a.els, tid);
        __builtin_amdgcn_sched_barrier(0);
        c1.commit(smem + OFF_K1, tid); c2.commit(smem + OFF_K2, tid); c3.commit(smem + OFF_D0, tid); c4.commit(smem + OFF_D1, tid);
        c5.commit(smem + OFF_EL, tid);
    }
    if (tid < 128) k0w[tid] = a.w[a.kenc0_w + tid];
    if (tid < 32) k0b[tid] = a.w[a.kenc0_b + tid];
    if (tid < 64) { bk1[tid] = a.w[a.kenc1_b + tid]; bd0[tid] = a.w[a.denc0_b + tid]; }
    if (tid < 128) { bk2[tid] = a.w[a.kenc2_b + tid]; bd1[tid] = a.w[a.denc1_b + tid]; bel[tid] = a.w[a.encl_b + tid]; }

    // ---- this lane's keypoint: (pair, frame, index) and its inputs ----
    const int P = a.N + a.M;
    const int b = pt / P, p = pt - b * P;
    const bool f1 = p >= a.N;
    const int n = f1 ? p - a.N : p, nside = f1 ? a.M : a.N;
    const size_t pi = (size_t)b * nside + n;
    float xyzs[4];
    float f[24];                 // FPFH entries 16 ks + 8 hi + j of this lane, ks = 0..2 (zero beyond 33)
    {
        const float* rec = f1 ? a.rec1 : a.rec0;
        const float* fp;
        if (rec) {
            const float* r = rec + pi * 37;
            xyzs[0] = r[0]; xyzs[1] = r[1]; xyzs[2] = r[2]; xyzs[3] = r[3];
            fp = r + 4;
        } else {
            const float* kp = (f1 ? a.kpts1 : a.kpts0) + pi * 3;
            xyzs[0] = kp[0]; xyzs[1] = kp[1]; xyzs[2] = kp[2];
            xyzs[3] = (f1 ? a.sigma1 : a.sigma0)[pi];
            fp = (f1 ? a.fpfh1 : a.fpfh0) + pi * MDGAT_FPFH;
        }
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * ks + 8 * hi + j;
                const float v = c < MDGAT_FPFH ? fp[c] : 0.f;
                f[8 * ks + j] = v;
                ss = fmaf(v, v, ss);
            }
        if (rec && a.normalize) {      // load_data.py:290-292: desc * (1 / ||desc||), float32 like numpy
            ss += __shfl_xor(ss, 32, 64);
            const float inv = 1.0f / sqrtf(ss);
#pragma unroll
            for (int i = 0; i < 24; ++i) f[i] *= inv;
        }
    }
    __syncthreads();

    // ---- keypoint encoder layer 0 (4 -> 32, VALU): channels 16 t + 8 hi + j -> k-steps t of the next product ----
    f16x8 k0h[2], k0l[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * t + 8 * hi + j;
            const f32x4 wv = *reinterpret_cast<const f32x4*>(k0w + c * 4);
            float acc = k0b[c];
            acc = fmaf(wv[0], xyzs[0], acc);
            acc = fmaf(wv[1], xyzs[1], acc);
            acc = fmaf(wv[2], xyzs[2], acc);
            acc = fmaf(wv[3], xyzs[3], acc);
            v[j] = acc < 0.f ? 0.f : acc;       // (ReLU that lets NaN through: max(NaN, 0) = 0 would hide an overflow from the range guard)
        }
        split8s(v, k0h[t], k0l[t]);
    }
    // relu(out + bias) of a 32-channel block, split into the two k-step fragments it feeds
    auto relu_split = [&](const f32x16& o, const float* bias, f16x8* dh, f16x8* dl) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float bv[8], v[8];
            load8(bias + 16 * t + 8 * hi, bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float y = o[8 * t + j] + bv[j]; v[j] = y < 0.f ? 0.f : y; }     // (NaN passes)
            split8s(v, dh[t], dl[t]);
        }
    };

    // ---- keypoint encoder layers 1 (32 -> 64) and 2 (64 -> 128) ----
    f16x8 k1h[4], k1l[4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        f32x16 o;
        block_mma<2, true>(smem + OFF_K1 + rb * 32 * RH32, wrow, hi, k0h, k0l, o);
        relu_split(o, bk1 + rb * 32, k1h + 2 * rb, k1l + 2 * rb);
    }
    f16x8 hh[16], hl[16];        // [descriptor hidden (k-steps 0..7) ; keypoint hidden (8..15)] of the last product
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        f32x16 o;
        block_mma<4, true>(smem + OFF_K2 + rb * 32 * RH64, wrow, hi, k1h, k1l, o);
        relu_split(o, bk2 + rb * 32, hh + 8 + 2 * rb, hl + 8 + 2 * rb);
    }
    // ---- descriptor encoder layers 0 (33 -> 64, K padded to 48) and 1 (64 -> 128) ----
    f16x8 fh[3], fl16[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[8 * ks + j];
        split8s(v, fh[ks], fl16[ks]);
    }
    f16x8 d0h[4], d0l[4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        f32x16 o;
        block_mma<3, true>(smem + OFF_D0 + rb * 32 * RH48, wrow, hi, fh, fl16, o);
        relu_split(o, bd0 + rb * 32, d0h + 2 * rb, d0l + 2 * rb);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        f32x16 o;
        block_mma<4, true>(smem + OFF_D1 + rb * 32 * RH64, wrow, hi, d0h, d0l, o);
        relu_split(o, bd1 + rb * 32, hh + 2 * rb, hl + 2 * rb);
    }

    // ---- x = W [denc hidden ; kenc hidden] + b: four 32-row blocks, the last two staged behind the first two ----
    float* xrow = a.x + (size_t)pt * 128;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        if (ob == 2) {
            __syncthreads();                 // blocks 0 and 1 consumed by every wave
            CopyRows<256, 64> c6;
            c6.issue(a.els + (size_t)64 * 512, tid);
            c6.commit(smem + OFF_EL, tid);
            __syncthreads();
        }
        f32x16 o;
        block_mma<16, true>(smem + OFF_EL + (ob & 1) * 32 * RH256, wrow, hi, hh, hl, o);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float bv[8], v[8];
            const int ch = ob * 32 + 16 * t + 8 * hi;
            load8(bel + ch, bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = o[8 * t + j] + bv[j];
            store8(xrow + ch, v);            // lanes past the end rewrite the last keypoint's values
        }
    }
}

// fp32 [rows][Kin] -> split [rows][2][Kpad] with zero padding of the K columns
__global__ __launch_bounds__(256) void split_rows_pad_kernel(const float* w, _Float16* out, int rows, int Kin, int Kpad) {
    const size_t total = (size_t)rows * Kpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / Kpad, c = i - r * Kpad;
        _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
        if ((int)c < Kin) mdgat_split(w[r * Kin + c], h, l);
        out[r * 2 * Kpad + c] = h;
        out[r * 2 * Kpad + Kpad + c] = l;
    }
}

}  // namespace

int launch_split_rows_pad(const float* w, _Float16* out, int rows, int Kin, int Kpad, hipStream_t s) {
    const size_t total = (size_t)rows * Kpad;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(split_rows_pad_kernel, dim3(blocks), dim3(256), 0, s, w, out, rows, Kin, Kpad);
    return mdgat_check_hip(hipGetLastError(), "split_rows_pad launch");
}

int launch_encoder(const EncoderLaunch& p, hipStream_t s) {
    if (p.B <= 0) return MDGAT_OK;
    EncArgs a{};
    a.kpts0 = p.kpts0; a.sigma0 = p.sigma0; a.fpfh0 = p.fpfh0; a.kpts1 = p.kpts1; a.sigma1 = p.sigma1; a.fpfh1 = p.fpfh1;
    a.rec0 = p.rec0; a.rec1 = p.rec1; a.normalize = p.normalize;
    a.w = p.w;
    a.kenc0_w = p.bl->kenc0_w; a.kenc0_b = p.bl->kenc0_b; a.denc0_b = p.bl->denc0_b; a.kenc1_b = p.bl->kenc1_b;
    a.kenc2_b = p.bl->kenc2_b; a.denc1_b = p.bl->denc1_b; a.encl_b = p.bl->encl_b;
    a.k1s = p.es; a.k2s = a.k1s + 64 * 64; a.d0s = a.k2s + 128 * 128; a.d1s = a.d0s + 64 * 96; a.els = a.d1s + 128 * 128;
    a.x = p.x; a.B = p.B; a.N = p.N; a.M = p.M; a.R = p.B * (p.N + p.M);
    const size_t lds = (size_t)OFF_END * sizeof(_Float16) + 672 * sizeof(float);
    static std::atomic<unsigned long long> optin;
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(encoder_kernel), lds, optin, "encoder LDS attribute")) return rc;
    hipLaunchKernelGGL(encoder_kernel, dim3((a.R + ENC_THREADS / 2 - 1) / (ENC_THREADS / 2)), dim3(ENC_THREADS), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "encoder launch");
}
