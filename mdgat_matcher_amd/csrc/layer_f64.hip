// Reference-exact mode: the TAIL of a propagation layer in fp64 as one launch -
//     hid  = ReLU(W1 [x ; msg] + b1)            mlp.0 + folded BN + ReLU   (mdgat.py:246-248; merge folded into W1 by pack.py)
//     x   += W2 hid + b2                        mlp.3 + residual           (mdgat.py:274)
//     qkv  = W3 x + b3                          the NEXT layer's q | k | v (mdgat.py:227-232)
// - what csrc/layer.hip does for the fp32-class path.  The three-launch form (three gemm_f64_kernel launches, f64.hip) writes the
// hidden activation to memory and reads it back (65 536 x 256 x 8 B = 134 MB each way per layer at batch 64), pays a launch's fixed
// 12-15 us three times per layer (start, first fetch, tail: profiles/NOTES_r5.md section 7), and at one pair per call (test.py:132)
// is three dependent launches of a few dozen workgroups each.
//
// gfx950 mapping.  A workgroup of eight waves owns TM = 16 NRB keypoints.  The waves split the OUTPUT CHANNELS of each product
// (16-channel blocks: 2 + 1 + 3 per wave) and every wave multiplies all NRB row blocks, so
//   * a weight is read ONCE per workgroup, straight from L2 into registers: the matrices are kept a second time in FRAGMENT order
//     ([channel block][pair of k-steps][lane][2] doubles - launch_frag64), a wave's load instruction is one contiguous KB, and no
//     weight passes through LDS (1.18 MB per workgroup and layer; the L2s deliver it at a fifth of their rate at batch 64);
//   * the activations every wave needs - the input tile [x ; msg], then the hidden layer, then the new x - take turns in ONE LDS
//     buffer of TM x 258 doubles (66 KB at TM = 32: two workgroups per CU): the tile is dead when the hidden layer is complete, the
//     hidden layer when the new x is; five barriers per workgroup, none inside a product loop;
//   * v_mfma_f64_16x16x4_f64 with the roles of gemm_f64_kernel (A = activations, B = weights), every accumulator walked through k
//     in the same order from zero, bias / ReLU / residual applied in the same order: the results are BIT-IDENTICAL to the
//     three-launch form (tests/test_gpu_f64.py::test_f64_fused_layer_tail_equals_three_launches), which stays for shapes the
//     kernel does not cover and as the reference of that test (mdgat_set_f64_layer_fusion).
#include "common.hpp"
#include "f64.hpp"
#include "f64_dev.hpp"

namespace {

constexpr int LF_LD = 258;          // row pitch of the LDS tile, doubles: 516 dwords = 4 mod 64 - the 32 lanes of a half-wave fragment read
                                    // (rows l15, k-slots g = 0, 1) fall on 32 distinct bank pairs
constexpr int LF_WAVES = 8;

// one product: acc[rb][c] += A[rows of block rb][k] W[channel block cb0 + c][k] over K = 8 JP, W in fragment order from L2, A in LDS.
// PF: pairs of k-steps the weight loads run ahead (the loads of pair jp + PF are issued into the registers pair jp has just been
// multiplied from).
// the first PF pairs of a product's weights.  They do not depend on the activations: a caller may request them BEFORE the barrier
// that makes the activations visible (the one-pair tile height does: a workgroup alone on its CU has nothing else to cover that
// round trip to L2 with).
template <int NCB, int JP, int PF>
__device__ __forceinline__ void lf_prefetch(const double* wf, int lane, f64x2 (&wb)[PF][NCB]) {
    const f64x2* wp = reinterpret_cast<const f64x2*>(wf) + lane;          // channel block c, pair jp: wp[(c * JP + jp) * 64]
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int c = 0; c < NCB; ++c) wb[p][c] = wp[(size_t)(c * JP + p) * 64];
}
template <int NRB, int NCB, int JP, int PF>
__device__ __forceinline__ void lf_product(const double* As, const double* wf, int lane, f64x4 (&acc)[NRB][NCB], f64x2 (&wb)[PF][NCB]) {
    static_assert(JP % PF == 0, "prefetch depth");
    const int l15 = lane & 15, g = lane >> 4;
    const double* ap = As + l15 * LF_LD + g;
    const f64x2* wp = reinterpret_cast<const f64x2*>(wf) + lane;
    double a[2][NRB][2];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) { a[0][rb][0] = ap[rb * 16 * LF_LD]; a[0][rb][1] = ap[rb * 16 * LF_LD + 4]; }
#pragma unroll 1
    for (int jp0 = 0; jp0 < JP; jp0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int jp = jp0 + p;
            // the activations of the next pair of k-steps travel from LDS under this pair's products
            if (jp + 1 < JP) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    a[(p + 1) & 1][rb][0] = ap[rb * 16 * LF_LD + 8 * (jp + 1)];
                    a[(p + 1) & 1][rb][1] = ap[rb * 16 * LF_LD + 8 * (jp + 1) + 4];
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int c = 0; c < NCB; ++c) acc[rb][c] = mfma64(a[p & 1][rb][t], wb[p][c][t], acc[rb][c]);
            if (jp + PF < JP) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) wb[p][c] = wp[(size_t)(c * JP + jp + PF) * 64];
            }
        }
    }
}

// a short product, one channel block per wave, fully unrolled: K = 4 KSTEPS (the last k-step may be zero padded, the last PAIR of the
// fragment layout half empty - its second k-step is then simply not multiplied, as gemm_f64_kernel does not either)
template <int NRB, int KSTEPS>
__device__ __forceinline__ void lf_product_u(const double* As, const double* wf, int lane, f64x4 (&acc)[NRB]) {
    constexpr int JP = (KSTEPS + 1) / 2;
    const int l15 = lane & 15, g = lane >> 4;
    const f64x2* wp = reinterpret_cast<const f64x2*>(wf) + lane;
    f64x2 wb[JP];
#pragma unroll
    for (int jp = 0; jp < JP; ++jp) wb[jp] = wp[(size_t)jp * 64];
    const double* ap = As + l15 * LF_LD + g;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb] = mfma64(ap[rb * 16 * LF_LD + 4 * j], wb[j >> 1][j & 1], acc[rb]);
}

// epilogue of an encoder stage: ReLU(acc + bias) of channel block cb -> tile columns col0 + 16 cb ..
template <int NRB>
__device__ __forceinline__ void lf_store_relu(double* tile, int col0, int cb, const double* bias, int lane, const f64x4 (&acc)[NRB], bool& bad) {
    const int l15 = lane & 15, g = lane >> 4;
    const double b = bias[cb * 16 + l15];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = acc[rb][i] + b;
            bad |= f64_out_of_range(v);
            v = v > 0.0 ? v : 0.0;
            tile[(rb * 16 + g + 4 * i) * LF_LD + col0 + cb * 16 + l15] = v;
        }
}

// Both encoders, their sum and the first layer's q | k | v projection as ONE launch (mdgat.py:184-188, 152-155, 392-393, 227-232):
// seven products whose operands never leave the workgroup's LDS tile - the three-launch-per-layer form's counterpart here is seven
// gemm_f64_kernel launches of K = 4 ... 256, each a few dozen workgroups at one pair per call.  The keypoint chain lives in tile
// columns 128 .., the descriptor chain in columns 0 ..; a stage reads, the workgroup meets at a barrier, the stage's output takes
// its input's place, a second barrier: [hd2 ; hk3] then stand side by side as the 256 input columns of the last encoder layer.
template <int NRB>
__global__ __launch_bounds__(64 * LF_WAVES) void encoder_f64_kernel(EncoderF64Args a) {
    constexpr int TM = 16 * NRB;
    extern __shared__ __attribute__((aligned(16))) double lfs[];      // [TM][LF_LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * TM;
    bool bad = false;
    // ---- inputs: FPFH (33) -> columns 0 .. 32 (33 .. 35: the zero padding of the ninth k-step), x y z saliency -> columns 128 .. 131 ----
    for (int e = tid; e < TM * 40; e += 64 * LF_WAVES) {
        const int r = e / 40, c = e - r * 40;
        const int row = min(row0 + r, a.R - 1);
        if (c < 36) lfs[r * LF_LD + c] = c < 33 ? a.in33[(size_t)row * 33 + c] : 0.0;
        else lfs[r * LF_LD + 128 + (c - 36)] = a.in4[(size_t)row * 4 + (c - 36)];
    }
    __syncthreads();
    f64x4 acc[NRB], acc2[NRB];
    auto zero = [&](f64x4 (&v)[NRB]) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) v[rb] = f64x4{0.0, 0.0, 0.0, 0.0};
    };
    // ---- stage 1: hk1 = ReLU(kenc.0 [4 -> 32]) on waves 0, 1; hd1 = ReLU(denc.0 [33 -> 64]) on waves 2 .. 5 ----
    zero(acc);
    if (wave < 2) lf_product_u<NRB, 1>(lfs + 128, a.wk0 + (size_t)wave * 1 * 128, lane, acc);
    else if (wave < 6) lf_product_u<NRB, 9>(lfs, a.wd0 + (size_t)(wave - 2) * 5 * 128, lane, acc);
    __syncthreads();
    if (wave < 2) lf_store_relu<NRB>(lfs, 128, wave, a.bk0, lane, acc, bad);
    else if (wave < 6) lf_store_relu<NRB>(lfs, 0, wave - 2, a.bd0, lane, acc, bad);
    __syncthreads();
    // ---- stage 2: hk2 = ReLU(kenc.3 [32 -> 64]) on waves 0 .. 3; hd2 = ReLU(denc.3 [64 -> 128]) on all eight ----
    zero(acc); zero(acc2);
    if (wave < 4) lf_product_u<NRB, 8>(lfs + 128, a.wk1 + (size_t)wave * 4 * 128, lane, acc);
    lf_product_u<NRB, 16>(lfs, a.wd1 + (size_t)wave * 8 * 128, lane, acc2);
    __syncthreads();
    if (wave < 4) lf_store_relu<NRB>(lfs, 128, wave, a.bk1, lane, acc, bad);
    lf_store_relu<NRB>(lfs, 0, wave, a.bd1, lane, acc2, bad);
    __syncthreads();
    // ---- stage 3: hk3 = ReLU(kenc.6 [64 -> 128]) ----
    zero(acc);
    lf_product_u<NRB, 16>(lfs + 128, a.wk2 + (size_t)wave * 8 * 128, lane, acc);
    __syncthreads();
    lf_store_relu<NRB>(lfs, 128, wave, a.bk2, lane, acc, bad);
    __syncthreads();
    // ---- stage 4: x = last encoder layers summed: one product over [hd2 ; hk3] (mdgat.py:392-393) ----
    {
        f64x4 ax[NRB][1];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) ax[rb][0] = f64x4{0.0, 0.0, 0.0, 0.0};
        f64x2 wbx[4][1];
        lf_prefetch<1, 32, 4>(a.wl + (size_t)wave * 32 * 128, lane, wbx);
        lf_product<NRB, 1, 32, 4>(lfs, a.wl + (size_t)wave * 32 * 128, lane, ax, wbx);
        __syncthreads();
        const int n = wave * 16 + l15;
        const double b = a.bl[n];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rb * 16 + g + 4 * i;
                const double v = ax[rb][0][i] + b;
                bad |= f64_out_of_range(v);
                lfs[r * LF_LD + n] = v;
                if (row0 + r < a.R) {
                    a.x[(size_t)(row0 + r) * 128 + n] = v;
                    if (a.x32) a.x32[(size_t)(row0 + r) * 128 + n] = (float)v;
                }
            }
    }
    if (a.wq) {
        __syncthreads();
        // ---- stage 5: q | k | v of layer 0 ----
        f64x4 aq[NRB][3];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int c = 0; c < 3; ++c) aq[rb][c] = f64x4{0.0, 0.0, 0.0, 0.0};
        f64x2 wbq[4][3];
        lf_prefetch<3, 16, 4>(a.wq + (size_t)(3 * wave) * 16 * 128, lane, wbq);
        lf_product<NRB, 3, 16, 4>(lfs, a.wq + (size_t)(3 * wave) * 16 * 128, lane, aq, wbq);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n = (3 * wave + c) * 16 + l15;
            const double b = a.bq[n];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + rb * 16 + g + 4 * i;
                    const double v = aq[rb][c][i] + b;
                    bad |= f64_out_of_range(v);
                    if (row < a.R) a.qkv[(size_t)row * 384 + n] = v;
                }
        }
    }
    if (bad) f64_raise(a.guard);
}

template <int NRB>
__global__ __launch_bounds__(64 * LF_WAVES) void layer_tail_f64_kernel(LayerF64Args a) {
    constexpr int TM = 16 * NRB;
    extern __shared__ __attribute__((aligned(16))) double lfs[];      // [TM][LF_LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * TM;
    bool bad = false;
    constexpr bool EARLY = NRB == 1;        // the first weights of a product requested before the barrier in front of it (registers: one-pair tiles only)
    constexpr int PF3 = NRB >= 4 ? 2 : 4;
    f64x2 wb1[4][2], wb2[4][1], wb3[PF3][3];
    if (EARLY) lf_prefetch<2, 32, 4>(a.w1f + (size_t)(2 * wave) * 32 * 128, lane, wb1);

    // ---- input tile [x ; msg] -> LDS (rows beyond R: the last row again; their results are never written) ----
    for (int e = tid; e < TM * 128; e += 64 * LF_WAVES) {
        const int r = e >> 7, c = (e & 127) * 2;
        const int row = min(row0 + r, a.R - 1);
        const double* src = c < 128 ? a.x + (size_t)row * 128 + c : a.msg + (size_t)row * 128 + (c - 128);
        *reinterpret_cast<f64x2*>(lfs + r * LF_LD + c) = *reinterpret_cast<const f64x2*>(src);
    }
    __syncthreads();

    // ---- hid = ReLU(W1 [x ; msg] + b1): 16 channel blocks, two per wave ----
    {
        f64x4 acc[NRB][2];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) { acc[rb][0] = f64x4{0.0, 0.0, 0.0, 0.0}; acc[rb][1] = acc[rb][0]; }
        if (!EARLY) lf_prefetch<2, 32, 4>(a.w1f + (size_t)(2 * wave) * 32 * 128, lane, wb1);
        lf_product<NRB, 2, 32, 4>(lfs, a.w1f + (size_t)(2 * wave) * 32 * 128, lane, acc, wb1);
        if (EARLY) lf_prefetch<1, 32, 4>(a.w2f + (size_t)wave * 32 * 128, lane, wb2);
        __syncthreads();                        // every wave has read the tile: the hidden layer takes its place
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = (2 * wave + c) * 16 + l15;
            const double bias = a.b1[n];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double v = acc[rb][c][i] + bias;
                    bad |= f64_out_of_range(v);
                    v = v > 0.0 ? v : 0.0;
                    lfs[(rb * 16 + g + 4 * i) * LF_LD + n] = v;
                }
        }
    }
    __syncthreads();

    // ---- x += W2 hid + b2: 8 channel blocks, one per wave ----
    {
        f64x4 acc[NRB][1];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = f64x4{0.0, 0.0, 0.0, 0.0};
        // (the residual rows travel under the product)
        const int n = wave * 16 + l15;
        double res[NRB][4];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) res[rb][i] = a.x[(size_t)min(row0 + rb * 16 + g + 4 * i, a.R - 1) * 128 + n];
        if (!EARLY) lf_prefetch<1, 32, 4>(a.w2f + (size_t)wave * 32 * 128, lane, wb2);
        lf_product<NRB, 1, 32, 4>(lfs, a.w2f + (size_t)wave * 32 * 128, lane, acc, wb2);
        if (EARLY && a.w3f) lf_prefetch<3, 16, PF3>(a.w3f + (size_t)(3 * wave) * 16 * 128, lane, wb3);
        __syncthreads();                        // every wave has read the hidden layer: the new x takes its place
        const double bias = a.b2[n];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rb * 16 + g + 4 * i;
                double v = acc[rb][0][i] + bias;
                bad |= f64_out_of_range(v);
                v += res[rb][i];
                bad |= f64_out_of_range(v);
                lfs[r * LF_LD + n] = v;
                if (row0 + r < a.R) {
                    a.x[(size_t)(row0 + r) * 128 + n] = v;
                    if (a.x32) a.x32[(size_t)(row0 + r) * 128 + n] = (float)v;      // the hand-over to the fp32-class layers
                }
            }
    }
    if (a.w3f) {
        __syncthreads();
        // ---- q | k | v of the next layer = W3 x + b3: 24 channel blocks, three per wave ----
        f64x4 acc[NRB][3];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[rb][c] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (!EARLY) lf_prefetch<3, 16, PF3>(a.w3f + (size_t)(3 * wave) * 16 * 128, lane, wb3);
        lf_product<NRB, 3, 16, PF3>(lfs, a.w3f + (size_t)(3 * wave) * 16 * 128, lane, acc, wb3);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n = (3 * wave + c) * 16 + l15;
            const double bias = a.b3[n];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + rb * 16 + g + 4 * i;
                    const double v = acc[rb][c][i] + bias;
                    bad |= f64_out_of_range(v);
                    if (row < a.R) a.qkv[(size_t)row * 384 + n] = v;
                }
        }
    }
    if (bad) f64_raise(a.guard);
}

// W [N][K] row-major -> fragment order [N / 16][ceil(K / 8)][64 lanes][2]: lane (l15 = lane & 15, g = lane >> 4) of channel block cb
// and k-step pair jp holds W[16 cb + l15][8 jp + 4 t + g], t = 0, 1 - the B operand of two consecutive v_mfma_f64_16x16x4_f64;
// zero beyond K (the encoders' K = 4 and K = 33)
__global__ __launch_bounds__(256) void frag64_kernel(const double* W, double* out, int N, int K) {
    const int JP = (K + 7) / 8;
    const size_t total = (size_t)N * JP * 8;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int t = (int)(e & 1), lane = (int)((e >> 1) & 63);
        const size_t blk = e >> 7;
        const int jp = (int)(blk % JP), cb = (int)(blk / JP);
        const int k = 8 * jp + 4 * t + (lane >> 4);
        out[e] = k < K ? W[(size_t)(cb * 16 + (lane & 15)) * K + k] : 0.0;
    }
}

}  // namespace

size_t layer_f64_frag_doubles() { return (size_t)256 * 256 + 128 * 256 + 384 * 128; }

size_t frag64_doubles(int N, int K) { return (size_t)N * ((K + 7) / 8) * 8; }
// the six encoder matrices: kenc.0 | denc.0 | kenc.3 | kenc.6 | denc.3 | last layers summed
size_t encoder_f64_frag_doubles() {
    return frag64_doubles(32, 4) + frag64_doubles(64, 33) + frag64_doubles(64, 32) + frag64_doubles(128, 64) + frag64_doubles(128, 64) + frag64_doubles(128, 256);
}

int launch_frag64(const double* W, double* out, int N, int K, hipStream_t s) {
    if (N % 16) { mdgat_set_error("launch_frag64: %d output channels are not whole fragments", N); return MDGAT_ERR_BAD_ARG; }
    const size_t total = frag64_doubles(N, K);
    hipLaunchKernelGGL(frag64_kernel, dim3((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), dim3(256), 0, s, W, out, N, K);
    return mdgat_check_hip(hipGetLastError(), "frag64 launch");
}

// 0: three launches per layer tail (gemm_f64_kernel); 1: the fused kernel, rows per workgroup chosen by the launch (default);
// 16 / 32 / 64: the fused kernel with that many rows per workgroup (tests, measurements).  MDGAT_F64_LAYER_FUSION in the environment.
static std::atomic<int> g_fusion{-1};
static int fusion_default() {
    static const int v = [] { const char* e = getenv("MDGAT_F64_LAYER_FUSION"); const int m = e ? atoi(e) : 1; return (m == 16 || m == 32 || m == 64) ? m : (m != 0); }();
    return v;
}
static int fusion_mode() { const int v = g_fusion.load(std::memory_order_relaxed); return v < 0 ? fusion_default() : v; }
bool layer_f64_fused() { return fusion_mode() != 0; }
extern "C" int mdgat_set_f64_layer_fusion(int mode) {
    const int m = mode < 0 ? -1 : (mode == 16 || mode == 32 || mode == 64) ? mode : (mode != 0);
    const int prev = g_fusion.exchange(m, std::memory_order_relaxed);
    return prev < 0 ? fusion_default() : prev;
}

static int lf_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    static std::atomic<int> cached[16];
    if (dev >= 0 && dev < 16 && (n = cached[dev].load(std::memory_order_relaxed)) > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < 16) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

int launch_layer_tail_f64(const LayerF64Args& a, hipStream_t s) {
    if (a.R <= 0) return MDGAT_OK;
    // Rows per workgroup.  32 (two workgroups per CU, four waves per SIMD) from a round of the device on; 16 below - one pair of 512
    // keypoints is 64 workgroups instead of 32, and a workgroup's chain of products half as long (mdgat_set_f64_layer_fusion(16 | 32 |
    // 64) forces one).
    int tm = (long)((a.R + 31) / 32) >= 2L * lf_cu_count() ? 32 : 16;
    if (fusion_mode() > 1) tm = fusion_mode();
    const size_t lds = (size_t)tm * LF_LD * sizeof(double);
    const dim3 grid((a.R + tm - 1) / tm);
    auto go = [&](auto kern, auto tag) -> int {
        (void)tag;
        static std::atomic<unsigned long long> done{0};
        if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(kern), lds, done, "layer_tail_f64 LDS")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * LF_WAVES), lds, s, a);
        return mdgat_check_hip(hipGetLastError(), "layer_tail_f64 launch");
    };
    if (tm == 16) return go(layer_tail_f64_kernel<1>, std::integral_constant<int, 1>());
    if (tm == 32) return go(layer_tail_f64_kernel<2>, std::integral_constant<int, 2>());
    return go(layer_tail_f64_kernel<4>, std::integral_constant<int, 4>());
}

int launch_encoder_f64(const EncoderF64Args& a, hipStream_t s) {
    if (a.R <= 0) return MDGAT_OK;
    int tm = (long)((a.R + 31) / 32) >= 2L * lf_cu_count() ? 32 : 16;
    if (fusion_mode() == 16 || fusion_mode() == 32) tm = fusion_mode();
    const size_t lds = (size_t)tm * LF_LD * sizeof(double);
    const dim3 grid((a.R + tm - 1) / tm);
    auto go = [&](auto kern, auto tag) -> int {
        (void)tag;
        static std::atomic<unsigned long long> done{0};
        if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(kern), lds, done, "encoder_f64 LDS")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * LF_WAVES), lds, s, a);
        return mdgat_check_hip(hipGetLastError(), "encoder_f64 launch");
    };
    if (tm == 16) return go(encoder_f64_kernel<1>, std::integral_constant<int, 1>());
    return go(encoder_f64_kernel<2>, std::integral_constant<int, 2>());
}
