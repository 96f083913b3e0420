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
#include "coop_chain.hpp"

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
    // (deeper weight prefetch at one row block per workgroup - 8 / 16 pairs of k-steps for mlp.0 / mlp.3 - changes nothing: B = 1, 2, 8
    // within 0.5 % either way; the chain of 576 matrix instructions per SIMD is 15 us of its 26)
    constexpr int PF1 = 4, PF2 = 4;
    f64x2 wb1[PF1][2], wb2[PF2][1], wb3[PF3][3];
    if (EARLY) lf_prefetch<2, 32, PF1>(a.w1f + (size_t)(2 * wave) * 32 * 128, lane, wb1);

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
        if (!EARLY) lf_prefetch<2, 32, PF1>(a.w1f + (size_t)(2 * wave) * 32 * 128, lane, wb1);
        lf_product<NRB, 2, 32, PF1>(lfs, a.w1f + (size_t)(2 * wave) * 32 * 128, lane, acc, wb1);
        if (EARLY) lf_prefetch<1, 32, PF2>(a.w2f + (size_t)wave * 32 * 128, lane, wb2);
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
        if (!EARLY) lf_prefetch<1, 32, PF2>(a.w2f + (size_t)wave * 32 * 128, lane, wb2);
        lf_product<NRB, 1, 32, PF2>(lfs, a.w2f + (size_t)wave * 32 * 128, lane, acc, wb2);
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

// ---- launches of at most a quarter of the CUs in 16-row blocks (one pair per call, test.py:132): a block is shared by a CLUSTER ----
// The kernel above gives a 16-row block to one workgroup: one pair of 512 keypoints is 64 workgroups on 256 CUs, each with the
// block's whole chain of 2 304 matrix instructions (15 us of matrix pipe alone; 26 us per launch, 18 launches: 40 % of the forward).
// Here FOUR workgroups (ranks) share the block and split the OUTPUT CHANNELS of every product between them - a quarter of the
// chain each, one channel block per wave (4 + 2 + 6 waves busy) - and exchange the hidden layer and the new x through memory:
// every rank stores its channels (plain stores when the four sit on one XCD - they do under round-robin dispatch, blockIdx % 8,
// checked per launch through the ranks' XCC ids - they stay in that L2; write-through otherwise), waits for them to be acknowledged,
// raises its flag, polls its partners' flags and fetches their channels with L1-bypassing loads.  Same operand roles, same order over
// k, every channel block by exactly one wave: BIT-IDENTICAL to the kernel above (and to the three-launch form).
// Flags: 64-bit words in a buffer the library owns per device, set to the launch's number (CoopChain::epoch) - nothing to clear per
// launch; launches of waiting kernels are chained one at a time per device (coop_chain.hpp), graph capture takes the kernel above.
constexpr int LC_RANKS = 4;
constexpr int LC_WORDS = 4;             // per rank: XCC id word | hidden-layer flag | x flag | pad

__device__ __forceinline__ bool lc_wait(const unsigned long long* f, unsigned long long want) {
    long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
        if (++spins > (1L << 22)) return false;
    return true;
}

// one channel block of a product with ALL its weights requested at once (a wave of the clustered kernel has one block and registers
// to spare: JP loads of 1 KB in flight instead of four - at four the product runs at the L2's latency, 175 ns a pair of k-steps
// against 53 ns of matrix pipe); the same order over k as lf_product
template <int JP>
__device__ __forceinline__ void lc_load_all(const double* wf, int lane, f64x2 (&w)[JP]) {
    const f64x2* wp = reinterpret_cast<const f64x2*>(wf) + lane;
#pragma unroll
    for (int jp = 0; jp < JP; ++jp) w[jp] = wp[(size_t)jp * 64];
}
template <int JP>
__device__ __forceinline__ void lc_product(const double* As, int lane, f64x4& acc, const f64x2 (&w)[JP]) {
    const double* ap = As + (lane & 15) * LF_LD + (lane >> 4);
#pragma unroll
    for (int jp = 0; jp < JP; ++jp) {
        acc = mfma64(ap[8 * jp], w[jp][0], acc);
        acc = mfma64(ap[8 * jp + 4], w[jp][1], acc);
    }
}

__global__ __launch_bounds__(64 * LF_WAVES) void layer_tail_f64_cluster_kernel(LayerF64Args a, unsigned long long* flags, unsigned long long epoch) {
    extern __shared__ __attribute__((aligned(16))) double lfs[];      // [16][LF_LD]
    __shared__ int lc_same, lc_dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // blockIdx = (cluster / 8) * 32 + rank * 8 + cluster % 8: the four ranks of a cluster share blockIdx % 8
    const int cl = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7), rank = (blockIdx.x >> 3) & 3;
    const int row0 = cl * 16;
    if (row0 >= a.R) return;                                          // (all four ranks)
    unsigned long long* fl = flags + (size_t)cl * LC_RANKS * LC_WORDS;
    bool bad = false;
    if (tid == 0) {
        lc_dead = 0;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;      // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(fl + rank * LC_WORDS, (epoch << 8) | (xcc + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    f64x2 w1[32], w2[32], w3[16];
    if (wave < 4) lc_load_all<32>(a.w1f + (size_t)(4 * rank + wave) * 32 * 128, lane, w1);
    for (int e = tid; e < 16 * 128; e += 64 * LF_WAVES) {
        const int r = e >> 7, c = (e & 127) * 2;
        const int row = min(row0 + r, a.R - 1);
        const double* src = c < 128 ? a.x + (size_t)row * 128 + c : a.msg + (size_t)row * 128 + (c - 128);
        *reinterpret_cast<f64x2*>(lfs + r * LF_LD + c) = *reinterpret_cast<const f64x2*>(src);
    }
    __syncthreads();

    // ---- hid = ReLU(W1 [x ; msg] + b1): channel blocks 4 rank .. 4 rank + 3, one per wave; an idle wave compares the ranks' XCDs ----
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    if (wave < 4) lc_product<32>(lfs, lane, acc, w1);
    else if (wave == 7) {
        bool same = true;
        if (lane < LC_RANKS && lane != rank) {
            const unsigned long long mine = (epoch << 8) | ((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf) + 1u);
            if (!lc_wait(fl + lane * LC_WORDS, epoch << 8)) lc_dead = 1;
            same = __hip_atomic_load(fl + lane * LC_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine;
        }
        const bool all_same = !__any(!same);
        if (lane == 0) lc_same = all_same;
    }
    __syncthreads();                            // every wave has read the tile: the hidden layer takes its place
    const bool same_xcd = lc_same != 0;
    auto publish = [&](double* p, double v) {
        if (same_xcd) *p = v;
        else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (wave < 4) {
        const int n = (4 * rank + wave) * 16 + l15;
        const double bias = a.b1[n];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = acc[i] + bias;
            bad |= f64_out_of_range(v);
            v = v > 0.0 ? v : 0.0;
            const int r = g + 4 * i;
            lfs[r * LF_LD + n] = v;
            if (row0 + r < a.R) publish(a.hid + (size_t)(row0 + r) * 256 + n, v);
        }
    }
    // the exchange: stores acknowledged (then the next product's weights are requested: they travel under the exchange), the
    // workgroup's flag, the partners' flags
    auto exchange = [&](int word) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (word == 1) { if (wave < 2) lc_load_all<32>(a.w2f + (size_t)(2 * rank + wave) * 32 * 128, lane, w2); }
        else if (wave < 6) lc_load_all<16>(a.w3f + (size_t)(6 * rank + wave) * 16 * 128, lane, w3);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(fl + rank * LC_WORDS + word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < LC_RANKS && tid != rank)
            if (!lc_wait(fl + tid * LC_WORDS + word, epoch)) lc_dead = 1;
        __syncthreads();
    };
    exchange(1);
    {   // the partners' 3 x 64 hidden channels of the 16 rows: six doubles per thread, all in flight
        double v[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int e = tid + 512 * t, r = e / 192, q = e - r * 192;
            const int p = q >> 6, pr = p + (p >= rank);
            v[t] = __hip_atomic_load(a.hid + (size_t)min(row0 + r, a.R - 1) * 256 + pr * 64 + (q & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int e = tid + 512 * t, r = e / 192, q = e - r * 192;
            const int p = q >> 6, pr = p + (p >= rank);
            lfs[r * LF_LD + pr * 64 + (q & 63)] = v[t];
        }
    }
    __syncthreads();

    // ---- x += W2 hid + b2: channel blocks 2 rank, 2 rank + 1 ----
    {
        acc = f64x4{0.0, 0.0, 0.0, 0.0};
        const int n = (2 * rank + wave) * 16 + l15;                   // (waves 0, 1)
        double res[4];
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) res[i] = a.x[(size_t)min(row0 + g + 4 * i, a.R - 1) * 128 + n];
            lc_product<32>(lfs, lane, acc, w2);
        }
        __syncthreads();                        // every wave has read the hidden layer: the new x takes its place
        if (wave < 2) {
            const double bias = a.b2[n];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = g + 4 * i;
                double v = acc[i] + bias;
                bad |= f64_out_of_range(v);
                v += res[i];
                bad |= f64_out_of_range(v);
                lfs[r * LF_LD + n] = v;
                if (row0 + r < a.R) {
                    publish(a.x + (size_t)(row0 + r) * 128 + n, v);
                    if (a.x32) a.x32[(size_t)(row0 + r) * 128 + n] = (float)v;
                }
            }
        }
    }
    if (a.w3f) {
        exchange(2);
        {   // the partners' 3 x 32 channels of the new x: three doubles per thread
            double v[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = tid + 512 * t, r = e / 96, q = e - r * 96;
                const int p = q >> 5, pr = p + (p >= rank);
                v[t] = __hip_atomic_load(a.x + (size_t)min(row0 + r, a.R - 1) * 128 + pr * 32 + (q & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = tid + 512 * t, r = e / 96, q = e - r * 96;
                const int p = q >> 5, pr = p + (p >= rank);
                lfs[r * LF_LD + pr * 32 + (q & 31)] = v[t];
            }
        }
        __syncthreads();
        // ---- q | k | v of the next layer: channel blocks 6 rank .. 6 rank + 5 ----
        if (wave < 6) {
            acc = f64x4{0.0, 0.0, 0.0, 0.0};
            lc_product<16>(lfs, lane, acc, w3);
            const int n = (6 * rank + wave) * 16 + l15;
            const double bias = a.b3[n];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + g + 4 * i;
                const double v = acc[i] + bias;
                bad |= f64_out_of_range(v);
                if (row < a.R) a.qkv[(size_t)row * 384 + n] = v;
            }
        }
    }
    if (bad || lc_dead) f64_raise(a.guard);          // (a partner that never answered: the call is refused, not wrong)
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

// 0: three launches per layer tail (gemm_f64_kernel); 1: the fused kernels, form and rows per workgroup chosen by the launch (default);
// 2: the same without the clustered form of small launches; 16 / 32 / 64: the one-workgroup-per-block kernel with that many rows per
// workgroup (tests, measurements).  MDGAT_F64_LAYER_FUSION in the environment.
static std::atomic<int> g_fusion{-1};
static int fusion_default() {
    static const int v = [] { const char* e = getenv("MDGAT_F64_LAYER_FUSION"); const int m = e ? atoi(e) : 1; return (m == 16 || m == 32 || m == 64 || m == 2) ? m : (m != 0); }();
    return v;
}
static int fusion_mode() { const int v = g_fusion.load(std::memory_order_relaxed); return v < 0 ? fusion_default() : v; }
bool layer_f64_fused() { return fusion_mode() != 0; }
extern "C" int mdgat_set_f64_layer_fusion(int mode) {
    const int m = mode < 0 ? -1 : (mode == 16 || mode == 32 || mode == 64 || mode == 2) ? mode : (mode != 0);
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
    if (fusion_mode() >= 16) tm = fusion_mode();
    // Launches of at most a quarter of the CUs in 16-row blocks (one pair of 512 keypoints: 64): four workgroups per block
    // (layer_tail_f64_cluster_kernel; bit-identical).  Not under graph capture (its flags count launches), not without the scratch,
    // not when a tile height is forced or mdgat_set_f64_layer_fusion(2) asks for the one-workgroup-per-block kernel.
    const int nblk = (a.R + 15) / 16;
    if (fusion_mode() == 1 && a.hid && LC_RANKS * nblk <= lf_cu_count() && !coop_stream_capturing(s)) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        CoopChain& ch = coop_chain_of(dev);
        std::lock_guard<std::recursive_mutex> lock(ch.m);
        constexpr size_t flag_bytes = (size_t)256 * LC_RANKS * LC_WORDS * sizeof(unsigned long long);      // up to 256 clusters
        if (!ch.cluster_flags) {
            void* p = nullptr;
            if (int rc = mdgat_check_hip(hipMalloc(&p, flag_bytes), "layer_tail_f64 cluster flags")) return rc;
            if (int rc = mdgat_check_hip(hipMemset(p, 0, flag_bytes), "layer_tail_f64 cluster flags (clear)")) { (void)hipFree(p); return rc; }
            ch.cluster_flags = static_cast<unsigned long long*>(p);
        }
        if (nblk <= 256) {
            if (int rc = mdgat_check_hip(coop_chain_wait(ch, s), "layer_tail_f64: wait for the previous waiting launch")) return rc;
            const size_t lds = (size_t)16 * LF_LD * sizeof(double);
            hipLaunchKernelGGL(layer_tail_f64_cluster_kernel, dim3((unsigned)(((nblk + 7) / 8) * 32)), dim3(64 * LF_WAVES), lds, s, a, ch.cluster_flags, ++ch.epoch);
            if (int rc = mdgat_check_hip(hipGetLastError(), "layer_tail_f64 (clustered) launch")) return rc;
            return mdgat_check_hip(coop_chain_record(ch, s), "layer_tail_f64: record");
        }
    }
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
