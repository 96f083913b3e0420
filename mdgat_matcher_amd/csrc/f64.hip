// Reference-exact arithmetic (mdgat_config.arithmetic == MDGAT_ARITH_FP64): the encoders and the propagation layers up to and
// including the last DYNAMIC layer in fp64 on v_mfma_f64_16x16x4_f64.
//
// Why: dynamic_attention (mdgat.py:196-210) keeps the k largest logits of a row - a discontinuous function of the logits.  The
// reference computes them in fp64; the fp32-class (split-f16) kernels of the default path resolve ~1e-5, so about 1.5 rows per
// pair at N = 512 keep another key than the reference does and Z moves by 1e-4 ... 8e-4 around that keypoint.  Re-deciding near
// ties inside a layer does not help (repair.hip): the flips arrive with the layer's INPUT, the accumulated error of the layers
// before.  So this mode carries the residual stream x, q / k / v, the logits and the softmax in fp64 through the last layer
// that selects; nothing behind that layer is discontinuous, and the forward hands over to the split-f16 kernels there (api.hip).
//
// gfx950 mapping.  v_mfma_f64_16x16x4_f64: A 16x4, B 4x16 one double per lane (row / col = lane & 15, k = lane >> 4); C/D
// col = lane & 15, row = (lane >> 4) + 4 reg - NOT the f32 layout (cdna_hip_programming.md section 3).  64 cycles per
// instruction and SIMD: 32 FLOP/clk/SIMD, 78.6 TFLOP/s at 2.4 GHz - the vector fp64 rate; what the matrix instruction buys
// is operand economy (1024 FMAs from two register pairs), not rate.  And it SHARES the SIMD's issue with the fp64 vector
// instructions (PMC: MFMA-busy + vector-busy add up to the kernels' time, profiles/NOTES_r5.md section 1): a vector
// instruction in a loop costs a sixteenth of a matrix instruction, so the loops are written for few of them (section 8: no NaN
// canonicalisation, the exponential's coefficients as scalar operands, fragments reloaded in place).  Otherwise the kernels are
// plain - LDS tiles for the GEMM, fragments straight from L2 for the attention - three resident workgroups per CU, and each
// launcher sizes its tiles by the launch (64 x 64 tiles, 64-deep chunks, 16-query workgroups for a pair or two: section 9).
//
//   gemm_f64_kernel       C = act(A W^T + b) (+ R): every Conv1d(k=1) of the path (mdgat.py:34-46 after BN folding; 152-155,
//                         184-188, 227-232, 246-248, 274), 64 x 64 / 64 x 128 tiles, 32- (64-) deep K chunks through LDS.
//   attention_f64_kernel  attention / dynamic_attention (mdgat.py:190-210) for 16 (or 32) queries of a (pair, frame, head) per
//                         workgroup, the KEYS split over the four waves: S^T = K Q^T puts a query's logits into the four lanes
//                         (q, q + 16, q + 32, q + 48), the D fragment of a 16-key block is the B operand of the P.V product as
//                         it stands.  Full attention: online softmax per wave, the waves combined at the end - or, for launches
//                         that fill the chip, the QUERIES split over the waves (SOLO: a wave walks all keys, nothing to combine).
//                         K / V fragments by buffer loads (scalar descriptor, constant lane offset).  Dynamic attention:
//                         pass A writes the fp32 roundings of the logits to LDS, one wave per row finds the
//                         exact k-th largest of them, pass B recomputes the fp64 logits and keeps what lies above; logits whose
//                         fp32 roundings TIE at the k-th place are ranked by their fp64 values (a short list per row, resolved
//                         after the pass).  Rounding is monotone, so the selection is the fp64 top-k exactly.
#include "common.hpp"
#include "f64.hpp"
#include "f64_dev.hpp"
#ifdef F64_TRACE
// measurement build only (tools/ab_build.sh f64 trace -DF64_TRACE; tools/f64_trace.py): s_memtime at the phase boundaries of the
// dynamic kernel, waves 0 and 3 of one workgroup in the middle of the grid
__device__ long long g_f64_trace[2 * 32];
extern "C" int mdgat_f64_trace_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f64_trace), n * sizeof(long long)); }
#define FT(k) do { if (blockIdx.x == gridDim.x / 2 + 3 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3)) \
    g_f64_trace[((threadIdx.x >> 6) == 3) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FT(k) do {} while (0)
#endif

#include "row_search.hpp"
#include "exp2_tab256.hpp"

namespace {

// ================================================================================================ GEMM
constexpr int G_BM = 64, G_KC = 32;      // row pitch KC + 2 doubles (34: 68 dwords; 66: 132): the 64 lanes of a fragment read (row l15, k g) hit 64 banks

// FAST: whole tiles only (M a multiple of 64, N of 32 WN, K and K0 of 32, 16-byte aligned rows): a chunk comes from ONE source,
// the tile copies are 16-byte loads without a predicate each (the general form runs 24 guarded 8-byte loads per thread and
// chunk, every one its own exec-masked branch)
// KC: depth of a chunk.  32; 64 (FAST only) for launches of less than a round of workgroups, which are a chain of chunk round trips
// (load -> registers -> LDS -> barrier, ~2 us each with nothing else on the CU to cover it): half as many.
template <int WN, bool FAST, int KC = G_KC>      // 16-column blocks per wave: workgroup tile 64 x (32 WN)
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmF64Args a) {
    static_assert(KC == 32 || (KC == 64 && FAST), "chunk depth");
    if (a.batch > 1) { a.A0 += (size_t)blockIdx.z * a.sA; a.W += (size_t)blockIdx.z * a.sW; a.C += (size_t)blockIdx.z * a.sC; }
    constexpr int BN = 32 * WN, G_LD = KC + 2, G_KC = KC, RSH = KC == 64 ? 5 : 4, RMASK = (1 << RSH) - 1;
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double* As = gsm;                       // [64][G_LD]
    double* Ws = gsm + G_BM * G_LD;         // [BN][G_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int row0 = blockIdx.x * G_BM, col0 = blockIdx.y * BN;
    constexpr int NA = G_BM * G_KC / 256, NW = BN * G_KC / 256;
    double ra[NA], rw[NW];
    auto fetch = [&](int k0) {
        if (FAST) {
            const double* src = k0 < a.K0 ? a.A0 + k0 : a.A1 + (k0 - a.K0);
            const int ld = k0 < a.K0 ? a.lda0 : a.lda1;
#pragma unroll
            for (int u = 0; u < NA / 2; ++u) {
                const int idx = tid + 256 * u, r = idx >> RSH, c = (idx & RMASK) * 2;
                const f64x2 v = *reinterpret_cast<const f64x2*>(src + (size_t)(row0 + r) * ld + c);
                ra[2 * u] = v[0]; ra[2 * u + 1] = v[1];
            }
#pragma unroll
            for (int u = 0; u < NW / 2; ++u) {
                const int idx = tid + 256 * u, r = idx >> RSH, c = (idx & RMASK) * 2;
                const f64x2 v = *reinterpret_cast<const f64x2*>(a.W + (size_t)(col0 + r) * a.ldw + k0 + c);
                rw[2 * u] = v[0]; rw[2 * u + 1] = v[1];
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + 256 * u, r = idx >> 5, c = idx & 31;
            const int row = row0 + r, kk = k0 + c;
            double v = 0.0;
            if (row < a.M && kk < a.K) v = kk < a.K0 ? a.A0[(size_t)row * a.lda0 + kk] : a.A1[(size_t)row * a.lda1 + (kk - a.K0)];
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int idx = tid + 256 * u, r = idx >> 5, c = idx & 31;
            const int n = col0 + r, kk = k0 + c;
            rw[u] = (n < a.N && kk < a.K) ? a.W[(size_t)n * a.ldw + kk] : 0.0;
        }
    };
    auto stash = [&]() {
        if (FAST) {
#pragma unroll
            for (int u = 0; u < NA / 2; ++u) { const int idx = tid + 256 * u; *reinterpret_cast<f64x2*>(As + (idx >> RSH) * G_LD + (idx & RMASK) * 2) = f64x2{ra[2 * u], ra[2 * u + 1]}; }
#pragma unroll
            for (int u = 0; u < NW / 2; ++u) { const int idx = tid + 256 * u; *reinterpret_cast<f64x2*>(Ws + (idx >> RSH) * G_LD + (idx & RMASK) * 2) = f64x2{rw[2 * u], rw[2 * u + 1]}; }
            return;
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) { const int idx = tid + 256 * u; As[(idx >> 5) * G_LD + (idx & 31)] = ra[u]; }
#pragma unroll
        for (int u = 0; u < NW; ++u) { const int idx = tid + 256 * u; Ws[(idx >> 5) * G_LD + (idx & 31)] = rw[u]; }
    };
    f64x4 acc[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    fetch(0);
    stash();
    __syncthreads();
    const double* ap = As + (wm * 32 + l15) * G_LD + g;
    const double* wp = Ws + (wn * 16 * WN + l15) * G_LD + g;
    for (int k0 = 0; k0 < a.K; k0 += G_KC) {
        const bool more = k0 + G_KC < a.K;
        if (more) fetch(k0 + G_KC);                     // the next chunk travels while this one is multiplied
        const int rem = a.K - k0;
        const int steps = rem >= G_KC ? G_KC / 4 : (rem + 3) >> 2;
        // (Measured and dropped, profiles/NOTES_r5.md section 1: the eight steps of a whole chunk unrolled - the compiler hoists every
        // fragment read: 296 registers, one wave per SIMD, 40.8 -> 29.6 TFLOP/s at 32768 x 256 x 256; the reads of step j + 1 issued
        // before the products of step j by rotating two register sets: 39.8 -> 37.8.  The second wave of the SIMD already covers
        // the read latency; what the matrix pipe waits for is elsewhere.)
        for (int j = 0; j < steps; ++j) {
            double fa[2], fw[WN];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = ap[i * 16 * G_LD + 4 * j];
#pragma unroll
            for (int i = 0; i < WN; ++i) fw[i] = wp[i * 16 * G_LD + 4 * j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < WN; ++n) acc[i][n] = mfma64(fa[i], fw[n], acc[i][n]);
        }
        if (more) {
            __syncthreads();
            stash();
            __syncthreads();
        }
    }
    // D: lane (column l15, g), register i -> row g + 4 i of the 16 x 16 block
    bool bad = false;
#pragma unroll
    for (int nb = 0; nb < WN; ++nb) {
        const int n = col0 + wn * 16 * WN + nb * 16 + l15;
        if (n >= a.N) continue;
        const double bias = a.bias ? a.bias[n] : 0.0;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + wm * 32 + mb * 16 + g + 4 * i;
                if (row >= a.M) continue;
                double v = acc[mb][nb][i] * a.scale + bias;          // (scale = 1: the same bits as acc + bias, contracted or not)
                bad |= f64_out_of_range(v);
                if (a.relu) v = v > 0.0 ? v : 0.0;
                if (a.R) { v += a.R[(size_t)row * a.ldr + n]; bad |= f64_out_of_range(v); }
                a.C[(size_t)row * a.ldc + n] = v;
            }
    }
    if (bad) f64_raise(a.guard);
}

// ================================================================================================ attention
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m, 64);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// a value of the four lanes (q, q + 16, q + 32, q + 48) of a row combined
__device__ __forceinline__ double quad_max(double v) { v = fmax(v, shfl_xor_f64(v, 16)); return fmax(v, shfl_xor_f64(v, 32)); }
__device__ __forceinline__ double quad_sum(double v) { v += shfl_xor_f64(v, 16); return v + shfl_xor_f64(v, 32); }

// exp(x) for the softmax numerators: x <= TAU_LAZY (x <= 0 except under the lazy reference of the full-attention loop), -inf for
// masked keys.  Table-driven: x = (256 q + j) ln2 / 256 + r, |r| <= ln2 / 512 = 1.35e-3,
//     exp(x) = 2^q T[j] (1 + r + r^2 / 2 + r^3 / 6 + r^4 / 24),      T[j] = 2^(j / 256) correctly rounded, 2 KB of LDS (exp2_tab256.hpp)
// (truncation r^5 / 120 = 3.8e-17).  n = 256 q + j falls out of the low word of x (256 / ln 2) + 1.5 2^52, the reduction is ONE fma
// against the correctly rounded ln2 / 256 (its rounding error acts like a relative perturbation of x by 2^-53: an ulp of the logit
// itself), the polynomial is a product and three fmas with at most one scalar operand each, one fma scales the table entry and
// v_ldexp_f64 applies 2^q.  Ten fp64 and three integer instructions and a ds_read_b64; the degree-12 polynomial this replaces ran 19 (+ a v_mov_b64 the compiler
// rematerialised for the leading coefficient) - on this part an fp64 vector instruction issues in the slot of a sixteenth of a
// v_mfma_f64_16x16x4 and the two share the pipe (profiles/NOTES_r5.md section 1), so the attention loops are their vector
// instruction count.  Keys masked with -inf get exp(-700) = 1e-304 instead of 0: nothing against a row's largest term, which is 1.
typedef __attribute__((address_space(3))) const double lds_cdouble;
struct ExpConst { double magic; };        // 1.5 2^52 held in a vector register pair for the whole kernel (the fma that uses it has its one
                                          // scalar slot taken by 256 / ln 2; left to the compiler the constant is rematerialised per call)
__device__ __forceinline__ ExpConst exp_const() {
    double m = 0x1.8p+52;
    asm volatile("" : "+v"(m));
    return ExpConst{m};
}
__device__ __forceinline__ double exp_fast(double x, const double* tab_, const ExpConst& ec) {
    lds_cdouble* tab = (lds_cdouble*)tab_;
    x = fmax(x, -700.0);
    double tm, r, a, b, s;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(tm) : "v"(x), "s"(0x1.71547652b82fep+8), "v"(ec.magic));       // x 256 / ln 2 + 1.5 2^52
    const int n = (int)(unsigned)__builtin_bit_cast(unsigned long long, tm);
    const double nd = tm - ec.magic;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(nd), "s"(-0x1.62e42fefa39efp-9), "v"(x));              // x - n ln2 / 256
    const double r2 = r * r;
    asm("v_fma_f64 %0, %1, %2, 0.5" : "=v"(a) : "v"(r), "s"(0x1.5555555555555p-3));                        // 1/2 + r / 6
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(b) : "v"(r2), "s"(0x1.5555555555555p-5), "v"(a));               // ... + r^2 / 24
    s = __builtin_fma(r2, b, r);                                                                          // exp(r) - 1
    const double T = tab[n & 255];
    return ldexp(__builtin_fma(T, s, T), n >> 8);       // (v_ashrrev + v_ldexp_f64; shift, mask and a 64-bit add into T's exponent field: one more, same time)
}
constexpr double TAU_LAZY = 8.0;          // full attention: the running reference of a row moves only when a logit exceeds it by this much

constexpr int A_LIST = 32;        // logits tied (as fp32 roundings) at the k-th place that are ranked by their fp64 values; more: key order
struct RowSel { float thr; int mode; int aux; int pad; };
// (on the fp32 roundings of the logits: the roundings equal to thr are TIED at the k-th place)
// mode 0: keep every rounding >= thr.  mode 1: keep those above thr, and of the tied those with key <= aux.  mode 2: keep those above
// thr; the tied ones go to the row's list and the `aux` largest of them (fp64 value, then lower key) are added after the pass.

// the row search of row_search.hpp in the terms of this kernel's pass B
template <int NV>
__device__ RowSel f64_row_select(const float* row, int nk, int k, float zq, int lane, int* hist) {
    const RowSearch r = topk_row_search<NV>(row, nk, k, zq, lane, A_LIST, hist);
    if (k >= nk || r.c_ge == k) return RowSel{r.thr, 0, 0, 0};
    if (r.c_ge - r.c_gt <= A_LIST) return RowSel{r.thr, 2, k - r.c_gt, 0};
    return RowSel{r.thr, 1, r.keylim, 0};
}

// LDS carve of the attention kernel.  The rounding images (dynamic layers) are dead once the rows have been searched - pass B
// recomputes the logits - so the output partials of the final combine share their space.
struct AttnLds {
    double* tab;         // [256] 2^(j / 256) (exp_fast)
    double* obuf;        // [4 waves][QT][32]
    double* mw;          // [4][QT] row maxima per wave
    double* lw;          // [4][QT] row sums per wave
    double* lS;          // [QT][A_LIST]   fp64 logits of the tied candidates
    int* lkey;           // [QT][A_LIST]   their keys (bit 31: kept, set by the resolution)
    int* lcount;         // [QT]
    RowSel* sel;         // [QT]
    int* hist;           // [4 waves][RS_HIST_INTS] radix-select histograms (row_search.hpp)
    float* img;          // [QT][imgld]  (aliases obuf)
};
__device__ __forceinline__ AttnLds attn_lds(double* base, int QT, bool topk, int hist_ints) {
    AttnLds s;
    s.tab = base; base += 256;
    s.mw = base; base += 4 * QT;
    s.lw = base; base += 4 * QT;
    if (!topk) {         // full attention: the row statistics and the output partials only (34 KB at 32 queries: four workgroups per CU)
        s.lS = nullptr; s.lkey = nullptr; s.lcount = nullptr; s.sel = nullptr; s.hist = nullptr;
        s.obuf = base; s.img = nullptr;
        return s;
    }
    s.lS = base; base += QT * A_LIST;
    s.lkey = reinterpret_cast<int*>(base);
    s.lcount = s.lkey + QT * A_LIST;
    s.sel = reinterpret_cast<RowSel*>(s.lcount + QT);
    s.hist = reinterpret_cast<int*>(s.sel + QT);
    s.obuf = reinterpret_cast<double*>(s.hist + hist_ints);
    s.img = reinterpret_cast<float*>(s.obuf);
    return s;
}
// Histograms of the row select.  One row per wave (more than 512 keys): RS_HIST_INTS per wave.  Four rows per wave (topk_quad_search,
// at most 512 keys): a row's histogram lives in the row's own rounding image (every value is in registers before the first clear)
// whenever the image row is long enough - more than 256 keys; launches with a frame of at most 256 keys get a region of their own.
int attn_hist_ints(int nk_min, int nk_max) {
    if (nk_max > 512) return 4 * RS_HIST_INTS;
    return ((nk_min + 63) & ~63) + 4 >= RQ_HIST_INTS ? 16 : 16 * RQ_HIST_INTS;
}
size_t attn_lds_bytes(int QT, int nk_max, bool topk, int hist_ints) {
    const size_t ob = (size_t)4 * QT * 32 * 8;
    if (!topk) return (size_t)(256 + 8 * QT) * 8 + ob;
    const size_t fixed = ((size_t)256 + 8 * QT + (size_t)QT * A_LIST) * 8 + ((size_t)QT * A_LIST + QT) * 4 + (size_t)QT * sizeof(RowSel) + (size_t)hist_ints * 4;
    const size_t im = topk ? (size_t)QT * (((nk_max + 63) & ~63) + 4) * 4 : 0;
    return fixed + (ob > im ? ob : im);
}

// KEEP (dynamic attention, at most 512 keys = 8 blocks per wave, QB = 1): the fp64 logits of pass A stay in registers (64 of them)
// and pass B neither reads K nor multiplies again - a third of the matrix work of the recomputing form.
// (32-query full attention: 146 registers, three waves per SIMD.  Held to 128 for a fourth - amdgpu_waves_per_eu(4), 16 spilled - it
// loses: 239 -> 260 us at batch 32.)
// (three waves per SIMD at least - 168 registers: the 512-key dynamic instance sits right at that line, and one register more costs it
// a third of its occupancy: 666 -> 797 us per launch at batch 64)
// SOLO (full attention, launches that fill the chip several times over): the QUERIES of a workgroup are split over its waves instead
// of the keys - a wave owns 16 QB queries and walks all the blocks of the frame, so there is nothing to combine: no partials through
// LDS, no barrier, no second pass with its exponentials, and the prologue (query fragments) and epilogue are paid once per 32
// blocks instead of once per 8.  Per 16 QB queries x 512 keys: 256 matrix instructions either way, ~1 250 vector instructions
// instead of ~2 140.  Not for small launches: a wave's chain of blocks is four times as long, and a pair of 512 keypoints is
// only 128 such waves.  (A row's terms are summed in another order than in the split form: the two agree to rounding, not bit for bit.)
template <bool TOPK, int QB, bool TAP, bool KEEP = false, bool SOLO = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attention_f64_kernel(AttnF64Args a) {
    static_assert(!KEEP || (TOPK && QB == 1), "KEEP: dynamic attention, one query block");
    static_assert(!SOLO || !TOPK, "SOLO: full attention");
    constexpr int QT = 16 * QB;
    extern __shared__ __attribute__((aligned(16))) double asmem[];
    const AttnLds sm = attn_lds(asmem, QT, TOPK, a.hist_ints);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // 1-D grid, XCD aware: workgroups i, i + 8, i + 16, ... (one XCD under round-robin dispatch) walk the query tiles of ONE unit
    // (pair, frame, head), whose keys and values they read straight from that XCD's L2
    const int slot = blockIdx.x >> 3;
    const int unit = (slot / a.tiles) * 8 + (blockIdx.x & 7), tile = slot % a.tiles;
    if (unit >= a.units) return;
    const int head = unit & 3, side = (unit >> 2) & 1, b = unit >> 3;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N, q_off = side ? a.N : 0;
    const int src = a.cross ? 1 - side : side;
    const int nk = src ? a.M : a.N, k_off = src ? a.N : 0;
    const int q0 = SOLO ? (tile * 4 + wave) * QT : tile * QT;
    if (!SOLO && q0 >= nq) return;
    sm.tab[tid] = MDGAT_EXP2_TAB256[tid];        // (256 threads)
    if (!TOPK) __syncthreads();                  // (the dynamic kernels pass two barriers before their first exponential)
    if (SOLO && q0 >= nq) return;                // (a wave of its own: behind the workgroup's only barrier)
    const ExpConst ec = exp_const();
    const int imgld = ((nk + 63) & ~63) + 4;
    const double* kbase = a.qkv + ((size_t)b * P + k_off) * 384 + 128 + head * 32;
    const double* vbase = kbase + 128;

    // this lane's query fragments: dims 8 g + j of query l15 (B operand of k-step j), pre-scaled by 1 / sqrt(32) (mdgat.py:192, 201)
    double qf[QB][8];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + qb * 16 + l15, nq - 1);
        const f64x2* p = reinterpret_cast<const f64x2*>(a.qkv + ((size_t)b * P + q_off + qrow) * 384 + head * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f64x2 v = p[j];
            qf[qb][2 * j] = v[0] * 0.17677669529663687;
            qf[qb][2 * j + 1] = v[1] * 0.17677669529663687;
        }
    }
    const int nblk = (nk + 15) >> 4;                    // 16-key blocks; this wave: blocks wave, wave + 4, ... (SOLO: all of them)
    // Buffer loads: a descriptor of this frame's keys (values) is built once in scalar registers, a block's offset into it is a scalar,
    // this lane's offset into the block a constant of the kernel - so a trip spends NO vector instruction on addresses (per-key pointers
    // clamped to the frame - add, min, 64-bit multiply-add - were 3 of them per block for K and 12 for V).  The scalar offset takes no
    // part in the descriptor's range check, so the last block of a ragged frame gets descriptors of its own that start at the block
    // (the k-step) and end with the frame: the keys it reaches beyond the frame load as zeros - their logits are masked below (logits()),
    // their value rows contribute 0 (the clamped loads multiplied a real row by e^-700).
    const int koff = (l15 * 384 + 8 * g) * (int)sizeof(double);
    const int voff = (g * 384 + 2 * l15) * (int)sizeof(double);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(kbase), 0, (nk - 1) * 3072 + 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(vbase), 0, (nk - 1) * 3072 + 256, 0x00020000);
    // K fragment of a block: dims 8 g .. 8 g + 7 of key 16 jb + l15 (A operand; dim 8 g + j at k-step j, as in qf)
    auto kload = [&](int jb, double (&kf)[8]) {
        f64x2 v[4];
        if (jb * 16 + 16 <= nk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(krs, koff, jb * (16 * 3072) + 16 * j, 0));
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(kbase + (size_t)jb * (16 * 384)), 0,
                                                                               (nk - 16 * jb - 1) * 3072 + 256, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs, koff + 16 * j, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { kf[2 * j] = v[j][0]; kf[2 * j + 1] = v[j][1]; }
    };
    // V fragments of a block: k-step s covers keys 16 jb + 4 s + g; this lane's A operand = dims 2 l15, 2 l15 + 1 of that key
    // (output dim blocks t = 0, 1: D register r of block t is dim 2 (g + 4 r) + t)
    auto vload = [&](int jb, f64x2 (&vf)[4]) {
        if (jb * 16 + 16 <= nk) {
#pragma unroll
            for (int s = 0; s < 4; ++s) vf[s] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, (jb * 16 + 4 * s) * 3072, 0));
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int left = nk - 16 * jb - 4 * s - 1;          // keys of the frame behind this step's first
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(vbase + (size_t)(jb * 16 + 4 * s) * 384), 0,
                                                                                   left < 0 ? 0 : left * 3072 + 256, 0x00020000);
                vf[s] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
            }
        }
    };
    // logits of one block for query block qb: S[r] = logit of key 16 jb + g + 4 r (pads: -inf)
    auto logits = [&](int jb, const double (&kf)[8], int qb) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = mfma64(kf[j], qf[qb][j], acc);
        int over = jb * 16 + 16 - nk;     // (> 0: the last block of a ragged frame only.  A scalar branch; the empty asm keeps the compiler from
        asm volatile("" : "+s"(over));    // turning it into eight selects in every trip - and from parking the condition in a vector register)
        if (over > 0) {
            int key;                      // (jb * 16 + g, computed HERE: as C++ the sum is hoisted in front of the branch, into every trip)
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(key) : "s"(jb * 16), "v"(g));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (key + 4 * r >= nk) acc[r] = -__builtin_inf();
        }
        return acc;
    };
    f64x4 O[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { O[qb][0] = f64x4{0.0, 0.0, 0.0, 0.0}; O[qb][1] = O[qb][0]; }
    double lsum[QB], mrun[QB], mthr[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { lsum[qb] = 0.0; mrun[qb] = -__builtin_inf(); mthr[qb] = -__builtin_inf(); }

    if (!TOPK) {
        // ---- full attention: online softmax over this wave's blocks ----
        // LAZY reference.  The exponentials of a row are taken against a reference mrun that only moves when a logit exceeds it by
        // more than TAU_LAZY (numerators up to e^8 instead of 1: harmless in fp64, and o / l does not depend on the reference).  The
        // test is lane-local - this lane's four logits against mrun + TAU_LAZY, one wave-uniform branch - so the common trip has
        // no cross-lane row maximum (two 64-bit shuffles through the LDS crossbar per query block, each waited for) and no
        // rescaling (an exponential and nine products per query block); after the first few blocks of a wave it is rarely taken.
        // (the next block's K fragments are requested into the SAME registers as soon as this block's Q K^T products are issued and
        // travel under the softmax and P V: a second register set copied at the top of the trip cost sixteen v_mov_b64 per block)
        double kf[8];
        f64x2 vf[4];
        constexpr int JS = SOLO ? 1 : 4;         // this wave's blocks: every one (SOLO), or wave, wave + 4, ...
        const int jb0 = SOLO ? 0 : wave;
        if (jb0 < nblk) kload(jb0, kf);
        for (int jb = jb0; jb < nblk; jb += JS) {
            vload(jb, vf);
            f64x4 Sq[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) Sq[qb] = logits(jb, kf, qb);
            if (jb + JS < nblk) kload(jb + JS, kf);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const f64x4 S = Sq[qb];
                const double lm = fmax(fmax(S[0], S[1]), fmax(S[2], S[3]));
                if (__any(lm > mthr[qb])) {
                    const double mnew = fmax(mrun[qb], quad_max(lm));
                    const double sc = exp_fast(mrun[qb] - mnew, sm.tab, ec);   // (rows that stay: exp(0) = 1 exactly; first block: 1e-304 x 0)
                    lsum[qb] *= sc;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) O[qb][t][r] *= sc;
                    mrun[qb] = mnew;
                    mthr[qb] = mnew + TAU_LAZY;
                }
                f64x4 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = exp_fast(S[r] - mrun[qb], sm.tab, ec);
                lsum[qb] += (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    O[qb][0] = mfma64(vf[s][0], p[s], O[qb][0]);
                    O[qb][1] = mfma64(vf[s][1], p[s], O[qb][1]);
                }
            }
        }
    } else {
        FT(0);
        // ---- dynamic attention, pass A: fp32 roundings of the logits -> LDS, row maxima ----
        f64x4 Sk[KEEP ? 8 : 1];
        if (KEEP) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int jb = wave + 4 * i;
                if (jb < nblk) {
                    double kf[8];
                    kload(jb, kf);
                    const f64x4 S = logits(jb, kf, 0);
                    Sk[i] = S;
                    mrun[0] = fmax(mrun[0], fmax(fmax(S[0], S[1]), fmax(S[2], S[3])));
                    float* row = sm.img + l15 * imgld + jb * 16 + g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[4 * r] = (float)S[r];
                }
            }
        } else {
            double kf[8];
            if (wave < nblk) kload(wave, kf);
            for (int jb = wave; jb < nblk; jb += 4) {
                f64x4 Sq[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) Sq[qb] = logits(jb, kf, qb);
                if (jb + 4 < nblk) kload(jb + 4, kf);          // (in place, under the stores below: see the full-attention loop)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const f64x4 S = Sq[qb];
                    mrun[qb] = fmax(mrun[qb], fmax(fmax(S[0], S[1]), fmax(S[2], S[3])));
                    float* row = sm.img + (qb * 16 + l15) * imgld + jb * 16 + g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[4 * r] = (float)S[r];
                }
            }
        }
        {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                mrun[qb] = quad_max(mrun[qb]);
                if (g == 0) sm.mw[wave * QT + qb * 16 + l15] = mrun[qb];
            }
            if (tid < QT) sm.lcount[tid] = 0;
        }
        FT(1);
        __syncthreads();
        FT(2);
        // ---- the exact k-th largest rounding of every row ----
        if (KEEP) {
            // at most 512 keys: the wave's four rows (4 wave .. 4 wave + 3) side by side, sixteen lanes each (row_search.hpp)
#ifndef F64_KO_SEARCH
            const float* rows = sm.img + 4 * wave * imgld;
            const bool own_space = imgld >= RQ_HIST_INTS;
            int* hb = own_space ? reinterpret_cast<int*>(sm.img + 4 * wave * imgld) : sm.hist + wave * 4 * RQ_HIST_INTS;
            const int hp = own_space ? imgld : RQ_HIST_INTS;
            const RowSearch r = nk <= 128 ? topk_quad_search<8>(rows, imgld, nk, a.topk, a.zq, lane, A_LIST, hb, hp)
                              : nk <= 256 ? topk_quad_search<16>(rows, imgld, nk, a.topk, a.zq, lane, A_LIST, hb, hp)
                                          : topk_quad_search<32>(rows, imgld, nk, a.topk, a.zq, lane, A_LIST, hb, hp);
            RowSel rs;
            if (a.topk >= nk || r.c_ge == a.topk) rs = RowSel{r.thr, 0, 0, 0};
            else if (r.c_ge - r.c_gt <= A_LIST) rs = RowSel{r.thr, 2, a.topk - r.c_gt, 0};
            else rs = RowSel{r.thr, 1, r.keylim, 0};
            if ((lane & 15) == 0) sm.sel[4 * wave + (lane >> 4)] = rs;
#else
            if (lane < 4) sm.sel[4 * wave + lane] = RowSel{0.f, 0, 0, 0};
#endif
        } else
        for (int q = wave; q < QT; q += 4) {
            const float* row = sm.img + q * imgld;
            int* hist = sm.hist + wave * RS_HIST_INTS;
#ifdef F64_KO_SEARCH
            if (lane == 0) sm.sel[q] = RowSel{0.f, 0, 0, 0};
            continue;
#endif
            const RowSel rs = nk <= 512 ? f64_row_select<8>(row, nk, a.topk, a.zq, lane, hist)
                            : nk <= 1024 ? f64_row_select<16>(row, nk, a.topk, a.zq, lane, hist) : f64_row_select<32>(row, nk, a.topk, a.zq, lane, hist);
            if (lane == 0) sm.sel[q] = rs;
        }
        FT(3);
        __syncthreads();
        FT(4);
        // ---- pass B: the fp64 logits again, masked softmax against the row maximum, P.V ----
        RowSel rs[QB];
        uint32_t* tap[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int q = qb * 16 + l15;
            rs[qb] = sm.sel[q];
            mrun[qb] = fmax(fmax(sm.mw[q], sm.mw[QT + q]), fmax(sm.mw[2 * QT + q], sm.mw[3 * QT + q]));
            tap[qb] = TAP ? a.sel + (((size_t)b * 4 + head) * P + q_off + min(q0 + q, nq - 1)) * a.selW : nullptr;
        }
        // one block of pass B for query block qb: classify, masked exponentials, P.V
        // Roundings tied at the k-th place (modes 1, 2) are rare - about one row in 60 000 on real-valued logits - and their handling is a
        // compare and a branch per logit: a wave none of whose rows has them (TIES = false: mode 0 everywhere, "keep every rounding
        // >= thr") runs a pass without it.  Knock-out: the tie branches were 9 % of the launch.
        auto pass_b = [&](int jb, const f64x4& S, const f64x2 (&vf)[4], int qb, auto ties) {
            constexpr bool TIES = decltype(ties)::value;
            f64x4 p;
            unsigned bits = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = jb * 16 + g + 4 * r;
                const float sf = (float)S[r];
                bool keep = TIES ? sf > rs[qb].thr : sf >= rs[qb].thr;
#ifdef F64_KO_TIES
                keep = sf >= rs[qb].thr;
                if (false) {
#else
                if (TIES && sf == rs[qb].thr) {
#endif
                    if (rs[qb].mode == 0) keep = true;
                    else if (rs[qb].mode == 1) keep = key <= rs[qb].aux;
                    else if (key < nk) {
                        const int q = qb * 16 + l15;
                        const int slot = atomicAdd(&sm.lcount[q], 1);
                        if (slot < A_LIST) { sm.lS[q * A_LIST + slot] = S[r]; sm.lkey[q * A_LIST + slot] = key; }
                    }
                }
                keep = keep && key < nk;
#ifdef F64_KO_EXP
                p[r] = keep ? S[r] - mrun[qb] : 0.0;
#else
                p[r] = keep ? exp_fast(S[r] - mrun[qb], sm.tab, ec) : 0.0;
#endif
                bits |= (unsigned)keep << (4 * r);        // keys 16 jb + g + 4 r
            }
            if (TAP && bits && q0 + qb * 16 + l15 < nq) atomicOr(tap[qb] + (jb >> 1), bits << (16 * (jb & 1) + g));
            lsum[qb] += (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                O[qb][0] = mfma64(vf[s][0], p[s], O[qb][0]);
                O[qb][1] = mfma64(vf[s][1], p[s], O[qb][1]);
            }
        };
        bool any_ties = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) any_ties |= rs[qb].mode != 0;
        any_ties = __any(any_ties);
        if (KEEP) {
            auto run = [&](auto ties) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int jb = wave + 4 * i;
                    if (jb < nblk) {
                        f64x2 vf[4];
                        vload(jb, vf);
                        pass_b(jb, Sk[i], vf, 0, ties);
                    }
                }
            };
            if (any_ties) run(std::true_type()); else run(std::false_type());
        } else {
            double kf[8];
            f64x2 vf[4];
            if (wave < nblk) kload(wave, kf);
            for (int jb = wave; jb < nblk; jb += 4) {
                vload(jb, vf);
                f64x4 Sq[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) Sq[qb] = logits(jb, kf, qb);
                if (jb + 4 < nblk) kload(jb + 4, kf);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) { if (any_ties) pass_b(jb, Sq[qb], vf, qb, std::true_type()); else pass_b(jb, Sq[qb], vf, qb, std::false_type()); }
            }
        }
    }

    if (TOPK) FT(5);
    if (SOLO) {
        // the wave's rows are complete: row sum over the row's four lanes, normalise, write (lane: dims 2 (g + 4 r), + 1 of query l15)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const double inv = 1.0 / quad_sum(lsum[qb]);
            const int q = q0 + qb * 16 + l15;
            f64x2 o[4];
            bool bad = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = f64x2{O[qb][0][r] * inv, O[qb][1][r] * inv};
                bad |= f64_out_of_range(o[r][0]) || f64_out_of_range(o[r][1]);
            }
            if (q < nq) {
                double* dst = a.msg + ((size_t)b * P + q_off + q) * 128 + head * 32 + 2 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<f64x2*>(dst + 8 * r) = o[r];
                if (bad) f64_raise(a.guard);
            }
        }
        return;
    }
    // ---- combine the four waves: row statistics and output partials through LDS ----
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const double l = quad_sum(lsum[qb]);
        const int q = qb * 16 + l15;
        if (g == 0) { sm.lw[wave * QT + q] = l; if (!TOPK) sm.mw[wave * QT + q] = mrun[qb]; }
        double* ob = sm.obuf + ((size_t)wave * QT + q) * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[2 * (g + 4 * r) + t] = O[qb][t][r];
    }
    __syncthreads();
    if (TOPK) {
        // rows with tied roundings at the k-th place: rank the listed candidates by fp64 value (then lower key), keep `aux`
        for (int q = wave; q < QT; q += 4) {
            const RowSel r = sm.sel[q];
            if (r.mode != 2) continue;
            const int n = min(sm.lcount[q], A_LIST);
            if (lane < n) {
                const double si = sm.lS[q * A_LIST + lane];
                const int ki = sm.lkey[q * A_LIST + lane];
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const double sj = sm.lS[q * A_LIST + j];
                    const int kj = sm.lkey[q * A_LIST + j];
                    rank += (sj > si) || (sj == si && kj < ki);
                }
                if (rank < r.aux) {
                    sm.lkey[q * A_LIST + lane] = ki | (int)0x80000000;
                    if (TAP && q0 + q < nq) atomicOr(a.sel + (((size_t)b * 4 + head) * P + q_off + q0 + q) * a.selW + (ki >> 5), 1u << (ki & 31));
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < QT * 16; e += 256) {
        const int q = e >> 4, d = (e & 15) * 2;
        if (q0 + q >= nq) continue;
        double m = -__builtin_inf();
#pragma unroll
        for (int w = 0; w < 4; ++w) m = fmax(m, sm.mw[w * QT + q]);
        double l = 0.0, o0 = 0.0, o1 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const double mwv = sm.mw[w * QT + q];
            const double f = TOPK ? 1.0 : (mwv == -__builtin_inf() ? 0.0 : exp_fast(mwv - m, sm.tab, ec));
            l += f * sm.lw[w * QT + q];
            o0 += f * sm.obuf[((size_t)w * QT + q) * 32 + d];
            o1 += f * sm.obuf[((size_t)w * QT + q) * 32 + d + 1];
        }
        if (TOPK && sm.sel[q].mode == 2) {
            const int n = min(sm.lcount[q], A_LIST);
            for (int j = 0; j < n; ++j) {
                const int kj = sm.lkey[q * A_LIST + j];
                if (kj >= 0) continue;
                const double p = exp_fast(sm.lS[q * A_LIST + j] - m, sm.tab, ec);
                const f64x2 v = *reinterpret_cast<const f64x2*>(vbase + (size_t)(kj & 0x7fffffff) * 384 + d);
                l += p; o0 += p * v[0]; o1 += p * v[1];
            }
        }
        if (TOPK) FT(6);
        const double inv = 1.0 / l;
        o0 *= inv; o1 *= inv;
        if (f64_out_of_range(o0) || f64_out_of_range(o1)) f64_raise(a.guard);
        *reinterpret_cast<f64x2*>(a.msg + ((size_t)b * P + q_off + q0 + q) * 128 + head * 32 + d) = f64x2{o0, o1};
    }
}

// ================================================================================================ small kernels
// Non-finite fp64 value, by its bits (this file is compiled with -fno-honor-nans: a floating-point test for NaN may be folded away).
// The reference lets NaN / inf inputs run through to NaN outputs; here ReLU would turn a NaN into 0 and the call would return
// plausible numbers - so the inputs (and the stream handed over to the fp32 kernels) are tested and the call is refused like
// any other out-of-range activation (mdgat_async_status: range_violation).
// (The values are LOADED as integers: applied to a floating-point value the same mask test is recognised as a class test and,
// under the flag, reduced to "is infinite" - measured: a NaN input went through unnoticed.)
__device__ __forceinline__ bool f64_bits_nonfinite(unsigned long long b) { return (b & 0x7ff0000000000000ull) == 0x7ff0000000000000ull; }
__device__ __forceinline__ bool f32_bits_nonfinite(unsigned b) { return (b & 0x7f800000u) == 0x7f800000u; }
// encoder inputs (mdgat.py:186-187, 154): in4 [R][4] = x y z saliency, in33 [R][33] = FPFH, rows pair-major (frame 0 then frame 1)
__global__ __launch_bounds__(256) void assemble_f64_kernel(const double* kpts0, const double* sigma0, const double* fpfh0, const double* kpts1,
                                                            const double* sigma1, const double* fpfh1, double* in4, double* in33, int B, int N, int M,
                                                            unsigned* guard) {
    const int P = N + M;
    const size_t total = (size_t)B * P * 37;
    bool bad = false;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t row = idx / 37;
        const int c = (int)(idx - row * 37);
        const int b = (int)(row / P), p = (int)(row - (size_t)b * P);
        const bool f1 = p >= N;
        const int n = f1 ? p - N : p, cnt = f1 ? M : N;
        const double* kp = f1 ? kpts1 : kpts0;
        const double* sg = f1 ? sigma1 : sigma0;
        const double* fp = f1 ? fpfh1 : fpfh0;
        const size_t r = (size_t)b * cnt + n;
        typedef const unsigned long long* bits_p;
        const unsigned long long v = c < 3 ? bits_p(kp)[r * 3 + c] : c == 3 ? bits_p(sg)[r] : bits_p(fp)[r * 33 + (c - 4)];
        bad |= f64_bits_nonfinite(v);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(c < 4 ? in4 + row * 4 + c : in33 + row * 33 + (c - 4));
        *dst = v;
    }
    if (bad) f64_raise(guard);
}

// The same from the loader's raw records [B][N][37] float32 = x y z saliency FPFH (load_data.py:146-165), one thread per keypoint.
// The reference normalises the FPFH row in FLOAT32 with numpy and widens afterwards (load_data.py:290-295:
// `np.linalg.norm(descs, axis=1)`, `np.multiply(descs, 1 / norm)`, `torch.tensor(..., dtype=torch.double)`); an input that
// differs from the reference's in its last float32 bit moves every logit by that much and flips near-tie top-k rows like an fp32
// network would.  So the arithmetic is numpy's to the bit: the squares rounded one by one, their sum in the order of numpy's
// pairwise reduction for 8 <= n <= 128 (eight strided partial sums, combined as a tree, the tail added last), correctly rounded
// square root, reciprocal and products, nothing contracted into an FMA.  tests/test_oracle_golden.py pins this order against the
// reference loader's own outputs; tests/test_gpu_f64.py asks for a bit-identical Z.
__global__ __launch_bounds__(256) void assemble_frames_f64_kernel(const float* rec0, const float* rec1, int normalize, double* in4, double* in33,
                                                                   int B, int N, int M, unsigned* guard) {
    // (every operation rounded by itself: no product contracted into the sum behind it.  Plain operators - HIP's __fmul_rn /
    // __fadd_rn ARE plain operators in this toolchain and __fsqrt_rn is the native approximation; `/` and __builtin_sqrtf are
    // correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt)
#pragma clang fp contract(off)
    const int P = N + M;
    const size_t rows = (size_t)B * P;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (size_t)gridDim.x * 256) {
        const int b = (int)(row / P), p = (int)(row - (size_t)b * P);
        const bool f1 = p >= N;
        const float* rec = (f1 ? rec1 + ((size_t)b * M + (p - N)) * 37 : rec0 + ((size_t)b * N + p) * 37);
        float v[37];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 37; ++c) {
            const unsigned b = reinterpret_cast<const unsigned*>(rec)[c];
            bad |= f32_bits_nonfinite(b);
            v[c] = __builtin_bit_cast(float, b);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) in4[row * 4 + c] = (double)v[c];
        float inv = 1.f;
        if (normalize) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = v[4 + j] * v[4 + j];
#pragma unroll
            for (int i = 8; i < 32; i += 8)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float q = v[4 + i + j] * v[4 + i + j]; r[j] = r[j] + q; }
            float sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            const float q = v[36] * v[36];
            sum = sum + q;
            inv = 1.f / __builtin_sqrtf(sum);
        }
        bad |= !(inv < 3.0e38f);                 // (a row of zeros: 1 / 0 - NaN descriptors in the reference)
#pragma unroll
        for (int c = 0; c < 33; ++c) in33[row * 33 + c] = (double)(normalize ? v[4 + c] * inv : v[4 + c]);
        if (bad) f64_raise(guard);
    }
}

__global__ __launch_bounds__(256) void f64_to_f32_kernel(const double* in, float* out, size_t n, unsigned* guard) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long b = reinterpret_cast<const unsigned long long*>(in)[i];
        bad |= f64_bits_nonfinite(b);
        out[i] = (float)__builtin_bit_cast(double, b);
    }
    if (bad) f64_raise(guard);
}

// measurement only: what v_mfma_f64_16x16x4_f64 sustains (two waves per SIMD, operands in registers, four accumulator chains)
__global__ __launch_bounds__(512, 2) void mfma64_probe_kernel(const double* src, double* sink, long long* ticks, int reps) {
    const int tid = threadIdx.x;
    double x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = src[tid * 16 + i]; y[i] = src[tid * 16 + 8 + i]; }
    f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0, c2 = c0, c3 = c0;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            asm volatile("" : "+v"(x[i]));
            c0 = mfma64(x[i], y[i], c0);
            c1 = mfma64(x[i + 1], y[i], c1);
            c2 = mfma64(x[i], y[i + 1], c2);
            c3 = mfma64(x[i + 1], y[i + 1], c3);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const f64x4 c = (c0 + c1) + (c2 + c3);
    const double s = (c[0] + c[1]) + (c[2] + c[3]);
    if (s == 123.456) sink[tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void mfma64_probe_fill(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned s = (unsigned)i * 1664525u + 1013904223u;
        s = s * 1664525u + 1013904223u;
        p[i] = ((int)(s >> 12) % 2000001 - 1000000) * 1e-6;
    }
}

}  // namespace

// ================================================================================================ launchers
// compute units of the current device (asked once per device)
static int f64_cu_count() {
    static std::atomic<int> cached[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

int launch_gemm_f64(const GemmF64Args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return MDGAT_OK;
    static std::atomic<unsigned long long> done2{0}, done4{0}, done2f{0}, done4f{0}, done2d{0};
    // Tile width.  64 x 128 tiles do a third more arithmetic per byte staged through LDS; 64 x 64 tiles are twice as many workgroups
    // (four resident per CU instead of three).  Up to a few rounds of workgroups the launch is bound by how evenly it fills the
    // CUs, not by its inner loop: one pair of 512 keypoints is 16 row tiles - 48 wide workgroups on 256 CUs for the q|k|v product,
    // 96 narrow ones, 20 -> 17 us; 8192 rows x 128 outputs 31 -> 21 us (profiles/NOTES_r5.md section 9).  The narrow tile is taken
    // when its rounds of resident workgroups, priced at 0.55 of a wide round, come out below the wide tile's.
    static const int wn_env = [] { const char* e = getenv("MDGAT_F64_GEMM_WN"); return e ? atoi(e) : 0; }();      // (measurements)
    int wn = 2;
    if (a.N > 64) {
        const int cus = f64_cu_count();
        const long tiles_m64 = (long)((a.M + G_BM - 1) / G_BM) * (a.batch > 1 ? a.batch : 1);
        const long tw = tiles_m64 * ((a.N + 127) / 128), tn = tiles_m64 * ((a.N + 63) / 64);
        const long rw = (tw + 3 * cus - 1) / (3 * cus), rn = (tn + 4 * cus - 1) / (4 * cus);
        wn = (tw <= 6L * cus && rn * 55 < rw * 100) ? 2 : 4;
        if (wn_env == 2 || wn_env == 4) wn = wn_env;
    }
    const int bn = 32 * wn;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool fast = a.M % G_BM == 0 && a.N % bn == 0 && a.K % G_KC == 0 && a.K0 % G_KC == 0 && a.lda0 % 2 == 0 && a.ldw % 2 == 0 && al16(a.A0) && al16(a.W) &&
                      (a.K0 >= a.K || (a.lda1 % 2 == 0 && al16(a.A1))) && (a.batch <= 1 || (a.sA % 2 == 0 && a.sW % 2 == 0));
    // (chunks of 64 for whole-tile launches of at most one 64 x 64 tile per CU: one or two pairs of 512 keypoints - 19.0 -> 17.8 us
    // at K = 256, the one-pair forward 1.54 -> 1.48 ms; from eight pairs on the shallower chunks' third resident workgroup wins)
    static const bool deep_off = [] { const char* e = getenv("MDGAT_F64_GEMM_DEEP"); return e && atoi(e) == 0; }();      // (measurements)
    const bool deep = !deep_off && fast && wn == 2 && a.K % 64 == 0 && (a.K0 >= a.K || a.K0 % 64 == 0) &&
                      (long)((a.M + G_BM - 1) / G_BM) * ((a.N + 63) / 64) * (a.batch > 1 ? a.batch : 1) <= (long)f64_cu_count();
    const size_t lds = (size_t)(G_BM + bn) * ((deep ? 64 : G_KC) + 2) * sizeof(double);
    const dim3 grid((a.M + G_BM - 1) / G_BM, (a.N + bn - 1) / bn, a.batch > 1 ? a.batch : 1);
    auto go = [&](auto kern, std::atomic<unsigned long long>& done) -> int {
        if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(kern), lds, done, "gemm_f64 LDS")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return MDGAT_OK;
    };
    int rc;
    if (wn == 4) rc = fast ? go(gemm_f64_kernel<4, true>, done4f) : go(gemm_f64_kernel<4, false>, done4);
    else if (deep) rc = go(gemm_f64_kernel<2, true, 64>, done2d);
    else rc = fast ? go(gemm_f64_kernel<2, true>, done2f) : go(gemm_f64_kernel<2, false>, done2);
    if (rc) return rc;
    return mdgat_check_hip(hipGetLastError(), "gemm_f64 launch");
}

// upper-tail standard normal quantile (only seeds the threshold search)
static float f64_normal_quantile_upper(double p) {
    if (p <= 0.0) return 8.f;
    if (p >= 1.0) return -8.f;
    // Abramowitz-Stegun 26.2.23 (|error| < 4.5e-4)
    const bool upper = p < 0.5;
    const double pp = upper ? p : 1.0 - p;
    const double t = sqrt(-2.0 * log(pp));
    const double x = t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
    return (float)(upper ? x : -x);
}

// full attention: workgroups (of four 32-query waves) per CU from which the one-wave-per-query-block form is launched
constexpr long F64_SOLO_MIN_WG_PER_CU = 4;
// -1: by launch size (default); 0: always the keys of a query tile split over the four waves (what a pair returns is then bit-identical
// whatever batch it travels in); 1: always one wave per 32 queries.  MDGAT_F64_ATTENTION_FORM in the environment.
static std::atomic<int> g_attention_form{-2};
int f64_attention_form() {
    const int v = g_attention_form.load(std::memory_order_relaxed);
    if (v != -2) return v;
    static const int env = [] { const char* e = getenv("MDGAT_F64_ATTENTION_FORM"); const int m = e ? atoi(e) : -1; return m == 0 || m == 1 ? m : -1; }();
    return env;
}
extern "C" int mdgat_set_f64_attention_form(int mode) {
    const int prev = f64_attention_form();
    g_attention_form.store(mode == 0 || mode == 1 ? mode : mode == -1 ? -1 : -2, std::memory_order_relaxed);
    return prev;
}

int launch_attention_f64(int B, int N, int M, int cross, int topk, const double* qkv, double* msg, uint32_t* sel, hipStream_t s, unsigned* guard) {
    if (B <= 0 || N <= 0 || M <= 0) return MDGAT_OK;
    const int nk_max = N > M ? N : M, nk_min = N < M ? N : M;
    if (topk > nk_min) {   // torch.topk raises (mdgat.py:202)
        mdgat_set_error("dynamic attention: k=%d exceeds the number of keys (%d)", topk, nk_min);
        return MDGAT_ERR_BAD_ARG;
    }
    if (nk_max > 2048 && topk > 0) {
        mdgat_set_error("dynamic attention (fp64): %d keys per frame > 2048 supported", nk_max);
        return MDGAT_ERR_UNSUPPORTED;
    }
    const bool dyn = topk > 0 && !(topk == N && topk == M);
    AttnF64Args a{};
    a.qkv = qkv; a.msg = msg; a.N = N; a.M = M; a.cross = cross; a.topk = dyn ? topk : 0;
    a.zq = dyn ? f64_normal_quantile_upper(((double)topk - 0.5) / (double)nk_max) : 0.f;
    a.selW = (nk_max + 31) / 32;
    a.sel = nullptr;
    a.units = B * 2 * MDGAT_HEADS;
    a.guard = guard;
    if (sel && topk > 0) {
        if (int rc = mdgat_check_hip(hipMemsetAsync(sel, dyn ? 0 : 0xff, mdgat_topk_sel_words(B, N, M) * sizeof(uint32_t), s), "memset(top-k tap)")) return rc;
        if (dyn) a.sel = sel;
    }
    const int ugroups = (a.units + 7) / 8;
    // (the LDS opt-in once per instantiation and device, at the largest size the instantiation is ever launched with: a limit that
    // followed each launch's key count raced between host threads with different keypoint counts, and cost a driver call per launch.
    // `cap` = that largest key count; the tag gives every call site its own instantiation of this lambda, hence its own `done` mask.)
    auto go = [&](auto kern, int QT, bool tk, int cap, auto tag) -> int {
        (void)tag;
        static std::atomic<unsigned long long> done{0};
        a.hist_ints = tk ? attn_hist_ints(nk_min, nk_max) : 0;
        const bool solo = QT > 32;        // (QT = the queries of a workgroup; the one-wave-per-query-block form keeps only the exponential's table in LDS)
        const size_t lds = solo ? 256 * sizeof(double) : attn_lds_bytes(QT, nk_max, tk, a.hist_ints);
        if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(kern), solo ? lds : attn_lds_bytes(QT, cap, tk, tk ? (cap > 512 ? 4 * RS_HIST_INTS : 16 * RQ_HIST_INTS) : 0), done, "attention_f64 LDS")) return rc;
        a.tiles = (nk_max + QT - 1) / QT;
        hipLaunchKernelGGL(kern, dim3(8 * a.tiles * ugroups), dim3(256), lds, s, a);
        return mdgat_check_hip(hipGetLastError(), "attention_f64 launch");
    };
    // full attention: 32 queries per workgroup (half the key traffic per query).  Dynamic attention: 16 - at most 512 keys the
    // fp64 logits stay in registers between the passes (KEEP), beyond that the rounding images of 32 rows would not fit the LDS
    if (!dyn) {
        // (fewer than two workgroups per CU with 32 queries each - a pair or two of 512 keypoints: 16 queries per workgroup, the
        // same arithmetic per row; MDGAT_F64_ATT_QB=1|2 forces one for measurements.  64 queries per workgroup - 234 registers, two
        // waves per SIMD - lose: 236 -> 289 us at batch 32)
        static const int qb_env = [] { const char* e = getenv("MDGAT_F64_ATT_QB"); return e ? atoi(e) : 0; }();
        // one wave per 32 queries (SOLO) from F64_SOLO_MIN_WG_PER_CU workgroups of four such waves per CU on; mdgat_set_f64_attention_form(1)
        // forces it at every size (tests: ragged frames, one pair), (0) never
        const int form = f64_attention_form();
        const long solo_wgs = 8L * ((nk_max + 127) / 128) * ugroups;
        const bool solo = qb_env == 0 && (form == 1 || (form != 0 && solo_wgs >= F64_SOLO_MIN_WG_PER_CU * (long)f64_cu_count()));
        if (solo) return go(attention_f64_kernel<false, 2, false, false, true>, 128, false, 0, std::integral_constant<int, 6>());
        const bool small = 8L * ((nk_max + 31) / 32) * ugroups < 2L * f64_cu_count();
        if (qb_env == 1 || (qb_env != 2 && small)) return go(attention_f64_kernel<false, 1, false>, 16, false, 0, std::integral_constant<int, 0>());
        return go(attention_f64_kernel<false, 2, false>, 32, false, 0, std::integral_constant<int, 1>());
    }
    if (nk_max <= 512) return a.sel ? go(attention_f64_kernel<true, 1, true, true>, 16, true, 512, std::integral_constant<int, 2>())
                                    : go(attention_f64_kernel<true, 1, false, true>, 16, true, 512, std::integral_constant<int, 3>());
    return a.sel ? go(attention_f64_kernel<true, 1, true>, 16, true, 2048, std::integral_constant<int, 4>())
                 : go(attention_f64_kernel<true, 1, false>, 16, true, 2048, std::integral_constant<int, 5>());
}

int launch_assemble_f64(int B, int N, int M, const double* kpts0, const double* sigma0, const double* fpfh0, const double* kpts1,
                        const double* sigma1, const double* fpfh1, double* in4, double* in33, unsigned* guard, hipStream_t s) {
    const size_t total = (size_t)B * (N + M) * 37;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(assemble_f64_kernel, dim3(blocks), dim3(256), 0, s, kpts0, sigma0, fpfh0, kpts1, sigma1, fpfh1, in4, in33, B, N, M, guard);
    return mdgat_check_hip(hipGetLastError(), "assemble_f64 launch");
}

int launch_assemble_frames_f64(int B, int N, int M, const float* rec0, const float* rec1, int normalize, double* in4, double* in33, unsigned* guard,
                               hipStream_t s) {
    const size_t rows = (size_t)B * (N + M);
    if (!rows) return MDGAT_OK;
    const int blocks = (int)((rows + 255) / 256 < 8192 ? (rows + 255) / 256 : 8192);
    hipLaunchKernelGGL(assemble_frames_f64_kernel, dim3(blocks), dim3(256), 0, s, rec0, rec1, normalize, in4, in33, B, N, M, guard);
    return mdgat_check_hip(hipGetLastError(), "assemble_frames_f64 launch");
}

int launch_f64_to_f32(const double* in, float* out, size_t n, unsigned* guard, hipStream_t s) {
    if (!n) return MDGAT_OK;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(blocks), dim3(256), 0, s, in, out, n, guard);
    return mdgat_check_hip(hipGetLastError(), "f64_to_f32 launch");
}

extern "C" int mdgat_mfma_f64_probe(int reps, void* workspace, size_t workspace_bytes, float* ms_out, double* flops_out,
                                    long long* ticks_out, void* stream) {
    const size_t need = (size_t)512 * 16 * 8 + 512 * 8 + 256;
    if (reps <= 0 || !workspace || workspace_bytes < need || !ms_out || !flops_out || !ticks_out) {
        mdgat_set_error("mdgat_mfma_f64_probe: bad argument (workspace >= %zu bytes)", need);
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int dev = 0, num_cu = 0;
    if (int rc = mdgat_check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (int rc = mdgat_check_hip(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
    double* src = static_cast<double*>(workspace);
    double* sink = src + 512 * 16;
    long long* ticks = reinterpret_cast<long long*>(sink + 512);
    hipLaunchKernelGGL(mfma64_probe_fill, dim3(32), dim3(256), 0, s, src, 512 * 16);
    const int grid = num_cu * 2;
    hipLaunchKernelGGL(mfma64_probe_kernel, dim3(grid), dim3(512), 0, s, src, sink, ticks, 8);      // warm-up
    hipEvent_t e0, e1;
    if (int rc = mdgat_check_hip(hipEventCreate(&e0), "hipEventCreate")) return rc;
    if (int rc = mdgat_check_hip(hipEventCreate(&e1), "hipEventCreate")) { (void)hipEventDestroy(e0); return rc; }
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfma64_probe_kernel, dim3(grid), dim3(512), 0, s, src, sink, ticks, reps);
    (void)hipEventRecord(e1, s);
    int rc = mdgat_check_hip(hipEventSynchronize(e1), "mfma64 probe");
    float ms = 0.f;
    if (!rc) rc = mdgat_check_hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
    long long t = 0;
    if (!rc) rc = mdgat_check_hip(hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost), "hipMemcpy(ticks)");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    *ms_out = ms;
    *flops_out = (double)grid * 8.0 /* waves */ * (double)reps * 16.0 /* MFMAs per rep */ * 2048.0;
    *ticks_out = t;
    return MDGAT_OK;
}
