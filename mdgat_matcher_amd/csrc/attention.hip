// Multi-head scaled-dot-product attention over keypoints, full (mdgat.py:190-194, attention) and
// dynamic (mdgat.py:196-210, dynamic_attention: keep the k largest logits of every row, softmax over
// those, zero elsewhere).  The N x M logits / probabilities are never written to memory.
//
// Arithmetic: fp32-equivalent products on the f16 matrix cores.  Every operand x is carried as two
// halves x = hi + lo (hi = f16(x), lo = f16(x - hi), the residual UNSCALED - f16 denormals included, which the
// matrix cores honour: common.hpp; 22 mantissa bits for operands of order one) and a product is three
// v_mfma_f32_32x32x16_f16: hi.hi, hi.lo and lo.hi.  The dropped lo.lo term is 2^-22 relative - the same class as
// the rounding of an fp32 FMA chain (tools/precision_probe.py: max|dZ| 1.3e-5 either way) at 16/3 times
// the rate of v_mfma_f32_32x32x2_f32.  (The probabilities are carried times 2048 so that their residual is a normal f16.)
//
// gfx950 mapping.  One workgroup = 4 waves owns one (pair, frame, head): the head's K rows (hi | lo
// halves, 144-byte padded rows) and V^T rows (keys contiguous, per plane) for all <= 512 keys are
// staged once in LDS (140 KB) and the workgroup loops over its 128-query tiles; a wave owns 32
// queries and computes S^T = K Q^T ("swapped" product): in the 32x32 C/D fragment layout a lane then
// holds, for ITS query (lane & 31), 16 logits per 32-key block - the row of a query lives in two lanes
// (l, l+32) x 16 registers per block, up to 256 registers for 512 keys (one wave per SIMD, 512-register
// budget).  Row max, row sum and the top-k count are register-local plus ONE lane^32 exchange.  The K
// rows are fed to the MFMA in a permuted order (bits 2 and 3 of the row index swapped) so that the 8
// accumulator registers of one half-block are 8 CONSECUTIVE keys: the probabilities, split to f16 in
// place, are then already the A operand of the P.V product whose B operand is one ds_read_b128 of V^T.
//
// Top-k: the exact k-th largest logit of a row is found by a per-row bracketing search on the
// threshold value t (count(s >= t) is monotone), see topk_threshold().  Masked softmax over "s >= t"
// equals softmax over the gathered top-k; exact ties at the k-th value are all kept (torch.topk would
// keep an arbitrary subset of them).
#include "common.hpp"
#include <cstdlib>

#ifdef WIDE_TRACE
__device__ long long g_wide_trace[4 * 8];
extern "C" int mdgat_wide_trace_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_trace), n * sizeof(long long)); }
#define WT(k) do { if (blockIdx.x == 8 * 5 + 3 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7)) \
    g_wide_trace[((threadIdx.x >> 6) == 7) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WT(k) do {} while (0)
#endif
namespace {

constexpr int KROWH = 72;   // K row in LDS, halves: 32 hi | 32 lo | 8 pad (144 B = 9 x 16 B: conflict free)

struct AttnArgs {
    const _Float16* q16;   // [B][P][4][2][32]   pre-scaled by log2(e)/sqrt(32)
    const _Float16* k16;   // [B][P][4][2][32]
    const _Float16* vt16;  // [B][4][2][32][PP]  keys of frame 0 at columns [0, N), frame 1 at [Npad, Npad + M); pads zero
    float* msg;            // [B][P][128]
    int N, M, Npad, PP, cross, topk;
    float zq;              // standard-normal quantile of the top-k fraction (first probe of the threshold search)
    uint32_t* sel;         // TAP kernels only: [B][4][P][selW] bit j of word w = key 32 w + j of the source frame was kept
    int selW;
    // XCD-aware 1-D grids (attention_topk16_kernel, attention_topk_wide_kernel): workgroup blockIdx.x = (ugroup * grid_split + part) * 8 + xcd
    // works on part `part` of the query tiles of unit ugroup * 8 + xcd, unit = (pair * 2 + frame) * 4 + head: the parts of a unit share
    // an XCD (round-robin dispatch), so its K and V^T are fetched into one L2 only.  The grid is padded to 8 units.
    int grid_split, grid_units;
};

// parity tap (mdgat_taps.topk_sel): OR `bits` (NB consecutive keys starting at key0, NB | 32) into the row's mask
__device__ __forceinline__ void tap_keys(uint32_t* row, int key0, unsigned bits) {
    if (bits) atomicOr(row + (key0 >> 5), bits << (key0 & 31));
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ int xor32i(int v) { return __shfl_xor(v, 32, 64); }

// x -> (f16 hi, f16 residual), unscaled residual: used for the probabilities, which are carried times
// 2048 so that the residual of every value that matters is a normal f16
__device__ __forceinline__ void split8(const float (&p)[8], f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)p[j];
    // residual p - (float)h straight from the packed halves: one v_fma_mix_f32 per value instead of a conversion
    // back and a subtraction (the kernels that use this are bound by their vector instruction count).
    // (Measured and dropped: v_fma_mixlo_f16 / v_fma_mixhi_f16 writing the residual halves directly - 8 instead of 12
    // instructions per 8 values, but a 16-bit destination write needs wait states the hazard recogniser does not supply
    // for inline asm (wrong residuals in some kernel instances), and as one fenced block the scheduler cannot interleave
    // it: attention_topk16_kernel 154 -> 161 us, attention_stream_kernel 65.5 -> 66.7 us.)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 hp = __builtin_bit_cast(u32x4, h);
    float r[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r[2 * j]) : "v"(p[2 * j]), "v"(hp[j]));
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[2 * j + 1]) : "v"(p[2 * j + 1]), "v"(hp[j]));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = (_Float16)r[j];
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Logits of magnitude below 2^-70 count as ZERO in every comparison of the top-k selection ("tied at zero"): a threshold t
// in that zone stands for T0 = the float above -2^-70, and "s >= T0" is exactly "s is zero, tiny or positive".  Real-valued
// data has no such logits other than exact zeros (zero key vectors); the rule exists so that the packed indicator below is
// EXACT for every float - it cannot resolve differences below 2^-100.  Every threshold is canonicalised before it is
// compared with anything (canon_thr), ties at a canonical T0 are the whole zone (tie_top).
#define MDGAT_TINY 0x1p-70f
#define MDGAT_T0 (-0x1.fffffep-71f)
__device__ __forceinline__ float canon_thr(float t) { return fabsf(t) < MDGAT_TINY ? MDGAT_T0 : t; }
// the logits tied at the canonical threshold thr are those in [thr, tie_top(thr)]
__device__ __forceinline__ float tie_top(float thr) { return thr == MDGAT_T0 ? 0x1.fffffep-71f : thr; }
// "s >= t" for two logits per instruction, as a number: clamp((s - t') 2^100) = 1 or 0, with t' the float just below the
// canonical t (the product is exact inside the FMA; |t| >= 2^-70 or t = T0, so every s >= t gives at least ulp(t) 2^100 >= 1
// and every s <= t' at most 0).  Compare / select / add-carry cost two instructions per logit and a lane mask in scalar
// registers each (the pass over a row ran out of them and spilled); the indicator costs half an instruction, and counting
// it another half (v_pk_add).  ge_const(t), t canonical, is the addend of the FMA; -inf pads give 0; t = -inf keeps
// every finite logit.
#define MDGAT_GE_BIG 1.2676506002282294e30f      // 2^100
__device__ __forceinline__ float ge_const(float t) {
    const int b = __builtin_bit_cast(int, t);
    const int bp = t > 0.f ? b - 1 : (b | (int)0x80000000) + 1;
    const float tp = __builtin_bit_cast(float, bp);
    return t == -__builtin_inff() ? 3.0e38f : -tp * MDGAT_GE_BIG;
}
__device__ __forceinline__ f32x2 ge_ind(f32x2 s, float c) {
    f32x2 d;
    const f32x2 c2 = {c, c};
    const unsigned long long big = 0x7180000071800000ull;       // {2^100, 2^100} in a scalar register pair
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(s), "s"(big), "v"(c2));
    return d;
}
// p[j] = exp2(s[j] - m11), zero below the threshold (dynamic layers; gc = ge_const(threshold), `kept` counts the logits
// at or above it in two packed halves); the row sum in two packed halves
// PK = false (the kernels whose row does not sit in vector registers proper: attention_topk_wide_kernel, the 512-logit
// instance of attention_kernel): gc is the threshold itself, compare and select.
template <bool TOPK, bool PK = true>
__device__ __forceinline__ void softmax8(const float* s, float m11, float gc, float (&p)[8], f32x2& l2, f32x2& kept) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 s2 = {s[j], s[j + 1]};
        const f32x2 d = s2 - f32x2{m11, m11};
        f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
        if (TOPK && !PK) {
            e[0] = (s[j] >= gc) ? e[0] : 0.f; e[1] = (s[j + 1] >= gc) ? e[1] : 0.f;
            kept[0] += (s[j] >= gc) ? 1.f : 0.f; kept[1] += (s[j + 1] >= gc) ? 1.f : 0.f;     // (dead code where the caller ignores it)
        } else if (TOPK) {
            const f32x2 ind = ge_ind(s2, gc);
            e *= ind;
            kept += ind;
        }
        p[j] = e[0]; p[j + 1] = e[1];
        l2 += e;
    }
}
__device__ __forceinline__ int kept_count(f32x2 kept) { return (int)(kept[0] + kept[1] + 0.5f); }

// How the two lanes of a row (and, in the split-key kernel, the two waves of a row) combine per-row
// partial results.  WaveComm: the row lives in one wave (lanes l, l ^ 32).
struct WaveComm {
    static constexpr bool LOCAL_VOTE = true;     // any() is one wave-level ballot
    static constexpr bool PACKED_COUNT = true;   // topk_threshold counts with ge_ind()
    static constexpr bool LEAN_SEARCH = false;   // (see WideComm)
    __device__ __forceinline__ float rsum(float v) { return v + xor32(v); }
    __device__ __forceinline__ int rsum(int v) { return v + xor32i(v); }
    __device__ __forceinline__ float rmin(float v) { return fminf(v, xor32(v)); }
    __device__ __forceinline__ float rmax(float v) { return fmaxf(v, xor32(v)); }
    __device__ __forceinline__ bool any(bool p) { return __any(p); }
    __device__ __forceinline__ void stats(float& mn, float& sum, float& sq) {
        mn = fminf(mn, xor32(mn)); sum += xor32(sum); sq += xor32(sq);
    }
    __device__ __forceinline__ void stats4(float&, float& mn, float& sum, float& sq) { stats(mn, sum, sq); }     // (LEAN_SEARCH comms only)
    // row count + "is any row of the voting domain still probing" in one step
    __device__ __forceinline__ int count_vote(int c, bool probing, bool& any_probing) {
        any_probing = __any(probing);
        return c + xor32i(c);
    }
};
// QuadComm: the row lives in four lanes of one wave (l & 15 = query; 16x16 MFMA fragments).
// The partners of a lane are lane ^ 16 and lane ^ 32: v_permlane16_swap / v_permlane32_swap on two copies of the value bring
// them in on the vector ALU (x = y = v; after the swap x holds the even 16-lane rows' values and y the odd rows' - for every lane
// the pair (x, y) is (own value, partner's) in some order), instead of a ds_bpermute round trip through the LDS per step.
// (inline asm: the clang builtins of this ROCm return the first result twice; the compiler inserts no wait states around an asm)
template <typename T, typename Op> __device__ __forceinline__ T quad_reduce(T v, Op op) {
    static_assert(sizeof(T) == 4, "32-bit values");
    unsigned x = __builtin_bit_cast(unsigned, v), y = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    v = op(__builtin_bit_cast(T, x), __builtin_bit_cast(T, y));
    x = __builtin_bit_cast(unsigned, v); y = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return op(__builtin_bit_cast(T, x), __builtin_bit_cast(T, y));
}
struct QuadComm {
    static constexpr bool LOCAL_VOTE = true;
    static constexpr bool PACKED_COUNT = true;
    static constexpr bool LEAN_SEARCH = false;
    __device__ __forceinline__ float rsum(float v) { return quad_reduce(v, [](float a, float b) { return a + b; }); }
    __device__ __forceinline__ int rsum(int v) { return quad_reduce(v, [](int a, int b) { return a + b; }); }
    __device__ __forceinline__ float rmin(float v) { return quad_reduce(v, [](float a, float b) { return fminf(a, b); }); }
    __device__ __forceinline__ float rmax(float v) { return quad_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }
    __device__ __forceinline__ bool any(bool p) { return __any(p); }
    __device__ __forceinline__ void stats(float& mn, float& sum, float& sq) { mn = rmin(mn); sum = rsum(sum); sq = rsum(sq); }
    __device__ __forceinline__ void stats4(float&, float& mn, float& sum, float& sq) { stats(mn, sum, sq); }
    __device__ __forceinline__ int count_vote(int c, bool probing, bool& any_probing) {
        any_probing = __any(probing);
        return rsum(c);
    }
};
// Exact k-th largest logit of every row (this wave holds 16 NBLK logits of the row per lane; Comm combines
// the partials of a row).  Probe thresholds t, count(s >= t) per row, keep a bracket lo < thr <= hv with counts
// clo > k > chi.  The first probe is the normal quantile of the row (mean + zq * sd); later probes step by
// (count - k) / density with the normal density at the probe, falling back to interpolation inside the
// bracket and to its midpoint.  A row stops probing when count == k, when no float lies strictly inside
// the bracket (exact ties at the k-th value: all kept), or when it is one element away from k on either
// side; those rows are finished by direct order-statistic passes after the loop (max below hv / second
// smallest at or above lo).  All rows run in lockstep, so the loop ends with its slowest row.
// More than k logits are >= the returned threshold only when the k-th place falls inside a group of exactly equal
// logits (fp32 logits closer than one ulp, duplicated keypoints); the kernels count what the softmax pass keeps and
// call topk_break_ties() for such rows, so that every row keeps exactly k keys like torch.topk.
template <int NBLK, bool EXACT, typename Comm>
__device__ __forceinline__ float topk_threshold(const f32x16 (&S)[NBLK], float& m, int k, int nk, float zq, Comm& comm, bool whole = false) {
    const float INF = __builtin_inff();
    // the packed indicator form needs the logits in vector registers proper: not the 256-logit instance (half of its
    // row lives in accumulation registers) and not the split-key kernel (no register to spare)
    constexpr bool PACKED = Comm::PACKED_COUNT && NBLK <= 8;
    constexpr bool LEAN = EXACT || Comm::LEAN_SEARCH;
    float smin = INF, sum = 0.f, sq = 0.f;
    float mu, sd;
    if (EXACT) {
        // every logit is finite; mean and deviation only seed the search: every 4th register is enough for them
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float s = S[jb][r];
                smin = fminf(fminf(smin, s), fminf(fminf(S[jb][r + 1], S[jb][r + 2]), S[jb][r + 3]));
                sum += s;
                sq = fmaf(s, s, sq);
            }
        comm.stats(smin, sum, sq);
        const float inv_n = 4.0f / (float)nk;
        mu = sum * inv_n;
        sd = sqrtf(fmaxf(sq * inv_n - mu * mu, 1e-12f));
    } else if (Comm::LEAN_SEARCH && whole) {
        // (LEAN_SEARCH, a row without pads - `whole`, the same for the whole workgroup: as above, and the row maximum - `m` arrives
        // as this wave's part of it - travels with the three statistics in ONE exchange instead of four)
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float s = S[jb][r];
                smin = fminf(fminf(smin, s), fminf(fminf(S[jb][r + 1], S[jb][r + 2]), S[jb][r + 3]));
                sum += s;
                sq = fmaf(s, s, sq);
            }
        comm.stats4(m, smin, sum, sq);
        const float inv_n = 4.0f / (float)nk;
        mu = sum * inv_n;
        sd = sqrtf(fmaxf(sq * inv_n - mu * mu, 1e-12f));
    } else {
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = S[jb][r];
                smin = fminf(smin, s == -INF ? INF : s);
                s = (s == -INF) ? 0.f : s;
                sum += s;
                sq = fmaf(s, s, sq);
            }
        if (Comm::LEAN_SEARCH) comm.stats4(m, smin, sum, sq);
        else comm.stats(smin, sum, sq);
        const float inv_n = 1.0f / (float)nk;
        mu = sum * inv_n;
        sd = sqrtf(fmaxf(sq * inv_n - mu * mu, 1e-12f));
    }
    const float inv_sd = 1.0f / sd;
    // count(s >= t): the indicator of ge_ind() summed in four independent packed accumulators, one instruction per
    // logit in all (was two: v_sub + a sign bit shifted into a register by v_alignbit)
    auto count_local = [&](float t) {
        t = canon_thr(t);
        if (!PACKED) {
            // sign bits of s - t shifted into four accumulators (v_alignbit), counted 32 at a time: two instructions
            // per logit, but fewer live registers than the packed form below
            int below = 0;
            unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
            for (int jb = 0; jb < NBLK; ++jb) {
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float d0 = S[jb][r + 0] - t, d1 = S[jb][r + 1] - t, d2 = S[jb][r + 2] - t, d3 = S[jb][r + 3] - t;
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc0) : "v"(d0));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc1) : "v"(d1));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc2) : "v"(d2));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc3) : "v"(d3));
                }
                if ((jb & 7) == 7 || jb == NBLK - 1) {
                    below += __builtin_popcount(acc0) + __builtin_popcount(acc1) + __builtin_popcount(acc2) + __builtin_popcount(acc3);
                    acc0 = acc1 = acc2 = acc3 = 0;
                }
            }
            return 16 * NBLK - below;
        }
        const float gc = ge_const(t);
        f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
#pragma unroll
            for (int r = 0; r < 16; r += 8) {
                const f32x2 i0 = ge_ind(f32x2{S[jb][r + 0], S[jb][r + 1]}, gc), i1 = ge_ind(f32x2{S[jb][r + 2], S[jb][r + 3]}, gc);
                const f32x2 i2 = ge_ind(f32x2{S[jb][r + 4], S[jb][r + 5]}, gc), i3 = ge_ind(f32x2{S[jb][r + 6], S[jb][r + 7]}, gc);
                a0 += i0; a1 += i1; a2 += i2; a3 += i3;
            }
        }
        const f32x2 a = (a0 + a1) + (a2 + a3);
        return (int)(a[0] + a[1] + 0.5f);             // -inf pads count as below
    };
    auto count_ge = [&](float t) { return comm.rsum(count_local(t)); };
    float thr = -INF;
    float lo = smin, hv = m;
    // ties AT the maximum only matter for tiny k (with chi taken as 1 a tied maximum still ends in "collapsed", which
    // keeps all ties); the exact count costs a pass, so it is only made when it can change the outcome
    int clo = nk, chi = (!LEAN || k <= 4) ? count_ge(m) : 1;
    // state: 0 probing, 1 done, 2 finish from above (k - chi == 1), 3 finish from below (clo - k == 1; !EXACT only:
    // with whole blocks two more probes are cheaper than the three passes of that finish)
    int state = 0;
    bool hv_est = LEAN && k > 4;                    // chi is still the assumption, not a measured count
    bool lo_meas = false;                            // lo is a probe (not the row minimum)
    if (chi >= k) { thr = m; state = 1; }            // ties at the maximum (or k == 1)
    if (nk <= k) { thr = -INF; state = 1; }          // this frame has exactly k keys: keep all
    if (state == 0 && k - chi == 1) state = 2;
    if (!LEAN && state == 0 && clo - k == 1) state = 3;
    float t = mu + zq * sd;
    // monotone integer image of a float (signed compare order) and back: bisection in this space closes ANY bracket in
    // at most 32 probes, whatever the distribution of the logits
    auto ordinal = [](float f) { const int b = __builtin_bit_cast(int, f); return b ^ ((b >> 31) & 0x7fffffff); };
    for (int it = 0; it < 80; ++it) {
        // (measured at N = 512, k = 128 / 64: 6.9 probes per wave of 16 rows; without this test one more full counting
        // pass ran only to learn that every row had finished)
        if (Comm::LOCAL_VOTE && !comm.any(state == 0)) break;
        // The density / interpolation steps below converge in ~6 probes on bell-shaped rows but only linearly on
        // bimodal or heavy-tailed ones: from the 9th probe on every other probe is the ordinal midpoint of the bracket
        // (36 of them by the end of the loop), so the search always terminates with the exact threshold.
        if (it >= 9 && (it & 1)) {
            const int ol = ordinal(lo), oh = ordinal(hv);
            const int om = ol + (int)(((unsigned)oh - (unsigned)ol) >> 1);
            t = __builtin_bit_cast(float, om ^ ((om >> 31) & 0x7fffffff));
        }
        if (!(t > lo && t < hv)) {
            t = lo + (hv - lo) * (((float)(clo - k) + 0.5f) * __builtin_amdgcn_rcpf((float)(clo - chi)));   // (only steers the search)
            if (!(t > lo && t < hv)) t = 0.5f * lo + 0.5f * hv;
        }
        const bool collapsed = !(t > lo && t < hv);   // no float strictly inside the bracket
        if (LEAN && collapsed && hv_est) t = hv;      // the count at the maximum was assumed: this pass measures it
        bool any_probing;
        const int c = comm.count_vote(count_local(t), state == 0, any_probing);   // one exchange per probe
        if (!any_probing) break;                       // every row had finished before this probe
        if (state == 0) {
            // ties at the k-th value (k or more of them at the maximum: only those)
            if (collapsed) { thr = (LEAN && hv_est && c >= k) ? hv : lo; state = 1; }
            else if (c == k) { thr = t; state = 1; }
            else {
                if (c > k) { lo = t; clo = c; lo_meas = true; } else { hv = t; chi = c; hv_est = false; }
                if (k - chi == 1) state = 2;
                else if (!LEAN && clo - k == 1) state = 3;
                else {
                    const float z = (t - mu) * inv_sd;
                    const float dens = (float)nk * 0.3989422804f * inv_sd * __builtin_amdgcn_exp2f(-0.7213475204f * z * z);
                    const float tn = t + (float)(c - k) * __builtin_amdgcn_rcpf(fmaxf(dens, 1e-3f * (float)nk * inv_sd));
                    // every 4th probe, and whenever both ends of the bracket are probes with few logits between them
                    // (the normal density says little about 48 logits): interpolate inside the bracket (t = lo does that
                    // at the top of the loop).  CPU simulation of this search, 16 rows in lockstep: 6.6 -> 5.9 probes
                    const bool narrow = lo_meas && !hv_est && clo - chi <= 48;
                    t = (((it & 3) == 3) || narrow) ? lo : tn;
                }
            }
        }
    }
    if (state == 0) { thr = lo; state = 1; }         // (probe cap; not reached, see the bisection above)
    if (comm.any(state == 2)) {   // thr = largest logit below hv: exactly k logits are >= it (more only on ties)
        float mx = -INF;
        const float hvc = canon_thr(hv);
        const float gch = ge_const(hvc);
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if (!PACKED) {
                    mx = fmaxf(mx, S[jb][r] < hvc ? S[jb][r] : -INF);
                    mx = fmaxf(mx, S[jb][r + 1] < hvc ? S[jb][r + 1] : -INF);
                    continue;
                }
                // logits at or above hv are pushed to -3e38 (indicator times -3e38 added), the rest pass unchanged
                const f32x2 s2 = {S[jb][r], S[jb][r + 1]};
                const f32x2 v = ge_ind(s2, gch) * f32x2{-3.0e38f, -3.0e38f} + s2;
                mx = fmaxf(mx, fmaxf(v[0], v[1]));
            }
        mx = comm.rmax(mx);
        if (state == 2) thr = mx;
    }
    if (!LEAN && comm.any(state == 3)) {   // k + 1 logits are >= lo: drop the smallest of them
        float e1 = INF;
        const float loc = canon_thr(lo);
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float s = S[jb][r]; e1 = fminf(e1, s >= loc ? s : INF); }
        e1 = comm.rmin(e1);
        float e2 = INF;
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float s = S[jb][r]; e2 = fminf(e2, s > e1 ? s : INF); }
        e2 = comm.rmin(e2);
        // (if the smallest is tied, dropping "it" is ambiguous: keep the ties, thr = e1)
        const int c2 = count_ge(e2);
        if (state == 3) thr = (c2 >= k) ? e2 : e1;
    }
    thr = canon_thr(thr);           // (what every later comparison uses)
    return thr;
}

// Exactly k keys per row (torch.topk keeps exactly k; which of several EQUAL logits it keeps is unspecified there -
// here the lowest key indices stay).  Rare path (about one row in 10^5 on real-valued data): the softmax pass counts
// the keys it keeps (one v_addc per logit, hidden behind the P.V MFMAs); when a row kept more than k, the tied logit
// with the largest key index is dropped, `surplus` (= kept - k, per row) times: either overwritten with -inf in the
// registers before the pass is (re)done, or - attention_topk16_kernel - its share is taken out of the written row again.
// Key index (within the source frame) of S[jb][r] in this lane = lane_off + Layout::koff(jb, r), the second part a
// compile-time constant.
// (KeyLayout32 / KeyLayout16: common.hpp)
struct NoDropHook { __device__ __forceinline__ void operator()(int, bool) const {} };
// SWEEP: write -inf over the dropped logits in the registers (the pass is then run on them).  on_drop(key, active) is
// called once per dropped key and round (`active` = this lane's row drops `key` in this round).
template <typename Layout, bool SWEEP = true, int NBLK, typename Comm, typename Hook = NoDropHook>
__device__ __forceinline__ void topk_break_ties(f32x16 (&S)[NBLK], float thr, int surplus, Comm& comm, int lane_off, Hook on_drop = Hook()) {
    // lim = smallest key index to drop: the surplus-th largest index among the tied logits, found one at a time (the
    // registers are only written after the loop: modifying S inside it costs the kernels dozens of spilled registers)
    int lim = 1 << 20;
    const float thi = tie_top(thr);
    bool dropped = false;                           // (the same in every lane that shares the vote: comm.any is wave- or workgroup-wide)
#pragma unroll 1
    while (comm.any(surplus > 0)) {
        dropped = true;
        const int below = lim - lane_off;
        int cand = -1;                              // largest koff below the limit of a tied logit in this lane
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) cand = (S[jb][r] >= thr && S[jb][r] <= thi && Layout::koff(jb, r) < below) ? Layout::koff(jb, r) : cand;   // (koff grows with (jb, r))
        // (key indices are < 4096: exact as floats)
        const int top = (int)comm.rmax(cand >= 0 ? (float)(cand + lane_off) : -1.0f);
        if (surplus > 0) lim = top;
        on_drop(top, surplus > 0);
        surplus -= 1;
    }
    if (SWEEP && dropped) {                         // (no row of the vote had a surplus - all but one tile in 10^4: nothing to sweep, and the
        const int from = lim - lane_off;           // sweep is three instructions per logit: 384 per tile of the 2048-key kernel)
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (S[jb][r] <= thi && Layout::koff(jb, r) >= from) S[jb][r] = -__builtin_inff();    // (logits below thr are not kept anyway)
    }
}

// NBLK  = 32-key blocks per chunk (16 NBLK row registers); full attention walks the keys chunk by chunk
//         with an online softmax (running max / sum, the output rescaled between chunks), dynamic
//         attention needs the whole row at once and therefore a single chunk.
// EXACT = the key count is a multiple of 32 NBLK: no per-block conditions, one basic block per chunk.
// LARGE = more than 512 keys (full attention only): LDS holds a window of 512 keys that is re-staged for
//         every query pass; the online softmax carries across windows.
// Threads: 512 (two waves per SIMD, <= 256 registers) for NBLK <= 8, else 256 (one wave per SIMD).
template <bool TOPK, int NBLK, bool EXACT, bool LARGE, bool TAP = false>
__global__ __launch_bounds__(NBLK <= 8 ? 512 : 256, NBLK <= 8 ? 2 : 1) void attention_kernel(AttnArgs a) {
    static_assert(TOPK || !TAP, "the selection tap belongs to the dynamic layers");
    static_assert(!(TOPK && LARGE), "dynamic attention needs the whole row in one chunk");
    constexpr int NT = NBLK <= 8 ? 512 : 256;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int head = blockIdx.y;
    const int b = blockIdx.z >> 1, side = blockIdx.z & 1;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N;
    const int q_off = side ? a.N : 0;
    const int src = a.cross ? (1 - side) : side;
    const int nk = src ? a.M : a.N;
    const int k_off = src ? a.N : 0;
    const int nblk = (nk + 31) >> 5;                    // all blocks of the row
    const int wcap = LARGE ? 16 : nblk;                 // blocks the LDS window holds
    const int VSTR = wcap * 32 + 8;         // V^T row stride in halves: (32 wcap + 8) / 8 is odd -> conflict free

    _Float16* Ks = smem;                    // [32 wcap][KROWH]
    _Float16* Vs = smem + wcap * 32 * KROWH;   // [2 planes][32 dims][VSTR]

    // ---- stage K (128 contiguous bytes per key) and V^T (contiguous halves per (plane, dim)) of the blocks
    //      [wb0, wb0 + wnb), four 16-byte loads in flight per thread ----
    auto stage = [&](int wb0, int wnb) {
        const _Float16* kg = a.k16 + (((size_t)b * P + k_off + wb0 * 32) * 4 + head) * 64;
        const int nk8 = wnb * 32 * 8;
        const int krem = nk - wb0 * 32;     // valid keys from the window start
        for (int base = tid; base < nk8; base += 4 * NT) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NT;
                const int row = idx >> 3, c = idx & 7;
                x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < nk8 && row < krem) x[u] = *reinterpret_cast<const f32x4*>(kg + (size_t)row * 256 + c * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NT;
                if (idx < nk8) *reinterpret_cast<f32x4*>(Ks + (idx >> 3) * KROWH + (idx & 7) * 8) = x[u];
            }
        }
        const _Float16* vg = a.vt16 + ((size_t)b * 4 + head) * 64 * a.PP + (src ? a.Npad : 0) + wb0 * 32;
        const int cpr = wnb * 4;            // 16-byte chunks per row
        const int nv = 64 * cpr;
        for (int base = tid; base < nv; base += 4 * NT) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NT;
                const int row = idx / cpr, c = idx - row * cpr;
                if (idx < nv) x[u] = *reinterpret_cast<const f32x4*>(vg + (size_t)row * a.PP + c * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NT;
                const int row = idx / cpr, c = idx - row * cpr;
                if (idx < nv) *reinterpret_cast<f32x4*>(Vs + row * VSTR + c * 8) = x[u];
            }
        }
    };
    if (!LARGE) {
        stage(0, nblk);
        __syncthreads();
    }

    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2 <-> 3
    const float NEG_INF = -__builtin_inff();
    // only the last block can be partially valid: register r of half `hi` holds key 16 (r >> 3) + 8 hi + (r & 7)
    const int last_lim = nk - (nblk - 1) * 32 - 8 * hi;
    const bool last_partial = (nk & 31) != 0;
    constexpr int QT = (NT / 64) * 32;     // queries per workgroup pass

    for (int q0 = blockIdx.x * QT; q0 < nq; q0 += gridDim.x * QT) {
        const int qw = q0 + wave * 32;
        if (!LARGE && qw >= nq) continue;   // wave-uniform; no barrier inside the loop (LARGE: barriers, every wave stays)

        // ---- this lane's query fragments: dims 16 t + 8 hi + j of query l31, planes hi / lo ----
        f16x8 qh[2], ql[2];
        {
            const int qrow = min(qw + l31, nq - 1);
            const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * hi;
            qh[0] = *reinterpret_cast<const f16x8*>(p);
            qh[1] = *reinterpret_cast<const f16x8*>(p + 16);
            ql[0] = *reinterpret_cast<const f16x8*>(p + 32);
            ql[1] = *reinterpret_cast<const f16x8*>(p + 48);
        }

        float m_run = NEG_INF;
        f32x2 l2 = {0.f, 0.f};
        f32x16 Om, Ox;                    // (main and residual terms apart: two independent chains of products)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Om[r] = 0.f; Ox[r] = 0.f; }
        bool redo = false;                                      // dynamic layers: exact ties at the k-th place
        int surplus = 0;
        const int kexp = nk <= a.topk ? (1 << 30) : a.topk;     // (every key is kept when the frame has just k of them)

        for (int wb0 = 0; wb0 < nblk; wb0 += wcap) {            // LDS windows (one unless LARGE)
            const int wnb = min(wcap, nblk - wb0);
            if (LARGE) {
                __syncthreads();            // the previous window has been consumed by every wave
                stage(wb0, wnb);
                __syncthreads();
            }
            for (int c0 = 0; c0 < wnb; c0 += NBLK) {
                // ---- S^T = K Q^T for the NBLK blocks of this chunk, resident in registers ----
                f32x16 S[NBLK];
#pragma unroll
                for (int jb = 0; jb < NBLK; ++jb) {
                    if (EXACT || c0 + jb < wnb) {
                        const _Float16* kp = Ks + ((c0 + jb) * 32 + krow) * KROWH + 8 * hi;
                        const f16x8 kh0 = *reinterpret_cast<const f16x8*>(kp);
                        const f16x8 kh1 = *reinterpret_cast<const f16x8*>(kp + 16);
                        const f16x8 kl0 = *reinterpret_cast<const f16x8*>(kp + 32);
                        const f16x8 kl1 = *reinterpret_cast<const f16x8*>(kp + 48);
                        // (unscaled residual planes, common.hpp; two accumulators all the same: one chain of six dependent
                        // products cost these kernels 5-50 %, most where a SIMD holds a single wave)
                        f32x16 acc, acx;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acx[r] = 0.f; }
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, qh[0], acc, 0, 0, 0);
                        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, ql[0], acx, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, qh[1], acc, 0, 0, 0);
                        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, ql[1], acx, 0, 0, 0);
                        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, qh[0], acx, 0, 0, 0);
                        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, qh[1], acx, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += acx[r];
                        if (!EXACT && last_partial && wb0 + c0 + jb == nblk - 1) {   // wave-uniform
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (16 * (r >> 3) + (r & 7) >= last_lim) acc[r] = NEG_INF;
                        }
                        S[jb] = acc;
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) S[jb][r] = NEG_INF;
                    }
                }

                // ---- chunk max, running max ----
                float m = NEG_INF;
#pragma unroll
                for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, S[jb][r]);
                m = fmaxf(m, xor32(m));

                // ---- exact top-k threshold (dynamic layers: the chunk is the whole row) ----
                float thr = NEG_INF;
                if (TOPK) {
                    WaveComm comm;
                    thr = topk_threshold<NBLK, EXACT>(S, m, a.topk, nk, a.zq, comm);
                    if (TAP) {      // the TAP build counts first, so that the selection it records is final
                        int c = 0;
#pragma unroll
                        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) c += S[jb][r] >= thr;
                        surplus = comm.rsum(c) - kexp;
                    }
                    if (redo || TAP) topk_break_ties<KeyLayout32>(S, thr, surplus, comm, c0 * 32 + 8 * hi);
                }
                if (TOPK && TAP && qw + l31 < nq) {
                    uint32_t* row = a.sel + (((size_t)b * 4 + head) * P + q_off + qw + l31) * a.selW;
#pragma unroll
                    for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int key0 = (c0 + jb) * 32 + 16 * t + 8 * hi;
                            unsigned bits = 0;
#pragma unroll
                            for (int j = 0; j < 8; ++j) bits |= (unsigned)(S[jb][8 * t + j] >= thr && key0 + j < nk) << j;
                            tap_keys(row, key0, bits);
                        }
                }

                if (!TOPK && (wb0 + c0) > 0) {
                    // online softmax: bring the running sum and output to the new maximum
                    const float m_new = fmaxf(m_run, m);
                    const float sc = __builtin_amdgcn_exp2f(m_run - m_new);
                    m = m_new;
                    if (!__all(sc == 1.0f)) {
                        l2 *= sc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float f = __shfl(sc, mfma32_row(r, hi), 64);   // output row r belongs to that query
                            Om[r] *= f;
                            Ox[r] *= f;
                        }
                    }
                }
                m_run = m;

                // ---- P' = 2048 exp2(s - m) split to f16 in place, row sum, O += P' V ----
                const float m11 = m - 11.0f;
                constexpr bool PKPASS = NBLK <= 8;          // (the 512-logit instance: 2x slower with the packed pass)
                const float gc = !TOPK ? 0.f : PKPASS ? ge_const(thr) : thr;
                f32x2 kept = {0.f, 0.f};
#pragma unroll
                for (int jb = 0; jb < NBLK; ++jb) {
                    if (EXACT || c0 + jb < wnb) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            float p[8], s8[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) s8[j] = S[jb][8 * t + j];
                            softmax8<TOPK, PKPASS>(s8, m11, gc, p, l2, kept);
                            f16x8 ph, pl;
                            split8(p, ph, pl);
                            const _Float16* vp = Vs + l31 * VSTR + (c0 + jb) * 32 + t * 16 + 8 * hi;
                            const f16x8 vh = *reinterpret_cast<const f16x8*>(vp);
                            const f16x8 vl = *reinterpret_cast<const f16x8*>(vp + 32 * VSTR);
                            Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, Om, 0, 0, 0);
                            Ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, Ox, 0, 0, 0);
                            Ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, Ox, 0, 0, 0);
                            // (dynamic layers: left to itself the scheduler hoists the V^T reads and splits of many steps
                            // over the indicator arithmetic - hundreds of spilled registers in the 512-register instance)
                            if (TOPK && PKPASS) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (TOPK) {
                    // a row kept more than k logits (exact ties at the k-th place): redo the chunk - the whole row -
                    // once, with the tie-break (topk_break_ties)
                    { const int kc = kept_count(kept); surplus = kc + xor32i(kc) - kexp; }
                    if (!redo && __any(surplus > 0)) {
                        redo = true;
                        c0 -= NBLK;
                        m_run = NEG_INF;
                        l2 = f32x2{0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < 16; ++r) { Om[r] = 0.f; Ox[r] = 0.f; }
                    }
                }
            }
        }
        float l = l2[0] + l2[1];
        l += xor32(l);
        const float inv_l = 1.0f / l;

        // ---- message rows: lane holds column (dim) l31 of queries mfma32_row(r, hi) ----
        float* out = a.msg + ((size_t)b * P + q_off) * 128 + head * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, hi);
            const float inv = __shfl(inv_l, row, 64);
            const int q = qw + row;
            if (q < nq) out[(size_t)q * 128] = (Om[r] + Ox[r]) * inv;
        }
    }
}

// Dynamic attention for exactly NKEY = 512 (or 256: one pair of 256 keypoints, BASELINE configs[0]) keys per frame, one wave = 16
// queries, no cross-wave traffic at all:
// S^T = K Q^T on v_mfma_f32_16x16x32_f16 (the whole 32-dim head in one k-step) puts the logits of a query in the
// four lanes (q, q + 16, q + 32, q + 48): lane (q, g) holds keys 16 b + 4 g + r of every 16-key block b, 128 registers
// for 512 keys (64 for 256), two waves per SIMD.  Row statistics and top-k counts are two lane shuffles (QuadComm), so the 8 waves
// of the workgroup search independently.  K sits in LDS as [plane][dim chunk g][key] 16-byte units (conflict free for
// this fragment shape), V^T as in the other kernels; P.V pairs two key blocks per k-step.
// FAST (mdgat_attention_mode F16): hi planes only, one MFMA per product.
template <bool FAST, bool TAP = false, int NKEY = 512>
__global__ __launch_bounds__(512, 2) void attention_topk16_kernel(AttnArgs a) {
    static_assert(NKEY == 512 || NKEY == 256, "whole 64-key register blocks, at most 128 logit registers");
    constexpr int NC = NKEY / 64;           // f32x16 logit registers blocks per lane
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l15 = lane & 15, g = lane >> 4;
    // (XCD-aware 1-D grid: AttnArgs.grid_split.  With the (part, head, pair x frame) grid the two or four parts of a unit sat on
    // different XCDs and each fetched the unit's keys: PMC 104 MB per 32-pair launch against 67 MB of q, k, v and messages)
    const int bslot = blockIdx.x >> 3;
    const int unit = (bslot / a.grid_split) * 8 + (blockIdx.x & 7), part = bslot % a.grid_split;
    if (unit >= a.grid_units) return;
    const int head = unit & 3, side = (unit >> 2) & 1, b = unit >> 3;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N;
    const int q_off = side ? a.N : 0;
    const int src = a.cross ? (1 - side) : side;
    const int nk = NKEY;
    const int k_off = src ? a.N : 0;
    constexpr int VSTR = NKEY + 8;

    // K: [512 keys][8 chunks of 16 B: hi dims 0-7, 8-15, 16-23, 24-31, lo ...], chunk c of key k stored at position
    // c ^ (k & 7): the staging writes (8 lanes = the 8 chunks of a key) and the fragment reads (lane = (key, chunk))
    // are both free of bank conflicts for the lane groups the LDS serves together
    _Float16* Ks = smem;
    _Float16* Vs = smem + 2 * 4 * NKEY * 8;       // [2 planes][32 dims][VSTR]
    {
        const _Float16* kg = a.k16 + (((size_t)b * P + k_off) * 4 + head) * 64;
        for (int base = tid; base < NKEY * 8; base += 4 * 512) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 512;
                x[u] = *reinterpret_cast<const f32x4*>(kg + (size_t)(idx >> 3) * 256 + (idx & 7) * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 512;
                const int key = idx >> 3, c = idx & 7;           // c = plane * 4 + chunk
                *reinterpret_cast<f32x4*>(Ks + ((size_t)key * 8 + (c ^ (key & 7))) * 8) = x[u];
            }
        }
        const _Float16* vg = a.vt16 + ((size_t)b * 4 + head) * 64 * a.PP + (src ? a.Npad : 0);
        constexpr int CPR = NKEY / 8;       // 16-byte pieces per V^T row
        for (int base = tid; base < 64 * CPR; base += 4 * 512) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 512;
                x[u] = *reinterpret_cast<const f32x4*>(vg + (size_t)(idx / CPR) * a.PP + (idx % CPR) * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 512;
                *reinterpret_cast<f32x4*>(Vs + (idx / CPR) * VSTR + (idx % CPR) * 8) = x[u];
            }
        }
    }
    // The 16-query tiles of this workgroup are handed out dynamically (a counter in LDS): the threshold search of a tile
    // takes 5 to 9 probes, and with a fixed four tiles per wave the workgroup would wait for its unluckiest wave.
    int* next_tile = reinterpret_cast<int*>(Vs + 2 * 32 * VSTR);
    if (tid == 0) *next_tile = 0;
    __syncthreads();

    QuadComm comm;
    const int tiles_all = (nq + 15) >> 4;
    const int tiles_wg = (tiles_all + a.grid_split - 1) / a.grid_split;  // this workgroup: tiles [part tiles_wg, ...)
    // the next 16-query tile of this wave (wave-uniform; -1: none left) and its query fragment: dims 8 g .. 8 g + 7 of query
    // l15 (B operand), planes hi / lo
    auto claim_tile = [&]() -> int {
        int t = 0;
        if (lane == 0) t = atomicAdd(next_tile, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= tiles_wg) return -1;
        const int q0 = (part * tiles_wg + t) * 16;
        return q0 < nq ? q0 : -1;
    };
    auto load_q = [&](int q0, f16x8& h, f16x8& l) {
        const int qrow = min(q0 + l15, nq - 1);
        const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * g;
        h = *reinterpret_cast<const f16x8*>(p);
        l = *reinterpret_cast<const f16x8*>(p + 32);
    };
    f16x8 qh_next, ql_next;
    int qw_next = claim_tile();
    if (qw_next >= 0) load_q(qw_next, qh_next, ql_next);
    for (;;) {
        const int qw = qw_next;
        if (qw < 0) break;                  // wave-uniform; no barrier inside the loop
        f16x8 qh = qh_next, ql = ql_next;
        // ---- S^T: NKEY / 16 blocks of 16 keys; S[c][4 j + r] = logit of key 16 (4 c + j) + 4 g + r ----
        const _Float16* kfrag_h = Ks + l15 * 64 + (g ^ (l15 & 7)) * 8;
        const _Float16* kfrag_l = Ks + l15 * 64 + ((4 + g) ^ (l15 & 7)) * 8;
        auto logits = [&](f32x16 (&S)[NC]) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int blk = 4 * c + j;
                    // A operand: key l15 of the block, dims 8 g ..; (key & 7) = (l15 & 7): the swizzle is a per-lane constant
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(kfrag_h + blk * (16 * 64));
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(kfrag_l + blk * (16 * 64));
                    // (the residual planes are unscaled, common.hpp; two accumulators all the same: three DEPENDENT 16x16x32
                    // products in a row cost this kernel more than the add they save - 148-152 -> 151-156 us per launch)
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acx = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh, acc, 0, 0, 0);
                    if (!FAST) {
                        acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql, acx, 0, 0, 0);
                        acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh, acx, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[c][4 * j + r] = FAST ? acc[r] : acc[r] + acx[r];
                }
            }
        };
        f32x16 S[NC];
        logits(S);
        float m = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, S[c][r]);
        m = comm.rmax(m);
        const float thr = topk_threshold<NC, true>(S, m, a.topk, nk, a.zq, comm);
        if (TAP) {
            // the selection the tap records is final: count and break exact ties at the k-th place before the pass
            {
                int c = 0;
#pragma unroll
                for (int jb = 0; jb < NC; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) c += S[jb][r] >= thr;
                topk_break_ties<KeyLayout16>(S, thr, comm.rsum(c) - a.topk, comm, 4 * g);
            }
            if (qw + l15 < nq) {
                uint32_t* row = a.sel + (((size_t)b * 4 + head) * P + q_off + qw + l15) * a.selW;
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned bits = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) bits |= (unsigned)(S[c2][4 * j + r] >= thr) << r;
                        tap_keys(row, 16 * (4 * c2 + j) + 4 * g, bits);
                    }
            }
        }

        // ---- P' = 2048 exp2(s - m), masked; O = P' V with two key blocks per k-step ----
        const float m11 = m - 11.0f;
        f32x2 l2 = {0.f, 0.f};
        f32x2 kept = {0.f, 0.f};
        const float gc = ge_const(thr);
        f32x4 Om[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) Om[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {             // key blocks 4 c + 2 jj and 4 c + 2 jj + 1
                float p[8], s8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) s8[i] = S[c][8 * jj + i];
                softmax8<true>(s8, m11, gc, p, l2, kept);
                f16x8 ph, pl;
                if (FAST) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ph[i] = (_Float16)p[i];
                    pl = ph;
                } else split8(p, ph, pl);
                const int key0 = (4 * c + 2 * jj) * 16 + 4 * g;  // this lane's keys: key0 .. key0 + 3 and key0 + 16 .. key0 + 19
                f16x8 vh[2], vl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {                   // dims 16 t + l15
                    const _Float16* vp = Vs + (16 * t + l15) * VSTR + key0;
                    const f16x4 vh0 = *reinterpret_cast<const f16x4*>(vp), vh1 = *reinterpret_cast<const f16x4*>(vp + 16);
                    const f16x4 vl0 = *reinterpret_cast<const f16x4*>(vp + 32 * VSTR), vl1 = *reinterpret_cast<const f16x4*>(vp + 32 * VSTR + 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { vh[t][i] = vh0[i]; vh[t][4 + i] = vh1[i]; vl[t][i] = vl0[i]; vl[t][4 + i] = vl1[i]; }
                }
                // one accumulator per dim half (the residual plane of V^T is unscaled); the two halves alternate so that no
                // product waits for the one before it
#pragma unroll
                for (int t = 0; t < 2; ++t) Om[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vh[t], Om[t], 0, 0, 0);
                if (!FAST) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) Om[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vl[t], Om[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) Om[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, vh[t], Om[t], 0, 0, 0);
                }
                // (left to itself the scheduler hoists the V^T reads and the splits of many steps: ~290 spilled registers)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float l_row = comm.rsum(l2[0] + l2[1]);
        const float inv_l = 1.0f / l_row;
        // ---- message rows: lane (dim l15 (+16 t), g) holds queries 4 g + r ----
        float* out = a.msg + ((size_t)b * P + q_off) * 128 + head * 32 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            const float inv = __shfl(inv_l, row, 64);
            const int q = qw + row;
            if (q < nq) {
                out[(size_t)q * 128] = Om[0][r] * inv;
                out[(size_t)q * 128 + 16] = Om[1][r] * inv;
            }
        }
        // the next tile is claimed and its query fragment requested HERE, before the (rare) tie repair below and the loop's
        // branch: 151-154 -> 148-151 us per launch.  (Requested earlier - right after the logits, so that the L2 round trip
        // hides behind the search - the fragment occupies 8 registers the pass needs: 76 spilled registers, 185 us.)
        qw_next = claim_tile();
        if (qw_next >= 0) load_q(qw_next, qh_next, ql_next);
        if (TAP) continue;
        // ---- exact ties at the k-th place: a row kept more than k logits (topk_break_ties; about one row in 10^5).
        // A launch waits for its slowest workgroup, so this has to be short and must not burden the code above: the
        // logits are computed again (they do not survive the pass: there are not enough registers), the tied logits
        // with the largest key indices are found one at a time and the share of each is taken out of the rows just
        // written:  o <- (o l - P' v) / (l - P'),  l <- l - P'.  (A second pass instead costs the launch 10 % at B = 64.)
        const int surplus = comm.rsum(kept_count(kept)) - a.topk;
        if (comm.any(surplus > 0)) {
            load_q(qw, qh, ql);             // (the fragment registers may have been handed to the prefetch)
            logits(S);
            const float e = __builtin_amdgcn_exp2f(thr - m11);                  // P' of a tied logit, as softmax8 computes it
            const float eh = (float)(_Float16)e;
            const float el = (float)(_Float16)(e - eh);
            topk_break_ties<KeyLayout16, false>(S, thr, surplus, comm, 4 * g, [&](int key, bool active) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {       // this lane wrote dims l15, 16 + l15 of the output rows 4 g + r
                    const int row = 4 * g + r;
                    const int x = __shfl(key, row, 64);
                    const float ph = __shfl(eh, row, 64), pl = __shfl(el, row, 64);
                    const float lr = __shfl(l_row, row, 64), er = __shfl(e, row, 64);
                    if (__shfl((int)active, row, 64) && qw + row < nq) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const _Float16* vp = Vs + (16 * t + l15) * VSTR + x;
                            const float vh = (float)vp[0], vl = (float)vp[32 * VSTR];
                            const float c = FAST ? ph * vh : fmaf(ph, vl, fmaf(pl, vh, ph * vh));
                            float* o = out + (size_t)(qw + row) * 128 + 16 * t;
                            *o = (*o * lr - c) / (lr - er);
                        }
                    }
                }
                if (active) l_row -= e;
            });
        }
    }
}

// WideComm: the keys of a row are spread over NW consecutive waves of an 8-wave workgroup (same lane = same
// query).  Partials cross through a double-buffered LDS slot with ONE barrier per exchange; every wave of the
// workgroup makes the same sequence of calls (the search is lockstep by construction).
template <int NW>
struct WideComm {
    static constexpr bool LOCAL_VOTE = false;    // any() crosses waves through LDS: the vote rides on the count exchange
    static constexpr bool PACKED_COUNT = false;  // (the kernel is at its register limit: 13 spilled registers this way, 250 with the packed count)
    // Every counting pass of this kernel is 256 instructions per wave AND a workgroup exchange: the search spends as few as it can -
    // the count at the row maximum is assumed 1 and only measured when the bracket collapses onto it (as the whole-block kernels do),
    // and a row one logit ABOVE k keeps probing instead of the three-pass finish from below (two masked minima of three
    // instructions per logit + a count: four probes' worth; simulation on rows of 2048: 9.7 -> 8.0 passes per tile).
    static constexpr bool LEAN_SEARCH = true;
    float* buf;      // [2][8 waves][64 lanes] exchange slots, [1024..1039] vote flags
    int wave, lane, par;
    float* buf4;     // [4][8 waves][64 lanes]: the four row statistics of a tile in one exchange (stats4; once per tile, so one buffer)
    template <typename Op>
    __device__ __forceinline__ float exch(float v, Op op) {
        float* b = buf + par * 512;
        b[wave * 64 + lane] = v;
        __syncthreads();
        const int g0 = (wave / NW) * NW;
        float r = b[g0 * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) r = op(r, b[(g0 + w) * 64 + lane]);
        par ^= 1;
        return r;
    }
    __device__ __forceinline__ float rsum(float v) { v += xor32(v); return exch(v, [](float x, float y) { return x + y; }); }
    __device__ __forceinline__ int rsum(int v) {
        v += xor32i(v);
        return __builtin_bit_cast(int, exch(__builtin_bit_cast(float, v), [](float x, float y) {
            return __builtin_bit_cast(float, __builtin_bit_cast(int, x) + __builtin_bit_cast(int, y)); }));
    }
    __device__ __forceinline__ float rmin(float v) { v = fminf(v, xor32(v)); return exch(v, [](float x, float y) { return fminf(x, y); }); }
    __device__ __forceinline__ float rmax(float v) { v = fmaxf(v, xor32(v)); return exch(v, [](float x, float y) { return fmaxf(x, y); }); }
    __device__ __forceinline__ bool any(bool p) { return __syncthreads_or(p) != 0; }
    __device__ __forceinline__ void stats(float& mn, float& sum, float& sq) { mn = rmin(mn); sum = rsum(sum); sq = rsum(sq); }
    // row maximum (in: this wave's part, lane pair not yet combined), minimum, sum and sum of squares through ONE barrier
    __device__ __forceinline__ void stats4(float& mx, float& mn, float& sum, float& sq) {
        mx = fmaxf(mx, xor32(mx)); mn = fminf(mn, xor32(mn)); sum += xor32(sum); sq += xor32(sq);
        float* b = buf4 + wave * 64 + lane;
        b[0] = mx; b[512] = mn; b[1024] = sum; b[1536] = sq;
        __syncthreads();
        const float* g = buf4 + (wave / NW) * NW * 64 + lane;
        mx = g[0]; mn = g[512]; sum = g[1024]; sq = g[1536];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            mx = fmaxf(mx, g[w * 64]); mn = fminf(mn, g[512 + w * 64]); sum += g[1024 + w * 64]; sq += g[1536 + w * 64];
        }
    }
    __device__ __forceinline__ int count_vote(int c, bool probing, bool& any_probing) {
        int* flags = reinterpret_cast<int*>(buf) + 1024 + par * 8;
        if (lane == 0) flags[wave] = 0;
        __builtin_amdgcn_wave_barrier();
        if (probing) flags[wave] = 1;
        const int r = rsum(c);                      // contains the barrier; `par` has flipped afterwards
        any_probing = (flags[0] | flags[1] | flags[2] | flags[3] | flags[4] | flags[5] | flags[6] | flags[7]) != 0;
        return r;
    }
};

// Dynamic attention for more than 512 keys per frame (up to 256 NW): NW waves share a 32-query tile, wave
// kw of the tile keeps the logits of keys [256 kw, 256 kw + 256) in 128 registers.  Nothing is shared between
// the waves except per-row scalars, so K and V^T fragments come straight from global memory (L2) instead of LDS.
template <int NW, bool TAP = false>
__global__ __launch_bounds__(512, 2) void attention_topk_wide_kernel(AttnArgs a) {
    constexpr int NBLK = 8;
    constexpr int NG = 8 / NW;                // query tiles per workgroup pass
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* xbuf = fsm;                        // WideComm: 1040 floats
    float* obuf = fsm + 1040;                 // [8 waves][18][64] output partials, row sums and kept counts
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int qgroup = wave / NW, kw = wave % NW;
    // 1-D grid, XCD aware: workgroups i, i + 8, i + 16, ... (one XCD under round-robin dispatch) walk the query tiles of ONE
    // (pair, frame, head), whose K and V^T they read straight from L2 - with the plain (pass, head, frame) grid every XCD
    // fetched every unit's keys: PMC 377 MB per launch at 8 x 2048 against 67 MB of q, k, v and messages
    const int slot = blockIdx.x >> 3, npg = a.grid_split;
    // blockIdx.x = slot * 8 + xcd with slot = ugroup * npg + pass0: unit = ugroup * 8 + xcd (the grid is padded to 8 units)
    const int unit = (slot / npg) * 8 + (blockIdx.x & 7);
    if (unit >= a.grid_units) return;
    const int pass0 = slot % npg;
    const int head = unit & 3, side = (unit >> 2) & 1, b = unit >> 3;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N;
    const int q_off = side ? a.N : 0;
    const int src = a.cross ? (1 - side) : side;
    const int nk = src ? a.M : a.N;
    const int k_off = src ? a.N : 0;
    const int nblk = (nk + 31) >> 5;
    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2 <-> 3
    const float NEG_INF = -__builtin_inff();
    const int last_lim = nk - (nblk - 1) * 32 - 8 * hi;
    const bool last_partial = (nk & 31) != 0;
    const _Float16* kg0 = a.k16 + (((size_t)b * P + k_off) * 4 + head) * 64;     // (wave-uniform: the base of the K descriptors)
    const int koff = (krow * 256 + 8 * hi) * (int)sizeof(_Float16);             // this lane's key row and half of the dims inside a block
    // (V^T likewise: one descriptor of this unit's two planes, the lane's row (dim l31) and half of a 16-key step as its offset, the
    // step's column as the scalar offset; the key blocks are clamped to the frame, the rows are padded: everything is in range)
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.vt16 + ((size_t)b * 4 + head) * 64 * a.PP + (src ? a.Npad : 0)), 0,
                                                                        64 * a.PP * (int)sizeof(_Float16), 0x00020000);
    const int voff = (l31 * a.PP + 8 * hi) * (int)sizeof(_Float16);
    WideComm<NW> comm{xbuf, wave, lane, 0, fsm + 1040 + 8 * 18 * 64};
    const bool whole = nk == 256 * NW;        // every wave's eight blocks are real keys: no pads anywhere in a row
    const int npass = (nq + 32 * NG - 1) / (32 * NG);

    for (int pass = pass0; pass < npass; pass += npg) {
        const int qw = (pass * NG + qgroup) * 32;
        f16x8 qh[2], ql[2];
        {
            const int qrow = min(qw + l31, nq - 1);
            const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * hi;
            qh[0] = *reinterpret_cast<const f16x8*>(p);
            qh[1] = *reinterpret_cast<const f16x8*>(p + 16);
            ql[0] = *reinterpret_cast<const f16x8*>(p + 32);
            ql[1] = *reinterpret_cast<const f16x8*>(p + 48);
        }
        const int kexp = nk <= a.topk ? (1 << 30) : a.topk;     // (every key is kept when the frame has just k of them)
        f32x16 S[NBLK];
        // K fragments come straight from L2.  ALL eight blocks' loads are issued before the first product: a block's fragments are
        // sixteen registers - exactly what its logits S[jb] will take - and S is dead at this point of a pass, so the loads of the
        // whole row cost no register the tile does not hold anyway, and the phase waits for ONE round trip to L2 instead of one per
        // block.  (Round 2-5: the loads of block jb + 1 issued before the products of block jb - one block of lookahead is ~400
        // cycles of work against a round trip of ~2 000: the phase took 18 700 cycles for 48 MFMAs.  Left to itself the compiler puts
        // every load right in front of its use and waits for it: ~24 exposed round trips per tile.)
        // Buffer loads: a block's descriptor (its first key's row of this head; bytes up to the end of the frame's keys) is built in
        // scalar registers, the lane's offset into the block is one register for the whole kernel.  Per-key 64-bit pointers - the
        // same in every pass - were hoisted out of the pass loop by the compiler, spilled (the kernel sits at 256 registers), and
        // reloaded from scratch in front of each block's loads: a scratch reload waits for EVERY load in flight.  Keys a ragged
        // last block reaches beyond the frame are out of range and load as zeros (masked below).
        f16x8 kf[NBLK][4];
        WT(0);
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
            const int gb = kw * NBLK + jb;
            const int left = nk - gb * 32 - 1;                    // keys of the frame behind this block's first
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(kg0 + (size_t)gb * (32 * 256)), 0,
                                                                               left < 0 ? 0 : left * 512 + 128, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) kf[jb][j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, koff, 32 * j, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
            const int gb = kw * NBLK + jb;
            const f16x8 kh0 = kf[jb][0], kh1 = kf[jb][1], kl0 = kf[jb][2], kl1 = kf[jb][3];
            if (gb < nblk) {
                f32x16 acc, acx;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acx[r] = 0.f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, qh[0], acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, ql[0], acx, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, qh[1], acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, ql[1], acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, qh[0], acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, qh[1], acx, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += acx[r];
                if (last_partial && gb == nblk - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (16 * (r >> 3) + (r & 7) >= last_lim) acc[r] = NEG_INF;
                }
                S[jb] = acc;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[jb][r] = NEG_INF;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        WT(1);
        float m = NEG_INF;
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, S[jb][r]);
        WT(2);
        // (m goes in as this wave's part of the row maximum and comes back as the row's: it travels with the search's statistics)
        const float thr = topk_threshold<NBLK, false>(S, m, a.topk, nk, a.zq, comm, whole);
        WT(3);
        // Exact ties at the k-th place (topk_break_ties; about one tile in 10^4).  The pass below counts what it keeps (softmax8); the
        // counts travel to LDS with the output partials, every wave sums them for the rows of the workgroup's query groups, and only
        // a tile that kept too much drops the surplus in the registers and runs the pass once more.  (Until round 6 a counting pass
        // of 256 instructions, a workgroup exchange and a vote ran in front of EVERY tile's pass: 9 000 of a tile's 85 000 cycles.)
        // The tap build still counts first: the selection it records must be final.
        if (TAP) {
            int c = 0;
#pragma unroll
            for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) c += S[jb][r] >= thr;
            const int surplus = comm.rsum(c) - kexp;
            topk_break_ties<KeyLayout32>(S, thr, surplus, comm, kw * NBLK * 32 + 8 * hi);
        }
        if (TAP && qw + l31 < nq) {
            uint32_t* row = a.sel + (((size_t)b * 4 + head) * P + q_off + qw + l31) * a.selW;
#pragma unroll
            for (int jb = 0; jb < NBLK; ++jb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int key0 = (kw * NBLK + jb) * 32 + 16 * t + 8 * hi;
                    unsigned bits = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bits |= (unsigned)(S[jb][8 * t + j] >= thr && key0 + j < nk) << j;
                    tap_keys(row, key0, bits);
                }
        }

        WT(4);
        const float m11 = m - 11.0f;
        bool redone = TAP;
#pragma unroll 1
      for (;;) {
        f32x2 l2 = {0.f, 0.f};
        f32x2 kept = {0.f, 0.f};
        f32x16 Om, Ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Om[r] = 0.f; Ox[r] = 0.f; }
        // V^T fragments likewise: the loads of step + 1 before the exponentials of this step
        f16x8 vhn, vln;
        auto vload = [&](int step, f16x8& h, f16x8& l) __attribute__((always_inline)) {
            const int gb = min(kw * NBLK + (step >> 1), nblk - 1);
            const int col = (gb * 32 + (step & 1) * 16) * (int)sizeof(_Float16);
            h = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, col, 0));
            l = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, col + 32 * a.PP * (int)sizeof(_Float16), 0));
        };
        vload(0, vhn, vln);
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
            const int gb = kw * NBLK + jb;
            {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f16x8 vh = vhn, vl = vln;
                    if (2 * jb + t + 1 < 2 * NBLK) vload(2 * jb + t + 1, vhn, vln);
                    __builtin_amdgcn_sched_barrier(0);
                    if (gb >= nblk) continue;
                    float p[8], s8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) s8[j] = S[jb][8 * t + j];
                    softmax8<true, false>(s8, m11, thr, p, l2, kept);
                    f16x8 ph, pl;
                    split8(p, ph, pl);
                    Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, Om, 0, 0, 0);
                    Ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, Ox, 0, 0, 0);
                    Ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, Ox, 0, 0, 0);
                }
            }
        }
        WT(5);
        float l = l2[0] + l2[1];
        l += xor32(l);
        int kc = kept_count(kept);
        kc += xor32i(kc);
        float* ob = obuf + wave * 18 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[r * 64 + lane] = Om[r] + Ox[r];
        ob[16 * 64 + lane] = l;
        ob[17 * 64 + lane] = __builtin_bit_cast(float, kc);
        __syncthreads();
        if (redone) break;
        {
            // what each row of the workgroup kept, summed over the waves that share it; the same numbers in every wave, so the vote
            // needs no exchange of its own
            int own = 0;
            bool any = false;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                int tot = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += __builtin_bit_cast(int, obuf[(gq * NW + w) * 18 * 64 + 17 * 64 + lane]);
                any |= tot > kexp;
                if (gq == qgroup) own = tot;
            }
            if (!__any(any)) break;
            __syncthreads();                 // (every wave has read the counts: the redone pass writes its partials over them)
            topk_break_ties<KeyLayout32>(S, thr, own - kexp, comm, kw * NBLK * 32 + 8 * hi);
            redone = true;
        }
      }
        if (kw == 0 && qw < nq) {
            float lt = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) lt += obuf[(qgroup * NW + w) * 18 * 64 + 16 * 64 + lane];
            const float inv_l = 1.0f / lt;
            // (the addresses below do not depend on the pass: hoisted out of the pass loop they live across the whole kernel,
            // are spilled, and come back as 17 serialised scratch round trips - `late` keeps them here)
            int late = 0;
            asm volatile("" : "+v"(late));
            float* out = a.msg + ((size_t)b * P + q_off) * 128 + head * 32 + l31 + late;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, hi + late);
                const float inv = __shfl(inv_l, row, 64);
                const int q = qw + row;
                float o = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) o += obuf[(qgroup * NW + w) * 18 * 64 + r * 64 + lane + late];
                if (q < nq) out[(size_t)q * 128] = o * inv;
            }
        }
        __syncthreads();     // obuf is reused by the next pass
        WT(7);
    }
}

static int launch_attention_topk_wide(const AttnArgs& a, int B, int nk_max, hipStream_t s) {
    const int nblk = (nk_max + 31) / 32;
    if (nblk > 64) {
        mdgat_set_error("dynamic attention: %d keys per frame > 2048 supported", nk_max);
        return MDGAT_ERR_UNSUPPORTED;
    }
    const size_t lds = (size_t)(1040 + 8 * 18 * 64 + 4 * 8 * 64) * sizeof(float);
    AttnArgs w = a;
    w.grid_units = B * 2 * MDGAT_HEADS;
    const int ugroups = (w.grid_units + 7) / 8;
    if (nblk <= 32) {
        w.grid_split = (nk_max + 63) / 64;
        const dim3 grid(8 * w.grid_split * ugroups);
        if (a.sel) hipLaunchKernelGGL((attention_topk_wide_kernel<4, true>), grid, dim3(512), lds, s, w);
        else hipLaunchKernelGGL(attention_topk_wide_kernel<4>, grid, dim3(512), lds, s, w);
    } else {
        w.grid_split = (nk_max + 31) / 32;
        const dim3 grid(8 * w.grid_split * ugroups);
        if (a.sel) hipLaunchKernelGGL((attention_topk_wide_kernel<8, true>), grid, dim3(512), lds, s, w);
        else hipLaunchKernelGGL(attention_topk_wide_kernel<8>, grid, dim3(512), lds, s, w);
    }
    return mdgat_check_hip(hipGetLastError(), "wide dynamic attention launch");
}

// fp32 q/k/v [B][P][3][4][32] -> the split-f16 operand layouts above (per-op entry point and tests; the
// forward's q/k/v projection writes these layouts directly from its epilogue)
__global__ __launch_bounds__(256) void qkv_split_kernel(const float* qkv, _Float16* q16, _Float16* k16, _Float16* vt16,
                                                         int B, int N, int M, int Npad, int PP) {
    const int P = N + M;
    const size_t total = (size_t)B * PP * 128;   // one thread per (b, padded column, head*32 + dim)
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx & 127);
        const size_t bc = idx >> 7;
        const int col = (int)(bc % PP), b = (int)(bc / PP);
        const int head = c >> 5, d = c & 31;
        int p = -1;
        if (col < Npad) { if (col < N) p = col; }
        else if (col - Npad < M) p = N + col - Npad;
        _Float16* vt = vt16 + (((size_t)b * 4 + head) * 2 * 32 + d) * PP + col;
        if (p < 0) { vt[0] = (_Float16)0.f; vt[(size_t)32 * PP] = (_Float16)0.f; continue; }
        const float* src = qkv + ((size_t)b * P + p) * 384 + c;
        const float q = src[0] * (MDGAT_LOG2E * 0.17677669529663687f), k = src[128], v = src[256];
        _Float16 h, l;
        const size_t o = (((size_t)b * P + p) * 4 + head) * 64 + d;
        mdgat_split_unscaled(q, h, l); q16[o] = h; q16[o + 32] = l;
        mdgat_split_unscaled(k, h, l); k16[o] = h; k16[o + 32] = l;
        mdgat_split_unscaled(v, h, l); vt[0] = h; vt[(size_t)32 * PP] = l;
    }
}

}  // namespace

size_t mdgat_qkv16_halves(int B, int N, int M) {
    const int Npad = (N + 31) & ~31, Mpad = (M + 31) & ~31;
    return (size_t)B * (N + M) * 256 * 2 + (size_t)B * 256 * (Npad + Mpad);
}

Qkv16 mdgat_qkv16_carve(_Float16* base, int B, int N, int M) {
    Qkv16 q{};
    q.Npad = (N + 31) & ~31;
    q.PP = q.Npad + ((M + 31) & ~31);
    q.q16 = base;
    q.k16 = base + (size_t)B * (N + M) * 256;
    q.vt16 = q.k16 + (size_t)B * (N + M) * 256;
    return q;
}

int launch_qkv_split(int B, int N, int M, const float* qkv, const Qkv16& o, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    const size_t total = (size_t)B * o.PP * 128;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(qkv_split_kernel, dim3(blocks), dim3(256), 0, s, qkv, o.q16, o.k16, o.vt16, B, N, M, o.Npad, o.PP);
    return mdgat_check_hip(hipGetLastError(), "qkv split launch");
}

// upper-tail standard normal quantile (Acklam's rational approximation, |error| < 1.2e-9 - only a search start)
static float normal_quantile_upper(double p) {
    if (p <= 0.0) return 8.f;
    if (p >= 1.0) return -8.f;
    const double q0 = 1.0 - p;   // lower-tail probability
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    double x;
    if (q0 < 0.02425) {
        const double q = sqrt(-2 * log(q0));
        x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    } else if (q0 <= 1 - 0.02425) {
        const double q = q0 - 0.5, r = q * q;
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
    } else {
        const double q = sqrt(-2 * log(1 - q0));
        x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    return (float)x;
}

extern "C" size_t mdgat_topk_sel_words(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return (size_t)B * 4 * (N + M) * (((N > M ? N : M) + 31) / 32); }

int launch_attention(int B, int N, int M, int cross, int topk, const Qkv16& qkv, float* msg, hipStream_t s, int mode, uint32_t* sel) {
    if (B <= 0 || N <= 0 || M <= 0) return MDGAT_OK;
    const int nk_max = N > M ? N : M;
    if (topk > 0) {
        // torch.topk raises when k exceeds the number of keys of either direction (mdgat.py:202)
        const int nk_min = N < M ? N : M;
        if (topk > nk_min) {
            mdgat_set_error("dynamic attention: k=%d exceeds the number of keys (%d)", topk, nk_min);
            return MDGAT_ERR_BAD_ARG;
        }
    }
    AttnArgs a{qkv.q16, qkv.k16, qkv.vt16, msg, N, M, qkv.Npad, qkv.PP, cross, topk, 0.f, nullptr, (nk_max + 31) / 32};
    if (topk > 0) a.zq = normal_quantile_upper(((double)topk - 0.5) / (double)nk_max);
    const int nkp = ((nk_max + 31) / 32) * 32;
    const int wkeys = nkp > 512 ? 512 : nkp;      // keys the LDS window holds
    const size_t lds = ((size_t)wkeys * KROWH + (size_t)64 * (wkeys + 8)) * sizeof(_Float16);
    // k == number of keys on both sides keeps every key: identical to full attention
    const bool dyn = topk > 0 && !(topk == N && topk == M);
    const int nblk = nkp / 32;
    if (sel && topk > 0) {
        // parity tap: the kept keys of every (pair, head, query) as a bit mask; k == number of keys keeps them all
        if (int rc = mdgat_check_hip(hipMemsetAsync(sel, dyn ? 0 : 0xff, mdgat_topk_sel_words(B, N, M) * sizeof(uint32_t), s), "memset(top-k tap)")) return rc;
        if (dyn) a.sel = sel;
    }
    if (!dyn && attention_stream_supported(N, M) && !getenv("MDGAT_ATTN_NOSTREAM")) return launch_attention_stream(B, N, M, cross, qkv, msg, s, mode);
    // one workgroup per (pair, frame, head) loops over its query tiles; split the tiles over more
    // workgroups only when there are too few (pair, frame, head) units to fill the chip twice
    auto go = [&](auto kern, int threads) {
        const int qt = (threads / 64) * 32;
        const int qtiles = (nk_max + qt - 1) / qt;
        int qsplit = (512 + B * 2 * MDGAT_HEADS - 1) / (B * 2 * MDGAT_HEADS);
        if (qsplit > qtiles) qsplit = qtiles;
        if (qsplit < 1) qsplit = 1;
        if (nk_max > 512) qsplit = qtiles;     // windowed kernel: every pass re-stages its keys anyway
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(qsplit, MDGAT_HEADS, B * 2), dim3(threads), lds, s, a);
    };
    const bool mult32 = (N % 32 == 0) && (M % 32 == 0);
    // exactly 512 (or 256) keys in both frames: one wave = 16 queries, the row in four lanes
    auto launch16 = [&](auto nkc) {
        constexpr int NKEY = decltype(nkc)::value;
        // a workgroup is 8 waves = 8 query tiles per pass: NKEY / 128 workgroups per (pair, frame, head) keep every wave busy;
        // more (every workgroup stages all keys: up to 4, half the waves idle at 256 keys) only while the launch is smaller than the part
        int qsplit = (512 + B * 2 * MDGAT_HEADS - 1) / (B * 2 * MDGAT_HEADS);
        const int qmax = B * 2 * MDGAT_HEADS * (NKEY / 128) >= 256 ? NKEY / 128 : 4;
        if (qsplit > qmax) qsplit = qmax;
        const size_t lds3 = ((size_t)2 * 4 * NKEY * 8 + (size_t)64 * (NKEY + 8)) * sizeof(_Float16) + 16;    // + the tile counter
        AttnArgs w = a;
        w.grid_split = qsplit;
        w.grid_units = B * 2 * MDGAT_HEADS;
        auto run = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipLaunchKernelGGL(kern, dim3(8 * qsplit * ((w.grid_units + 7) / 8)), dim3(512), lds3, s, w);
        };
        if (a.sel) { if (mode == 1) run(attention_topk16_kernel<true, true, NKEY>); else run(attention_topk16_kernel<false, true, NKEY>); }
        else if (mode == 1) run(attention_topk16_kernel<true, false, NKEY>);
        else run(attention_topk16_kernel<false, false, NKEY>);
    };
    if (dyn) {
        // the whole row in one chunk
        if (N == 256 && M == 256) launch16(std::integral_constant<int, 256>{});
        else if (a.sel && !(N == 512 && M == 512) && nblk <= 16) {
            if (nblk <= 4) go(attention_kernel<true, 4, false, false, true>, 512);
            else if (nblk <= 8) go(attention_kernel<true, 8, false, false, true>, 512);
            else go(attention_kernel<true, 16, false, false, true>, 256);
        } else if (nblk <= 4) go(attention_kernel<true, 4, false, false>, 512);
        else if (nblk <= 8) go(attention_kernel<true, 8, false, false>, 512);
        else if (N == 512 && M == 512) launch16(std::integral_constant<int, 512>{});
        else if (nblk <= 16) go(attention_kernel<true, 16, false, false>, 256);
        else return launch_attention_topk_wide(a, B, nk_max, s);
    } else {
        // chunks of 8 blocks (256 keys), two waves per SIMD
        if (nblk <= 4) go(attention_kernel<false, 4, false, false>, 512);
        else if (nblk > 16) {
            if (mult32 && N % 256 == 0 && M % 256 == 0) go(attention_kernel<false, 8, true, true>, 512);
            else go(attention_kernel<false, 8, false, true>, 512);
        } else if (mult32 && N % 256 == 0 && M % 256 == 0) go(attention_kernel<false, 8, true, false>, 512);
        else go(attention_kernel<false, 8, false, false>, 512);
    }
    return mdgat_check_hip(hipGetLastError(), "attention launch");
}
