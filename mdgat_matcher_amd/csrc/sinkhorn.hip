// Log-domain Sinkhorn optimal transport with dustbins and match extraction.
//
// Replaces: log_optimal_transport (mdgat.py:288-308), log_sinkhorn_iterations (279-285) and the match
// extraction of mdgat.py:441-483.
//
// The (N+1) x (M+1) coupling matrix of mdgat.py:298-299 is the N x M score block bordered by the constant
// bin score, so the kernel never materialises it: only the scores are read, the border terms are
// closed-form (alpha + v_M for every row, alpha + u_N for every column, and the dustbin row/column
// potentials are log-sum-exps of the potential vectors themselves).
//
// gfx950 mapping (round 1): one workgroup per pair, NW waves.  Per iteration the score block is streamed
// ONCE (it is L2/MALL resident: 1 MB per pair at N=M=512): a wave takes rows i = wave, wave+NW, ...;
// lane l holds columns l, l+64, ... of the row in registers, adds the column potentials (registers),
// reduces max / sum-of-exp2 across the wave -> u_i, and immediately folds the same registers, now with
// the fresh u_i, into per-lane running (max, sum) accumulators of the COLUMN log-sum-exps.  The NW
// per-wave column partials are merged through LDS -> v_j.  Everything is kept in the base-2 log domain
// (potentials carry a factor log2(e)) so each element costs v_exp_f32 without a pre-multiply.
#include "common.hpp"
#include <cstdlib>

#ifdef SK_TRACE
// phase trace of the cluster kernel (tools/ab_build.sh sinkhorn sktrace -DSK_TRACE; tools/sinkhorn_trace.py): s_memtime stamps
// of waves 0 and 7 of workgroup 0 in iterations 40-47, 11 points per iteration
__device__ long long g_sk_trace[2 * 8 * 12];
extern "C" int mdgat_sk_trace_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sk_trace), n * sizeof(long long)); }
#define SK_TP(k) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7) && it >= 40 && it < 48) \
    g_sk_trace[((wave == 7) * 8 + (it - 40)) * 12 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// coarse phases of the whole kernel (workgroup 0, waves 0 and 7; tools/sinkhorn_phases.py): start | scores loaded | row maxima
// absorbed | XCD handshake | iterations done | Z rows + row arg-maxes | end
__device__ long long g_sk_phase[2 * 8];
extern "C" int mdgat_sk_phase_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sk_phase), n * sizeof(long long)); }
#define SK_PH(k) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) g_sk_phase[(wave == 7) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SK_TP(k) do {} while (0)
#define SK_PH(k) do {} while (0)
#endif
namespace {

constexpr float NEG_BIG = -1.0e30f;   // finite stand-in for -inf (keeps a - b well defined)

struct SkArgs {
    const float* scores;      // [B][N][M]
    const float* alpha_dev;   // device scalar or nullptr
    float alpha_host;
    float* Z;                 // [B][N+1][M+1]
    int N, M, iters;
    // fallback use: a pair is only redone if bit 0 of *only_if is set (the cluster kernel's error word: the launch lost a partner
    // workgroup - every pair is redone) or its own word of pair_flags is (the pair's scores are beyond the range of the scaling
    // form - the other pairs of the launch keep the cluster kernel's result: what a pair gets does not depend on its batch);
    // a workgroup that runs raises status_fallback (host-mapped, optional)
    const unsigned* only_if;
    unsigned* status_fallback;
    const unsigned* pair_flags;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32 = log2

// log2-sum-exp2 over `n` LDS values plus one extra term, by one wave
__device__ __forceinline__ float wave_lse2(const float* x, int n, float extra, int lane) {
    float m = extra;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, x[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += ex2(x[i] - m);
    s = wave_sum(s) + ex2(extra - m);
    return m + lg2(s);
}

template <int NC, int NW>
__global__ __launch_bounds__(NW * 64) void sinkhorn_kernel(SkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, M = a.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.only_if) {
        const bool mine = (__hip_atomic_load(a.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0 ||
                          (a.pair_flags && __hip_atomic_load(a.pair_flags + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        if (!mine) return;
        if (tid == 0 && a.status_fallback) __hip_atomic_store(a.status_fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float* S = a.scores + (size_t)blockIdx.x * N * M;
    float* Z = a.Z + (size_t)blockIdx.x * (N + 1) * (M + 1);

    float* u = smem;                 // [N+1]
    float* v = u + (N + 1);          // [M+1]
    float* pm = v + (M + 1);         // [NW][M]
    float* ps = pm + NW * M;         // [NW][M]

    const float alpha = (a.alpha_dev ? *a.alpha_dev : a.alpha_host) * MDGAT_LOG2E;
    const float norm = -logf((float)(N + M));                 // mdgat.py:301
    const float lmu = norm * MDGAT_LOG2E;                     // rows 0..N-1 (302)
    const float lmuN = (logf((float)M) + norm) * MDGAT_LOG2E; // dustbin row
    const float lnu = lmu;                                    // cols 0..M-1 (303)
    const float lnuM = (logf((float)N) + norm) * MDGAT_LOG2E; // dustbin column

    for (int i = tid; i <= N; i += NW * 64) u[i] = 0.f;
    for (int j = tid; j <= M; j += NW * 64) v[j] = 0.f;
    __syncthreads();

    for (int it = 0; it < a.iters; ++it) {
        // column potentials of this lane's columns
        float vr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = c * 64 + lane;
            vr[c] = j < M ? v[j] : 0.f;
        }
        const float bM = alpha + v[M];   // dustbin-column term of every row
        if (wave == NW - 1) {
            // dustbin row: u_N = log_mu_N - LSE_j(alpha + v_j), j = 0..M
            const float lse = alpha + wave_lse2(v, M, v[M], lane);
            if (lane == 0) u[N] = lmuN - lse;
        }
        float cm[NC], cs[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { cm[c] = NEG_BIG; cs[c] = 0.f; }

        for (int i = wave; i < N; i += NW) {
            const float* row = S + (size_t)i * M;
            float s[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = c * 64 + lane;
                s[c] = j < M ? row[j] * MDGAT_LOG2E : NEG_BIG;
            }
            float m = bM;
#pragma unroll
            for (int c = 0; c < NC; ++c) m = fmaxf(m, s[c] + vr[c]);
            m = wave_max(m);
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) e += ex2(s[c] + vr[c] - m);
            e = wave_sum(e) + ex2(bM - m);
            const float ui = lmu - (m + lg2(e));          // mdgat.py:283
            if (lane == 0) u[i] = ui;
#pragma unroll
            for (int c = 0; c < NC; ++c) {                // fold into the column LSEs (mdgat.py:284)
                const float w = s[c] + ui;
                const float nm = fmaxf(cm[c], w);
                cs[c] = cs[c] * ex2(cm[c] - nm) + ex2(w - nm);
                cm[c] = nm;
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = c * 64 + lane;
            if (j < M) { pm[wave * M + j] = cm[c]; ps[wave * M + j] = cs[c]; }
        }
        __syncthreads();
        const float bN = alpha + u[N];   // dustbin-row term of every column
        for (int j = tid; j < M; j += NW * 64) {
            float m = bN;
#pragma unroll
            for (int w = 0; w < NW; ++w) m = fmaxf(m, pm[w * M + j]);
            float e = ex2(bN - m);
#pragma unroll
            for (int w = 0; w < NW; ++w) e += ps[w * M + j] * ex2(pm[w * M + j] - m);
            v[j] = lnu - (m + lg2(e));
        }
        if (wave == 0) {
            // dustbin column: v_M = log_nu_M - LSE_i(alpha + u_i), i = 0..N
            const float lse = alpha + wave_lse2(u, N, u[N], lane);
            if (lane == 0) v[M] = lnuM - lse;
        }
        __syncthreads();
    }

    // Z = couplings + u + v - norm (mdgat.py:285, 307), back in natural-log units
    const float uN = u[N], vM = v[M];
    for (int i = wave; i < N; i += NW) {
        const float ui = u[i];
        const float* row = S + (size_t)i * M;
        float* zr = Z + (size_t)i * (M + 1);
        for (int j = lane; j < M; j += 64) zr[j] = (row[j] * MDGAT_LOG2E + ui + v[j]) * MDGAT_LN2 - norm;
        if (lane == 0) zr[M] = (alpha + ui + vM) * MDGAT_LN2 - norm;
    }
    float* zl = Z + (size_t)N * (M + 1);
    for (int j = tid; j <= M; j += NW * 64) zl[j] = (alpha + uN + v[j]) * MDGAT_LN2 - norm;
}


// ------------------------------------------------------------------------------------------------
// helpers of the cluster kernel
constexpr int SLOT_STRIDE = 520;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
// wave-wide reductions on the VALU (DPP row shifts + row broadcasts); result broadcast from lane 63
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_move<0x111, 0xf>(NEG_BIG, v));   // row_shr:1
    v = fmaxf(v, dpp_move<0x112, 0xf>(NEG_BIG, v));   // row_shr:2
    v = fmaxf(v, dpp_move<0x114, 0xf>(NEG_BIG, v));   // row_shr:4
    v = fmaxf(v, dpp_move<0x118, 0xf>(NEG_BIG, v));   // row_shr:8
    v = fmaxf(v, dpp_move<0x142, 0xa>(NEG_BIG, v));   // row_bcast:15 -> rows 1, 3
    v = fmaxf(v, dpp_move<0x143, 0xc>(NEG_BIG, v));   // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_move<0x111, 0xf>(0.f, v);
    v += dpp_move<0x112, 0xf>(0.f, v);
    v += dpp_move<0x114, 0xf>(0.f, v);
    v += dpp_move<0x118, 0xf>(0.f, v);
    v += dpp_move<0x142, 0xa>(0.f, v);
    v += dpp_move<0x143, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sixteen wave-wide sums at once: acc[r] = this lane's partial of row r; returns, in lane r (< 16), the wave total of
// row r.  Each stage halves the number of registers while doubling the lanes already summed: lane-half swap
// (v_permlane32_swap), 16-lane-row swap (v_permlane16_swap), row rotation by 8, bank shifts by 4, two quad steps -
// about 40 instructions and short dependency chains instead of sixteen 7-deep DPP chains.  (The permlane swaps go
// through inline asm: the clang builtins of this ROCm return the first result twice.)
// The compiler does not see through an asm, so it inserts none of the wait states these instructions need after a
// vector write of their operands / before a vector read of their results (observed: wrong values without them).
__device__ __forceinline__ void swap_halves32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_rows16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_bank_move(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ float wave_sum16(float (&acc)[16], int lane) {
    float s[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { float x = acc[r], y = acc[r + 8]; swap_halves32(x, y); s[r] = x + y; }   // lanes < 32: row r, others row r + 8
    float t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { float x = s[r], y = s[r + 4]; swap_rows16(x, y); t[r] = x + y; }      // 16-lane row q: row r + 4 q
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    float v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float keep = b3 ? t[r + 2] : t[r], give = b3 ? t[r] : t[r + 2];
        v[r] = keep + dpp_bank_move<0x128, 0xf>(0.f, give);            // row_ror:8
    }
    const float keep = b2 ? v[1] : v[0], give = b2 ? v[0] : v[1];
    float o = dpp_bank_move<0x104, 0x5>(0.f, give);                    // row_shl:4 into lanes 0-3, 8-11 of each row
    o = dpp_bank_move<0x114, 0xa>(o, give);                            // row_shr:4 into lanes 4-7, 12-15
    float w = keep + o;
    w += dpp_bank_move<0xb1, 0xf>(0.f, w);                             // quad_perm [1, 0, 3, 2]
    w += dpp_bank_move<0x4e, 0xf>(0.f, w);                             // quad_perm [2, 3, 0, 1]
    // lanes 4 q .. 4 q + 3 hold the total of row q: bring it to lane q
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(16 * lane, __builtin_bit_cast(int, w)));
}

// Eight wave-wide sums at once (the same scheme one stage shorter): returns, in lane r (< 8), the wave total of row r.
__device__ __forceinline__ float wave_sum8(float (&acc)[8], int lane) {
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { float x = acc[r], y = acc[r + 4]; swap_halves32(x, y); s[r] = x + y; }   // lanes < 32: row r, others row r + 4
    float t[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) { float x = s[r], y = s[r + 2]; swap_rows16(x, y); t[r] = x + y; }      // 16-lane row q: row r + 2 q
    const bool b3 = (lane & 8) != 0;
    const float keep = b3 ? t[1] : t[0], give = b3 ? t[0] : t[1];
    float w = keep + dpp_bank_move<0x128, 0xf>(0.f, give);             // row_ror:8: lanes 16 q + 0..7 row 2 q, lanes 16 q + 8..15 row 2 q + 1
    float o = dpp_bank_move<0x104, 0x5>(0.f, w);                       // the value of lane ^ 4
    o = dpp_bank_move<0x114, 0xa>(o, w);
    w += o;
    w += dpp_bank_move<0xb1, 0xf>(0.f, w);                             // quad_perm [1, 0, 3, 2]
    w += dpp_bank_move<0x4e, 0xf>(0.f, w);                             // quad_perm [2, 3, 0, 1]
    // lanes 8 q .. 8 q + 7 hold the total of row q: bring it to lane q
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(32 * lane, __builtin_bit_cast(int, w)));
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
// cache policy of the polling loads: agent scope (served by the L2, never by a CU's L1).  "nt" measures the same and had
// been used until round 3; "sc0" alone hits stale L1 lines (partners time out).
#define SK_LD "sc1"

// ------------------------------------------------------------------------------------------------
// Scaling-form cluster kernel (N, M <= 512), the production path.
//
// The log-domain iteration of mdgat.py:283-284 is the classical Sinkhorn matrix scaling written for
// numerical range.  Here the range is handled once: with K_ij = exp2(s_ij + u0_i + v0_j) for absorbed
// potentials (u0_i = - max(row max, dustbin) at start, v0 = 0) the SAME iteration is
//      a_i = mu_i / (sum_j K_ij b_j),   b_j = nu_j / (sum_i K_ij a_i),      u = u0 + log2 a,  v = v0 + log2 b
// i.e. two FMAs per matrix element and iteration instead of two exp2 and eight adds / maxes, and no
// transcendental in the loop.  Every row of K has its largest entry equal to 1 and the dustbin row /
// column entries are positive, so no sum can vanish; entries that underflow are < 2^-126 of their
// row's largest and cannot matter.  When a scaling factor leaves [2^-40, 2^40] it is folded back into
// K and the absorbed potentials (b: decided on b alone, which is bit-identical in all workgroups of a
// pair; a: per row, purely local), so any dynamic range the log-domain form handles is handled here.
//
// G workgroups (8 waves each, one per CU) share one pair: workgroup j owns rows [128 j, 128 j + 128), wave w of it
// 16 of those rows, lane l the columns 8 l .. 8 l + 7: a lane keeps a 16 x 8 block of K (128 VGPRs) for all
// iterations.  Row sums: 8 in-lane FMAs per row + one transposed wave reduction for the 16 rows (wave_sum16).  Column
// sums: in-lane over the wave's 16 rows, the 8 waves merged through LDS, the G workgroups through L2 as 8-byte
// {epoch, value} granules (the data is the flag, two slot sets alternate by epoch parity; relaxed agent-scope atomics,
// or non-temporal accesses once the partners have agreed that they share an XCD; partners are placed on one XCD for
// speed only; spins are bounded: a timeout raises the launch's error word, and the one-workgroup-per-pair streaming kernel
// that follows every cluster launch - it leaves at once otherwise - then redoes the whole launch: the result is correct
// whatever else occupies the device, only late).  The launch is a plain one when every workgroup has a CU of its own
// (launch_scaling), cooperative otherwise.
struct SksArgs {
    const float* scores;
    const float* alpha_dev;
    float alpha_host;
    float* Z;
    unsigned long long* slots;   // per group: column slots [2][GC][GR][SLOT_STRIDE], then row slots [2][GR][GC][ROW_STRIDE]; zeroed per launch
    unsigned* error_word;        // bit 0: a workgroup gave up waiting for a partner (the whole launch is redone); bit 1: some pair is out of range
    unsigned* pair_flags;        // [B], zeroed per launch: this pair's scores are beyond the range of the scaling form (it alone is redone)
    unsigned* range_guard;       // optional, host-mapped: set when a score is not finite (an activation upstream left the f16 range)
    int B, N, M, iters, ngroups, GR, GC;
    int xcd_map;         // workgroup blockIdx = (slot * P + partner) * 8 + xcd: the partners of a pair share blockIdx % 8 (launch_scaling)
    // fused arg-max of the match extraction (mdgat.py:441-483): per row over this workgroup's columns, per column over
    // its rows (merged later); ext_mode < 0: off.  Z may be NULL when only the matches are wanted.
    int ext_mode;
    int* rbest_idx;      // [B][GC][N]
    float* rbest_val;    // [B][GC][N]
    int* cbest_idx;      // [B][GR][M]
    float* cbest_val;    // [B][GR][M]
};

constexpr int SKS_THREADS = 512;
constexpr int SKS_LDS_FLOATS = 520 + 8 * 512 + 8 + 8 + 8 * 512;
constexpr int ROW_STRIDE = 136;

// Granule traffic between the workgroups of a pair.  Agent scope (sc1) works wherever the partners run.  When all of
// them report the same XCC_ID (checked per pair with an agent-scope exchange first), they share one L2: plain 8-byte
// stores and agent-scope loads issued by hand (several in flight, one wait) carry the hand-off at lower latency.
__device__ __forceinline__ unsigned long long xload(gu64* p, bool same_xcd) {
    if (same_xcd) {
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off " SK_LD "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xstore(gu64* p, unsigned long long v, bool same_xcd) {
    // A PLAIN store: the CU's L1 writes through to the XCD's L2, where the partners' (L1-bypassing) loads find it, and it stays
    // there.  With `nt` - as until round 3 - the L2 streamed every granule on to HBM: 115 MB written per launch at B = 64
    // (PMC WRITE_SIZE) against 11 MB now, and the hand-off waited for it: 3.2 -> 2.85 us per iteration, 391 -> 355 us per launch.
    if (same_xcd) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll the granules `base[p * stride]`, p in [0, n) except `self`, until they carry `tag`; all outstanding partners are
// polled concurrently.  vals[p] receives the payloads.  Bounded: a timeout sets the error word.
// `extra` (optional): one more granule polled in the same round trips (the dustbin column's partial of a partner, by the
// lanes of the last wave that own it): extra_val receives its payload.
template <int NMAX>
__device__ __forceinline__ void poll_partners(gu64* base, size_t stride, int n, int self, unsigned tag, float (&vals)[NMAX],
                                              bool& failed, unsigned* error_word, bool same_xcd = false,
                                              gu64* extra = nullptr, float* extra_val = nullptr) {
    unsigned pending = ((1u << n) - 1u) & ~(1u << self);
    bool extra_pending = extra != nullptr;
    unsigned spins = 0;
    while (pending || extra_pending) {
        unsigned long long x[NMAX], xe = 0;
        if (NMAX <= 4 && same_xcd) {
            // all outstanding partners in ONE round trip: the loads are issued back to back and awaited together (a wait per
            // load cost 346 against 337 us per launch at B = 64 once the stores had become plain; with up to 16 partners
            // the sixteen granules in flight cost 86 more spilled registers: those kernels keep the serial poll)
#pragma unroll
            for (int p = 0; p < NMAX; ++p)
                if (pending & (1u << p)) asm volatile("global_load_dwordx2 %0, %1, off " SK_LD : "=v"(x[p]) : "v"(base + (size_t)p * stride) : "memory");
            if (extra_pending) asm volatile("global_load_dwordx2 %0, %1, off " SK_LD : "=v"(xe) : "v"(extra) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < NMAX; ++p)
                if (pending & (1u << p)) asm volatile("" : "+v"(x[p]));       // (uses stay behind the wait)
            asm volatile("" : "+v"(xe));
        } else if (same_xcd) {
            // more than four partners on one XCD (up to 16 row slabs, N <= 2048 with M <= 512): four granules in flight per round
            // trip, consumed before the next four are requested (all sixteen at once cost 86 spilled registers; one at a time - a
            // wait per load, as until round 4 - a round trip each: 8 x 2048 x 512 0.747 -> 0.592 ms per 100 iterations)
#pragma unroll
            for (int p = 0; p < NMAX; ++p) x[p] = 0;
#pragma unroll
            for (int p0 = 0; p0 < NMAX; p0 += 4) {
                if (!((pending >> p0) & 0xfu) && !(extra_pending && p0 == 0)) continue;
                unsigned long long y[4] = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (pending & (1u << (p0 + u))) asm volatile("global_load_dwordx2 %0, %1, off " SK_LD : "=v"(y[u]) : "v"(base + (size_t)(p0 + u) * stride) : "memory");
                if (extra_pending && p0 == 0) asm volatile("global_load_dwordx2 %0, %1, off " SK_LD : "=v"(xe) : "v"(extra) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    asm volatile("" : "+v"(y[u]));
                    if ((pending & (1u << (p0 + u))) && (unsigned)(y[u] >> 32) == tag) {
                        vals[p0 + u] = __builtin_bit_cast(float, (unsigned)y[u]);
                        pending &= ~(1u << (p0 + u));
                    }
                }
                asm volatile("" : "+v"(xe));
            }
        } else {
            // partners on different XCDs: agent-scope atomic loads, which the compiler pipelines (all outstanding ones in flight) -
            // as long as this loop holds nothing else: until round 4 it went through xload() with its run-time same_xcd branch per
            // load, which serialised the sixteen loads of the two-dimensional kernels (8 x 2048 x 2048 1.81 -> 1.55 ms per 100
            // iterations, 8 x 1024 x 1024 0.805 -> 0.687)
#pragma unroll
            for (int p = 0; p < NMAX; ++p)
                if (pending & (1u << p)) x[p] = __hip_atomic_load(base + (size_t)p * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (extra_pending) xe = __hip_atomic_load(extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int p = 0; p < NMAX; ++p)
            if ((pending & (1u << p)) && (unsigned)(x[p] >> 32) == tag) {
                vals[p] = __builtin_bit_cast(float, (unsigned)x[p]);
                pending &= ~(1u << p);
            }
        if (extra_pending && (unsigned)(xe >> 32) == tag) {
            *extra_val = __builtin_bit_cast(float, (unsigned)xe);
            extra_pending = false;
        }
        if (pending || extra_pending) {
            if (failed || ++spins > (1u << 22)) { failed = true; atomicOr(error_word, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// The pair's matrix is tiled GR x GC: workgroup (jr, jc) keeps rows [128 jr, 128 jr + 128) x columns [512 jc, 512 jc + 512)
// in registers: wave w 16 of those rows, lane l the 8 columns 8 l .. 8 l + 7 (RPW x 8 block, 128 VGPRs).  Per-row
// quantities (absorbed potential, dustbin-column entry, scaling a) live in lane r of the wave for row r; the scaling is
// broadcast for the column pass with v_readlane.  Per-column quantities of the lane's 8 columns live in registers; the
// thread that finalises column t keeps its own copies.  Row sums cross the GC column slabs, column sums the GR row slabs
// (TWO_D = more than one column slab: N, M up to 2048).
// GMAX = compile-time bound on the row slabs (4: N <= 512, 16: N <= 2048)
template <int RPW, bool TWO_D, int GMAX>
__global__ __launch_bounds__(SKS_THREADS, 2) void sinkhorn_scaling_kernel(SksArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[SKS_LDS_FLOATS];
    float* bvec = lds;                    // [512] column scalings of the slab, [512] = dustbin column
    float* colp = lds + 520;              // [8 waves][512] per-wave column sums
    float* pdust = colp + 8 * 512;        // [8] per-wave sums of the dustbin column
    int* flags = reinterpret_cast<int*>(pdust + 8);   // [0]: a column scaling of the slab left the safe range
    int* colpi = flags + 8;               // [8 waves][512] row indices of the per-wave column maxima (fused extraction)

    const int N = a.N, M = a.M, GR = a.GR, GC = TWO_D ? a.GC : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = GR * GC;
    int group, w;
    if (a.xcd_map) {                      // partners share blockIdx % 8 (observed: the XCD); the grid is padded to 8 groups per slot
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        w = q % P;
        group = (q / P) * 8 + xcd;
        if (group >= a.ngroups) return;   // (padding: the whole workgroup leaves)
    } else {
        group = blockIdx.x / P;
        w = blockIdx.x % P;
    }
    const int jr = w / GC, jc = w % GC;
    const float alpha = (a.alpha_dev ? *a.alpha_dev : a.alpha_host) * MDGAT_LOG2E;
    const float norm = -logf((float)(N + M));
    const float mu = 1.0f / (float)(N + M);                    // exp(log_mu) of mdgat.py:302 (rows 0..N-1)
    const float muN = (float)M / (float)(N + M);               // dustbin row
    const float nu = mu;                                       // columns 0..M-1 (mdgat.py:303)
    const float nuM = (float)N / (float)(N + M);               // dustbin column
    const int row0 = (jr * 8 + wave) * RPW;   // first row of this wave
    const int col0 = lane * 8;                // first local column of this lane
    const int gcol0 = jc * 512 + col0;        // ... global
    const bool my_row_valid = lane < RPW && row0 + lane < N;   // lane r carries the per-row state of row r
    const size_t col_slots = (size_t)2 * GC * GR * SLOT_STRIDE, row_slots = (size_t)2 * GR * GC * ROW_STRIDE;
    gu64* cslots = (gu64*)(a.slots) + (size_t)group * (col_slots + row_slots);
    gu64* rslots = cslots + col_slots;
    unsigned cep = 0, rep_ = 0, xep = 0;  // column / row / XCC-id exchange counters (granule tags)
    bool failed = false;
    const float RANGE_HI = 1.099511627776e12f, RANGE_LO = 9.094947017729282e-13f;   // 2^40, 2^-40

    // sum (or max) of one per-row value (lanes < RPW, plus lane RPW of wave 0 for the dustbin row) over the GC column slabs
    auto row_exchange = [&](float v, bool take_max) -> float {
        if (!TWO_D) return v;
        ++rep_;
        const bool active = lane <= RPW;          // lane RPW: the dustbin row (every wave publishes / polls the same value)
        const int idx = lane < RPW ? wave * RPW + lane : 128;
        float out = v;
        if (active) {
            gu64* base = rslots + ((size_t)(rep_ & 1) * GR + jr) * GC * ROW_STRIDE + idx;
            __hip_atomic_store(base + (size_t)jc * ROW_STRIDE, ((unsigned long long)rep_ << 32) | __builtin_bit_cast(unsigned, v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float vals[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) vals[p] = take_max ? NEG_BIG : 0.f;
            poll_partners<4>(base, ROW_STRIDE, GC, jc, rep_, vals, failed, a.error_word);
            out = take_max ? NEG_BIG : 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) {             // fixed order: bit-identical in every partner (GC <= 4)
                const float x = (p == jc) ? v : vals[p];
                out = take_max ? fmaxf(out, x) : out + x;
            }
        }
        return out;
    };

    for (int pair = group; pair < a.B; pair += a.ngroups) {
        const float* S = a.scores + (size_t)pair * N * M;
        SK_PH(0);
        // ---- this lane's RPW x 8 block of scores (base-2 log units); invalid entries -> exp2 gives 0 ----
        float K[RPW][8];
        const bool vec_ok = (M & 3) == 0 && gcol0 + 8 <= M;
        unsigned gmax = 0;                // largest |score| of this lane as an integer image: NaN / inf on top
        // (the path is chosen per LANE, outside the row loop: chosen per row, the three-way branch kept the loads of different rows
        // in different basic blocks - the ragged path then ran its 128 loads one round trip at a time)
        if (vec_ok) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int i = row0 + r;
                const float* row = S + (size_t)min(i, N - 1) * M;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(row + gcol0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(row + gcol0 + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    gmax = max(gmax, max(__builtin_bit_cast(unsigned, x0[c]) & 0x7fffffffu, __builtin_bit_cast(unsigned, x1[c]) & 0x7fffffffu));
                    K[r][c] = i < N ? x0[c] * MDGAT_LOG2E : NEG_BIG;
                    K[r][4 + c] = i < N ? x1[c] * MDGAT_LOG2E : NEG_BIG;
                }
            }
        } else if (gcol0 >= M) {
            // a lane whose eight columns all lie beyond M loads nothing (M = 256: half of every wave - as clamped scalar
            // loads, 128 per lane, they cost the launch of one pair 10 us: tools/sinkhorn_phases.py)
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) K[r][c] = NEG_BIG;
        } else {
            // ragged columns (M not a multiple of 4, or the lane that holds the last valid columns): every load first ...
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float* row = S + (size_t)min(row0 + r, N - 1) * M;
#pragma unroll
                for (int c = 0; c < 8; ++c) K[r][c] = row[min(gcol0 + c, M - 1)];
            }
            // ... then the guard, the scale and the mask
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = K[r][c];
                    gmax = max(gmax, __builtin_bit_cast(unsigned, x) & 0x7fffffffu);
                    K[r][c] = (row0 + r < N && gcol0 + c < M) ? x * MDGAT_LOG2E : NEG_BIG;
                }
        }
        if (a.range_guard && gmax >= 0x7f800000u) __hip_atomic_store(a.range_guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        SK_PH(1);
        // ---- absorb the row maximum (all column slabs, dustbin column included): every row of K has largest entry <= 1 ----
        float u0r = 0.f, kbr = 0.f, ar = 0.f;     // lane r: absorbed potential, dustbin-column entry, scaling of row r
        {
            float mrow = NEG_BIG;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float m = K[r][0];
#pragma unroll
                for (int c = 1; c < 8; ++c) m = fmaxf(m, K[r][c]);
                m = wave_max_dpp(m);                   // wave-uniform
                mrow = (lane == r) ? m : mrow;
            }
            mrow = fmaxf(row_exchange(mrow, true), alpha);
            u0r = -mrow;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mrow), r));
#pragma unroll
                for (int c = 0; c < 8; ++c) K[r][c] = ex2(K[r][c] - m);
            }
        }
        if (my_row_valid) { kbr = ex2(alpha + u0r); ar = 1.f; } else { u0r = 0.f; }
        // Range of the scaling form: an entry more than ~100 octaves below its row's largest - a score 69 below the row
        // maximum, or a bin score that far from it - is on its way out of fp32 (2^-126, sooner after a fold), and a column
        // (or the dustbin column) made of such entries only has no sum left, where the log-domain reference still resolves
        // it.  No trained network is near that; when the inputs are (tools/fuzz_sinkhorn.py: scores spread over 100+ units),
        // the launch is handed to the log-domain streaming kernel: raise the error word.
        {
            float kmin = my_row_valid ? kbr : 1.f;
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    const bool v0_ = row0 + r < N && gcol0 + c < M, v1_ = row0 + r < N && gcol0 + c + 1 < M;
                    kmin = fminf(kmin, fminf(v0_ ? K[r][c] : 1.f, v1_ ? K[r][c + 1] : 1.f));
                }
            if (kmin < 0x1p-100f) {
                atomicOr(a.error_word, 2u);
                __hip_atomic_store(a.pair_flags + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // absorbed dustbin row: u0_N = -alpha, so its entries are exp2(v0_j) = 1
        float u0N = -alpha, aN = 1.f;
        float v0[8], kr[8], b[8];         // per lane column: absorbed potential, dustbin-row entry, column scaling
#pragma unroll
        for (int c = 0; c < 8; ++c) { v0[c] = 0.f; kr[c] = (gcol0 + c < M) ? 1.f : 0.f; b[c] = (gcol0 + c < M) ? 1.f : 0.f; }
        float v0M = 0.f, kc = 1.f, bM = 1.f;
        // state of the columns this thread finalises: local column tid (global 512 jc + tid); thread 0 also the dustbin column
        const bool tcol_valid = jc * 512 + tid < M;
        float krt = 1.f, v0t = 0.f, bt = 1.f;
        SK_PH(2);
        __syncthreads();                  // previous pair's readers of the LDS vectors are done
        if (tid == 0) flags[0] = 0;
        // do all row-slab partners of this pair sit on one XCD?  (slot 513 of the parity-0 buffer, agent scope)
        bool same_xcd = false;
        if (!TWO_D && GR > 1) {
            ++xep;
            if (tid == 0) {
                const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;     // HW_REG_XCC_ID[3:0]
                gu64* xb = cslots + (size_t)jc * GR * SLOT_STRIDE + 513;
                const unsigned xtag = 0x80000000u | xep;
                __hip_atomic_store(xb + (size_t)jr * SLOT_STRIDE, ((unsigned long long)xtag << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float ids[GMAX];
#pragma unroll
                for (int pp = 0; pp < GMAX; ++pp) ids[pp] = 0.f;
                poll_partners<GMAX>(xb, SLOT_STRIDE, GR, jr, xtag, ids, failed, a.error_word);
                bool same = !failed;
#pragma unroll
                for (int pp = 0; pp < GMAX; ++pp)
                    if (pp < GR && pp != jr && __builtin_bit_cast(unsigned, ids[pp]) != my_xcc) same = false;
                flags[1] = same ? 1 : 0;
            }
            __syncthreads();
            same_xcd = flags[1] != 0;
        }

        SK_PH(3);
        for (int it = 0; it < a.iters; ++it) {
            // ---- row update (mdgat.py:283): a_i = mu_i / sum_j K_ij b_j ----
            SK_TP(0);
            float psum = 0.f;
            if (RPW == 16) {
                float racc[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float acc = K[r][0] * b[0];
#pragma unroll
                    for (int c = 1; c < 8; ++c) acc = fmaf(K[r][c], b[c], acc);
                    racc[r] = acc;
                }
                SK_TP(1);
                psum = wave_sum16(racc, lane);                 // lane r: row r (lanes >= 16 are not used)
                psum = lane < 16 ? psum : 0.f;
            } else if (RPW == 8) {
                float racc[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float acc = K[r][0] * b[0];
#pragma unroll
                    for (int c = 1; c < 8; ++c) acc = fmaf(K[r][c], b[c], acc);
                    racc[r] = acc;
                }
                psum = wave_sum8(racc, lane);                  // lane r: row r
                psum = lane < 8 ? psum : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float acc = K[r][0] * b[0];
#pragma unroll
                    for (int c = 1; c < 8; ++c) acc = fmaf(K[r][c], b[c], acc);
                    const float tot = wave_sum_dpp(acc);          // wave-uniform
                    psum = (lane == r) ? tot : psum;
                }
            }
            {
                float pd = kr[0] * b[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) pd = fmaf(kr[c], b[c], pd);
                pd = wave_sum_dpp(pd);                         // dustbin row over this slab's columns
                if (TWO_D) {
                    psum = (lane == RPW) ? pd : psum;
                    psum = row_exchange(psum, false);          // rows of this wave and the dustbin row, over all column slabs
                    pd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, psum), RPW));
                }
                aN = muN * __builtin_amdgcn_rcpf(fmaf(kc, bM, pd));
            }
            ar = my_row_valid ? mu * __builtin_amdgcn_rcpf(fmaf(kbr, bM, psum)) : 0.f;
            const float dsum = wave_sum_dpp(kbr * ar);
            SK_TP(2);
            // ---- column update, wave-local part (mdgat.py:284): sum_i K_ij a_i over this wave's rows ----
            float q[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) q[c] = 0.f;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float arr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ar), r));
#pragma unroll
                for (int c = 0; c < 8; ++c) q[c] = fmaf(K[r][c], arr, q[c]);
            }
            SK_TP(3);
            {
                f32x4* qw = reinterpret_cast<f32x4*>(colp + wave * 512 + col0);
                qw[0] = f32x4{q[0], q[1], q[2], q[3]};
                qw[1] = f32x4{q[4], q[5], q[6], q[7]};
                if (lane == 0) pdust[wave] = dsum;
            }
            SK_TP(4);
            __syncthreads();
            SK_TP(5);
            // ---- merge the 8 waves, exchange with the row-slab partners, new column scalings ----
            ++cep;
            {
                gu64* base = cslots + ((size_t)(cep & 1) * GC + jc) * GR * SLOT_STRIDE;
                const unsigned long long tagbits = (unsigned long long)cep << 32;
                // (b0) wave 7 publishes the dustbin column's partial sum (slot 512) BEFORE the column exchange, so that its
                //      own poll of the partners' dustbin slots after (a) finds them already there
                float dloc = 0.f;
                if (wave == 7) {
                    dloc = pdust[0];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) dloc += pdust[ww];
                    if (GR > 1 && lane == 0) xstore(base + (size_t)jr * SLOT_STRIDE + 512, tagbits | __builtin_bit_cast(unsigned, dloc), same_xcd);
                }
                // (a) every thread: its own column of the slab, all row-slab partners polled concurrently; lane p of the last
                //     wave polls row slab p's dustbin partial in the same round trips (a second poll after this one kept
                //     the whole workgroup waiting at the barrier for one more L2 round trip per iteration)
                bool have_dust = false;
                float dust_in = 0.f;
                if (tcol_valid) {
                    float loc = colp[tid];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) loc += colp[ww * 512 + tid];
                    float vals[GMAX];
#pragma unroll
                    for (int pp = 0; pp < GMAX; ++pp) vals[pp] = 0.f;
                    if (GR > 1) {
                        xstore(base + (size_t)jr * SLOT_STRIDE + tid, tagbits | __builtin_bit_cast(unsigned, loc), same_xcd);
                        SK_TP(6);
                        const bool my_dust = wave == 7 && lane < GR && lane != jr;
                        poll_partners<GMAX>(base + tid, SLOT_STRIDE, GR, jr, cep, vals, failed, a.error_word, same_xcd,
                                            my_dust ? base + (size_t)lane * SLOT_STRIDE + 512 : nullptr, &dust_in);
                        have_dust = my_dust && !failed;
                        SK_TP(7);
                    }
                    float total = 0.f;
#pragma unroll
                    for (int pp = 0; pp < GMAX; ++pp) total += (pp == jr) ? loc : vals[pp];   // fixed order: bit-identical in every partner
                    bt = nu * __builtin_amdgcn_rcpf(fmaf(krt, aN, total));
                    bvec[tid] = bt;
                    if (!(bt > RANGE_LO && bt < RANGE_HI)) flags[0] = 1;
                }
                // (b) wave 7: the dustbin column (slot 512), lane p polls row slab p
                if (wave == 7) {
                    float mine = have_dust ? dust_in : dloc;
                    if (GR > 1) {
                        if (lane < GR && lane != jr && !have_dust) {
                            unsigned spins = 0;
                            while (true) {
                                const unsigned long long x = xload(base + (size_t)lane * SLOT_STRIDE + 512, same_xcd);
                                if ((unsigned)(x >> 32) == cep) { mine = __builtin_bit_cast(float, (unsigned)x); break; }
                                if (failed || ++spins > (1u << 22)) { failed = true; atomicOr(a.error_word, 1u); break; }
                                __builtin_amdgcn_s_sleep(1);
                            }
                        }
                    }
                    float total = 0.f;
#pragma unroll
                    for (int pp = 0; pp < GMAX; ++pp)
                        if (pp < GR) total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), pp));
                    if (lane == 0) bvec[512] = nuM * __builtin_amdgcn_rcpf(fmaf(kc, aN, total));
                }
            }
            SK_TP(8);
            __syncthreads();
            SK_TP(9);
            {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(bvec + col0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(bvec + col0 + 4);
                b[0] = x0[0]; b[1] = x0[1]; b[2] = x0[2]; b[3] = x0[3]; b[4] = x1[0]; b[5] = x1[1]; b[6] = x1[2]; b[7] = x1[3];
#pragma unroll
                for (int c = 0; c < 8; ++c) if (gcol0 + c >= M) b[c] = 0.f;
                bM = bvec[512];
            }
            SK_TP(10);
            // ---- fold scalings that left [2^-40, 2^40] back into K and the absorbed potentials (rare) ----
            // (Measured: with the folds hoisted out of the iteration loop - an inner loop that only reads the block - the
            // 128 register copies per iteration disappear from the N <= 512 kernel, for -1 %; the two larger kernels spill
            // 75 / 195 registers instead of 59 / 49: N = 2048 +12 %.)
            if (flags[0] != 0) {                   // identical in all row-slab partners (same b for this column slab)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (gcol0 + c < M) {
#pragma unroll
                        for (int r = 0; r < RPW; ++r) K[r][c] *= b[c];
                        v0[c] += lg2(b[c]);
                        kr[c] *= b[c];
                        b[c] = 1.f;
                    }
                }
                if (tcol_valid) { krt *= bt; v0t += lg2(bt); bt = 1.f; }
                __syncthreads();                   // everyone has read the flag and bvec
                if (tid == 0) flags[0] = 0;
            }
            if (!(bM > RANGE_LO && bM < RANGE_HI)) {      // uniform everywhere: bM is replicated bit-identically
                kbr *= bM;
                kc *= bM;
                v0M += lg2(bM);
                bM = 1.f;
            }
            if (!(aN > RANGE_LO && aN < RANGE_HI)) {      // uniform everywhere
#pragma unroll
                for (int c = 0; c < 8; ++c) kr[c] *= aN;
                krt *= aN;
                kc *= aN;
                u0N += lg2(aN);
                aN = 1.f;
            }
            {
                const bool fold = my_row_valid && !(ar > RANGE_LO && ar < RANGE_HI);   // identical in all column-slab partners
                const unsigned long long fm = __ballot(fold);
                if (fm) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        if (fm & (1ull << r)) {
                            const float arr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ar), r));
#pragma unroll
                            for (int c = 0; c < 8; ++c) K[r][c] *= arr;
                        }
                    }
                    if (fold) { kbr *= ar; u0r += lg2(ar); ar = 1.f; }
                }
            }
        }

        SK_PH(4);
        // ---- Z = couplings + u + v - norm (mdgat.py:285, 307), natural-log units; fused arg-max ----
        float* Zp = a.Z ? a.Z + (size_t)pair * (N + 1) * (M + 1) : nullptr;
        const bool partner_lost = (__hip_atomic_load(a.error_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0 ||
                                  __hip_atomic_load(a.pair_flags + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // (or: this pair out of range)
        const float poison = partner_lost ? __builtin_nanf("") : 0.f;   // a partner never arrived: whatever this launch writes is
                                                                       // overwritten by the gated streaming kernel that follows
        const bool ran = a.iters > 0;    // with zero iterations u = v = 0 (the absorbed potentials are not potentials)
        const float VM = ran ? v0M + lg2(bM) + poison : 0.f;
        const float Ur = (ran && my_row_valid) ? u0r + lg2(ar) : 0.f;    // lane r: potential of row r
        float V[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) V[c] = (ran && gcol0 + c < M) ? v0[c] + lg2(b[c]) : 0.f;
        // Z (base-2) = s + U + V with s = lg2(K) - u0 - v0 from the register block: the scores are not read a second time
        // (67 MB per launch at B = 64).  K = exp2(s + u0 + v0) <= 1 carries s to ~1e-7; an entry that underflowed to 0 OR INTO THE
        // DENORMALS (v_log_f32 returns -inf for a denormal: found by tools/fuzz_forward.py on scores spanning 100 units)
        // (more than 126 octaves below its row maximum) is re-read.  Only where the block's registers have room for it:
        // in the kernels for more than 512 keypoints the longer live range of K costs 12 more spilled registers inside
        // the iteration loop (N = 2048: 7.5 -> 9.1 us per iteration), so they read the scores again instead.
        constexpr bool FROM_K = !TWO_D && GMAX == 4;
        const float dUr = Ur - u0r;                   // lane r
        float dV[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) dV[c] = V[c] - v0[c];
        const bool ext = a.ext_mode >= 0;
        const bool inner = a.ext_mode >= MDGAT_EXTRACT_THRESHOLD;   // arg-max over the inner N x M block only
        const bool last_c = jc == GC - 1, last_r = jr == GR - 1;
        float cbv[8];
        int cbi[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { cbv[c] = -__builtin_inff(); cbi[c] = 0; }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int i = row0 + r;
            if (i < N) {
                const float U = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Ur), r));
                const float dU = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dUr), r));
                const float* row = S + (size_t)i * M;
                float z[8];
                if (FROM_K) {
                    bool under = false;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        under |= (gcol0 + c < M) && !(K[r][c] >= 1.17549435e-38f);
                        z[c] = (gcol0 + c < M) ? (lg2(K[r][c]) + dU + dV[c]) * MDGAT_LN2 - norm : -__builtin_inff();
                    }
                    if (__any(under)) {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if ((gcol0 + c < M) && !(K[r][c] >= 1.17549435e-38f)) z[c] = (row[gcol0 + c] * MDGAT_LOG2E + U + V[c]) * MDGAT_LN2 - norm;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        z[c] = (gcol0 + c < M) ? (row[min(gcol0 + c, M - 1)] * MDGAT_LOG2E + U + V[c]) * MDGAT_LN2 - norm : -__builtin_inff();
                }
                const float zM = (alpha + U + VM) * MDGAT_LN2 - norm;
                if (Zp) {
                    float* zr = Zp + (size_t)i * (M + 1);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (gcol0 + c < M) zr[gcol0 + c] = z[c];
                    if (last_c && lane == 0) zr[M] = zM;
                }
                if (ext) {
                    // row: first maximal column of this slab (torch.max); the last slab adds the dustbin column unless `inner`
                    float bv = z[0];
                    int bi = gcol0;
#pragma unroll
                    for (int c = 1; c < 8; ++c)
                        if (z[c] > bv) { bv = z[c]; bi = gcol0 + c; }
                    {
                        // wave maximum on the vector ALU (DPP), then the LOWEST lane that holds it (lanes are in column order
                        // and bi is the lane's own first maximum: torch.max's first maximal index) - one ballot and two
                        // readlanes instead of twelve dependent ds_bpermute per row
                        const float mx = wave_max_dpp(bv);
                        const unsigned long long hit = __ballot(bv == mx);
                        const int first = hit ? __builtin_ctzll(hit) : 0;        // (no lane: NaN-poisoned row)
                        bi = __builtin_amdgcn_readlane(bi, first);
                        bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), first));
                    }
                    if (!inner && last_c && zM > bv) { bv = zM; bi = M; }
                    if (lane == 0) {
                        a.rbest_idx[((size_t)pair * GC + jc) * N + i] = bi;
                        a.rbest_val[((size_t)pair * GC + jc) * N + i] = bv;
                    }
                    // columns: first maximal row among this wave's rows (ascending, strict compare)
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (z[c] > cbv[c]) { cbv[c] = z[c]; cbi[c] = i; }
                }
            }
        }
        SK_PH(5);
        const float UN = ran ? u0N + lg2(aN) : 0.f;
        const float zNt = (alpha + UN + (ran ? v0t + lg2(bt) + poison : 0.f)) * MDGAT_LN2 - norm;   // Z[N][512 jc + tid]
        if (last_r && Zp) {
            float* zl = Zp + (size_t)N * (M + 1);
            if (tcol_valid) zl[jc * 512 + tid] = zNt;
            if (last_c && tid == 0) zl[M] = (alpha + UN + VM) * MDGAT_LN2 - norm;
        }
        if (ext) {
            // merge the 8 waves (ascending rows), add the dustbin row (last row slab only, not `inner`)
            __syncthreads();             // colp's readers of the last iteration are done
            {
                f32x4* qw = reinterpret_cast<f32x4*>(colp + wave * 512 + col0);
                qw[0] = f32x4{cbv[0], cbv[1], cbv[2], cbv[3]};
                qw[1] = f32x4{cbv[4], cbv[5], cbv[6], cbv[7]};
                int* qi = colpi + wave * 512 + col0;
#pragma unroll
                for (int c = 0; c < 8; ++c) qi[c] = cbi[c];
            }
            __syncthreads();
            if (tcol_valid) {
                float bv = colp[tid];
                int bi = colpi[tid];
#pragma unroll
                for (int ww = 1; ww < 8; ++ww) {
                    const float v = colp[ww * 512 + tid];
                    if (v > bv) { bv = v; bi = colpi[ww * 512 + tid]; }
                }
                if (!inner && last_r && zNt > bv) { bv = zNt; bi = N; }
                a.cbest_val[((size_t)pair * GR + jr) * M + jc * 512 + tid] = bv;
                a.cbest_idx[((size_t)pair * GR + jr) * M + jc * 512 + tid] = bi;
            }
        }
        SK_PH(6);
    }
}

// ------------------------------------------------------------------------------------------------
// match extraction: one workgroup per pair
struct ExArgs {
    const float* Z;
    int N, M, mode;
    float thr;
    int64_t* m0; int64_t* m1;
    float* s0; float* s1;
    const unsigned* sk_error;   // optional: error word of the cluster kernel that produced the arg-maxes (bit 0: a lost partner) ...
    const float* Zfb;           // ... in which case the streaming fallback has written Z here: scan it instead
    const unsigned* pair_flags; // ... or, with sk_error, this pair alone was redone by the streaming kernel ([B])
    // optional, zeroed before the launch (B < 65536): one ticket word, (workgroups whose pair matched anything) << 16 |
    // workgroups done.  The last workgroup to finish applies the batch-wide rule of mdgat.py:465-467 (nothing matched
    // anywhere: all scores zero) - no separate fix-up launch
    unsigned* alldust_counters;
    int B;
    // Z == NULL: the arg-maxes were computed by the Sinkhorn kernel (row bests per column slab [B][GC][N], column bests
    // per row slab [B][GR][M])
    const int* rbest_idx; const float* rbest_val; const int* cbest_idx; const float* cbest_val; int GR, GC;
    // optional (host-mapped word): set to matched_token when a frame-0 keypoint of this pair is matched - what the host-side test
    // of mdgat.py:465 (`valid0.sum() == 0`) needs to know, without a reduction kernel and a copy (mdgat_matched_any)
    unsigned* matched; unsigned matched_token;
};

__global__ __launch_bounds__(1024) void extract_kernel(ExArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, M = a.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the cluster kernel lost a partner workgroup (bounded spin ran out): its fused arg-maxes are garbage; the gated streaming
    // kernel has recomputed Z since
    const bool redone = a.sk_error && ((__hip_atomic_load(a.sk_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0 ||
                                       (a.pair_flags && __hip_atomic_load(a.pair_flags + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0));
    const float* Zsrc = redone ? a.Zfb : a.Z;
    const float* Z = Zsrc + (size_t)blockIdx.x * (N + 1) * (M + 1);
    int* idx0 = reinterpret_cast<int*>(smem);   // [N]
    int* idx1 = idx0 + N;                       // [M]
    float* val0 = reinterpret_cast<float*>(idx1 + M);   // [N]
    float* val1 = val0 + N;                     // [M]
    const bool inner = a.mode >= MDGAT_EXTRACT_THRESHOLD;   // arg-max over the inner N x M block only
    const int ncol = inner ? M : M + 1;   // columns scanned per row
    const int nrow = inner ? N : N + 1;   // rows scanned per column
    if (!Zsrc) {
        for (int i = tid; i < N; i += 1024) {
            const size_t base = (size_t)blockIdx.x * a.GC * N + i;
            float bv = a.rbest_val[base];
            int bi = a.rbest_idx[base];
            for (int g = 1; g < a.GC; ++g) {         // ascending column slabs, strict compare: first maximal column
                const float v = a.rbest_val[base + (size_t)g * N];
                if (v > bv) { bv = v; bi = a.rbest_idx[base + (size_t)g * N]; }
            }
            idx0[i] = bi; val0[i] = bv;
        }
        for (int j = tid; j < M; j += 1024) {
            const size_t base = (size_t)blockIdx.x * a.GR * M + j;
            float bv = a.cbest_val[base];
            int bi = a.cbest_idx[base];
            for (int g = 1; g < a.GR; ++g) {         // ascending row slabs, strict compare: first maximal row
                const float v = a.cbest_val[base + (size_t)g * M];
                if (v > bv) { bv = v; bi = a.cbest_idx[base + (size_t)g * M]; }
            }
            idx1[j] = bi; val1[j] = bv;
        }
    } else {
    // rows: first maximal index (torch.max semantics)
    for (int i = wave; i < N; i += 16) {
        const float* zr = Z + (size_t)i * (M + 1);
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int j = lane; j < ncol; j += 64) {
            const float z = zr[j];
            if (z > bv) { bv = z; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { idx0[i] = bi; val0[i] = bv; }
    }
    // columns
    for (int j = tid; j < M; j += 1024) {
        float bv = -__builtin_inff();
        int bi = 0;
        for (int i = 0; i < nrow; ++i) {
            const float z = Z[(size_t)i * (M + 1) + j];
            if (z > bv) { bv = z; bi = i; }
        }
        idx1[j] = bi; val1[j] = bv;
    }
    }
    __syncthreads();

    int64_t* m0 = a.m0 + (size_t)blockIdx.x * N;
    int64_t* m1 = a.m1 + (size_t)blockIdx.x * M;
    float* s0 = a.s0 + (size_t)blockIdx.x * N;
    float* s1 = a.s1 + (size_t)blockIdx.x * M;

    int nmatch = 0;          // frame-0 keypoints of this thread with a match (matches0 >= 0)
    if (a.mode == MDGAT_EXTRACT_DUSTBIN || a.mode == MDGAT_EXTRACT_DUSTBIN_MUTUAL) {
        const bool mutual = a.mode == MDGAT_EXTRACT_DUSTBIN_MUTUAL;
        int nvalid = 0;
        for (int i = tid; i < N; i += 1024) {
            const int j = idx0[i];
            const bool valid = j < M;
            const bool keep = valid && (!mutual || idx1[j] == i);
            m0[i] = valid ? j : -1;
            s0[i] = keep ? expf(val0[i]) : 0.f;
            nvalid += valid;
        }
        for (int j = tid; j < M; j += 1024) {
            const int i = idx1[j];
            const bool valid = i < N;
            const bool keep = valid && (!mutual || idx0[i] == j);
            m1[j] = valid ? i : -1;
            s1[j] = keep ? expf(val1[j]) : 0.f;
        }
        nmatch = nvalid;
        if (a.alldust_counters) {
            // mdgat.py:465-467 over the whole batch: one ticket word = (workgroups that matched anything) << 16 | workgroups done;
            // the last workgroup to arrive decides
            const int any_valid = __syncthreads_or(nvalid > 0);         // (also: this workgroup's stores are issued)
            __shared__ int last;
            if (tid == 0) {
                __threadfence();                                         // scores of this pair before the ticket
                const unsigned old = atomicAdd(a.alldust_counters, (any_valid ? 0x10000u : 0u) + 1u);
                last = (old & 0xffffu) == (unsigned)a.B - 1 && (old >> 16) == 0 && !any_valid;
                __threadfence();
            }
            __syncthreads();
            if (last) {                                                  // (s0 is zero already: no row was valid)
                float* s1all = a.s1;
                for (size_t i = tid; i < (size_t)a.B * M; i += 1024) s1all[i] = 0.f;
            }
        }
    } else if (a.mode == MDGAT_EXTRACT_THRESHOLD) {
        for (int i = tid; i < N; i += 1024) {
            const float e = expf(val0[i]);
            const bool valid = e > a.thr;
            m0[i] = valid ? idx0[i] : -1;
            s0[i] = valid ? e : 0.f;
            nmatch += valid;
        }
        for (int j = tid; j < M; j += 1024) {
            const float e = expf(val1[j]);
            const bool valid = e > a.thr;
            m1[j] = valid ? idx1[j] : -1;
            s1[j] = valid ? e : 0.f;
        }
    } else {   // MDGAT_EXTRACT_THRESHOLD_MUTUAL (mdgat.py:447-453)
        for (int i = tid; i < N; i += 1024) {
            const bool mutual0 = idx1[idx0[i]] == i;
            const float ms0 = mutual0 ? expf(val0[i]) : 0.f;
            const bool valid0 = mutual0 && ms0 > a.thr;
            m0[i] = valid0 ? idx0[i] : -1;
            s0[i] = ms0;
            nmatch += valid0;
        }
        for (int j = tid; j < M; j += 1024) {
            const int i = idx1[j];
            const bool mutual1 = idx0[i] == j;
            const bool mutual0_i = idx1[idx0[i]] == i;
            const float ms0_i = mutual0_i ? expf(val0[i]) : 0.f;
            const float ms1 = mutual1 ? ms0_i : 0.f;
            const bool valid1 = mutual1 && (mutual0_i && ms0_i > a.thr);
            m1[j] = valid1 ? i : -1;
            s1[j] = ms1;
        }
    }
    if (a.matched) {
        const int any = __syncthreads_or(nmatch > 0);
        if (any && tid == 0) __hip_atomic_store(a.matched, a.matched_token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// mdgat.py:465-467: when NO frame-0 keypoint of the whole batch is matched (valid0.sum() == 0), both score vectors are
// zeros.  matches0 >= 0 <=> valid0 in both dustbin modes, so the written matches are scanned (no counter shared between
// launches): every workgroup stops at the first chunk that holds a match - the first one on real data.
__global__ __launch_bounds__(256) void extract_alldust_fixup(const int64_t* m0, size_t n0, float* s1, size_t n1) {
    for (size_t base = 0; base < n0; base += 256) {
        const size_t i = base + threadIdx.x;
        if (__syncthreads_or(i < n0 && m0[i] >= 0)) return;
    }
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n1; i += (size_t)gridDim.x * 256) s1[i] = 0.f;
}

template <int NC, int NW>
int launch_sk(const SkArgs& a, int B, hipStream_t s) {
    const size_t lds = ((size_t)(a.N + 1) + (a.M + 1) + 2 * (size_t)NW * a.M) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_kernel<NC, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mdgat_check_hip(e, "sinkhorn LDS attribute");
    hipLaunchKernelGGL((sinkhorn_kernel<NC, NW>), dim3(B), dim3(NW * 64), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "sinkhorn launch");
}

}  // namespace

// rows per wave for launches of few pairs (see sk_rpw); 16 = the throughput shape everywhere until measured otherwise
static int mdgat_sk_small_rpw(int B, int N, int M) { (void)B; (void)N; (void)M; return 16; }
// tiling of a pair: GR row slabs of 8 RPW rows (RPW rows per wave: 16) x GC column slabs of 512 columns, one workgroup each
static void sk_tiling(int N, int M, int& GR, int& GC, int rpw = 16) { GR = (N + 8 * rpw - 1) / (8 * rpw); GC = (M + 511) / 512; }
// Rows per wave of a launch.  16 (128-row slabs) is the throughput shape.  Few pairs leave most of the part idle - one pair of
// 512 keypoints is 4 workgroups on 256 CUs - and thinner slabs (RPW = 8, 4: 8 / 16 workgroups per pair, an iteration's
// issue-bound row and column passes 2x / 4x shorter, 7 / 15 partners to poll instead of 3) trade that idleness for exchange:
// MDGAT_SK_RPW (measurements) forces a value where the shape allows it (one column slab, at most 16 row slabs).
static int sk_rpw(int B, int N, int M) {
    static const int forced = [] { const char* e = getenv("MDGAT_SK_RPW"); return e ? atoi(e) : 0; }();
    int rpw = 16;
    if (forced == 4 || forced == 8) rpw = forced;
    else if (forced == 0) rpw = mdgat_sk_small_rpw(B, N, M);
    if (rpw != 16 && (M > 512 || (N + 8 * rpw - 1) / (8 * rpw) > 16)) rpw = 16;
    return rpw;
}
// (sizes are taken for the thinnest slabs a shape may run with: the layout inside is the launch's own)
static int sk_max_groups(int N, int M, int rpw = 16) {      // pairs in flight on a 256-CU part (upper bound used for sizing)
    int GR, GC;
    sk_tiling(N, M, GR, GC, rpw);
    int g = 256 / (GR * GC);
    return g > 64 ? 64 : (g < 1 ? 1 : g);
}
static size_t slots_bytes_rpw(int N, int M, int rpw) {
    int GR, GC;
    sk_tiling(N, M, GR, GC, rpw);
    const size_t per_group = ((size_t)2 * GC * GR * SLOT_STRIDE + (size_t)2 * GR * GC * ROW_STRIDE) * sizeof(unsigned long long);
    return (256 + per_group * sk_max_groups(N, M, rpw) + 255) & ~(size_t)255;
}
static size_t slots_bytes(int N, int M) {
    size_t b = slots_bytes_rpw(N, M, 16);
    for (int rpw : {8, 4})
        if (M <= 512 && (N + 8 * rpw - 1) / (8 * rpw) <= 16) { const size_t x = slots_bytes_rpw(N, M, rpw); b = x > b ? x : b; }
    return b;
}
static size_t flags_bytes(int B) { return ((size_t)B * sizeof(unsigned) + 255) & ~(size_t)255; }
size_t sinkhorn_cluster_workspace_bytes(int B, int N, int M) {
    if (N > 2048 || M > 2048) return 0;
    int GR, GC;
    sk_tiling(N, M, GR, GC);
    if (M <= 512 && (N + 31) / 32 <= 16) GR = (N + 31) / 32;       // (the thinnest slabs the shape may run with: sk_rpw)
    // exchange slots + per-pair range flags + fused arg-max scratch: row bests [B][GC][N] (int + float), column bests [B][GR][M] (int + float)
    return slots_bytes(N, M) + flags_bytes(B) + ((size_t)B * GC * N * 2 + (size_t)B * GR * M * 2) * sizeof(float);
}

static int launch_extract_impl(int B, int N, int M, ExArgs a, hipStream_t s, bool defer_alldust = false);

// the one-workgroup-per-pair streaming kernel: any shape up to M = 2048, no workspace
static bool streaming_supported(int N, int M) { return M <= 512 || (M <= 2048 && N <= 4096); }
static int launch_streaming(const SkArgs& a, int B, hipStream_t s) {
    const int N = a.N, M = a.M;
    if (M <= 64) return launch_sk<1, 16>(a, B, s);
    if (M <= 128) return launch_sk<2, 16>(a, B, s);
    if (M <= 256) return launch_sk<4, 16>(a, B, s);
    if (M <= 512) return launch_sk<8, 16>(a, B, s);
    if (M <= 1024 && N <= 4096) return launch_sk<16, 8>(a, B, s);
    if (M <= 2048 && N <= 4096) return launch_sk<32, 8>(a, B, s);
    mdgat_set_error("sinkhorn: M=%d > 2048 unsupported", M);
    return MDGAT_ERR_UNSUPPORTED;
}

size_t sinkhorn_slots_clear_bytes(int B, int N, int M) { return (N > 2048 || M > 2048) ? 0 : slots_bytes(N, M) + flags_bytes(B); }

template <int RPW>
static int launch_scaling(int B, int N, int M, const float* scores, const float* alpha_dev, float alpha_host, int iters,
                          float* Z, void* ws, int num_cu, const SkExtract* ex, unsigned* status, float* Zfb, bool slots_cleared, hipStream_t s) {
    int GR, GC;
    sk_tiling(N, M, GR, GC, RPW);
    const int P = GR * GC;                         // workgroups per pair, one per CU
    int ngroups = num_cu / P;
    if (ngroups > sk_max_groups(N, M, RPW)) ngroups = sk_max_groups(N, M, RPW);
    if (ngroups > B) ngroups = B;
    // Placement: the workgroups of a pair exchange through L2 every iteration, which is fast and steady only inside one XCD
    // (workgroup i runs on XCD i % 8): group g takes the workgroups with blockIdx % 8 == g % 8, and the grid is padded to a
    // multiple of 8 groups - the surplus workgroups leave at once.  (Before: only batches that are multiples of 8 were placed,
    // others ran their partners on four different XCDs - 1.0 ms against an erratic 1.0 ... 3.5 ms per forward at B = 2 ... 4 -
    // and a batch of 12 ran as 8 + 4.)  A pair's workgroups must fit the 32 CUs of an XCD next to the other groups placed there.
    static const bool cooperative = getenv("MDGAT_SK_COOPERATIVE") != nullptr;
    const int ng8 = (ngroups + 7) & ~7;
    const bool xcd_map = P * (ng8 / 8) <= num_cu / 8 && (!cooperative || ngroups == ng8);
    const int grid = xcd_map ? ng8 * P : ngroups * P;
    if (ngroups < 1) { mdgat_set_error("sinkhorn: %d workgroups per pair do not fit the device", P); return MDGAT_ERR_UNSUPPORTED; }
    const size_t per_group = ((size_t)2 * GC * GR * SLOT_STRIDE + (size_t)2 * GR * GC * ROW_STRIDE) * sizeof(unsigned long long);
    if (!slots_cleared) {    // (the forward has the score kernel clear them)
        if (int rc = mdgat_check_hip(hipMemsetAsync(ws, 0, 256 + per_group * ngroups, s), "memset(sinkhorn slots)")) return rc;
        if (int rc = mdgat_check_hip(hipMemsetAsync(static_cast<char*>(ws) + slots_bytes(N, M), 0, flags_bytes(B), s), "memset(sinkhorn pair flags)")) return rc;
    }
    // test hook (tests/test_gpu_ops.py): pretend a partner was lost - the launch's error word starts out set, so the gated
    // streaming kernel and the extraction from its Z run for real
    if (const char* f = getenv("MDGAT_SK_FORCE_FALLBACK"); f && *f == '1')
        if (int rc = mdgat_check_hip(hipMemsetAsync(ws, 1, sizeof(unsigned), s), "memset(sinkhorn error word)")) return rc;
    SksArgs a{scores, alpha_dev, alpha_host, Z, reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + 256),
              static_cast<unsigned*>(ws), reinterpret_cast<unsigned*>(static_cast<char*>(ws) + slots_bytes(N, M)),
              status ? status + MDGAT_STATUS_RANGE : nullptr, B, N, M, iters, ngroups, GR, GC, xcd_map ? 1 : 0, -1, nullptr, nullptr, nullptr, nullptr};
    if (ex) {
        char* p = static_cast<char*>(ws) + slots_bytes(N, M) + flags_bytes(B);
        a.ext_mode = ex->mode;
        a.rbest_idx = reinterpret_cast<int*>(p);                      p += (size_t)B * GC * N * sizeof(int);
        a.rbest_val = reinterpret_cast<float*>(p);                    p += (size_t)B * GC * N * sizeof(float);
        a.cbest_idx = reinterpret_cast<int*>(p);                      p += (size_t)B * GR * M * sizeof(int);
        a.cbest_val = reinterpret_cast<float*>(p);
    }
    void* args[] = {&a};
    const void* kern;
    if constexpr (RPW == 16)
        kern = GC > 1 ? reinterpret_cast<const void*>(sinkhorn_scaling_kernel<RPW, true, 16>)
             : GR > 4 ? reinterpret_cast<const void*>(sinkhorn_scaling_kernel<RPW, false, 16>)
                      : reinterpret_cast<const void*>(sinkhorn_scaling_kernel<RPW, false, 4>);
    else kern = reinterpret_cast<const void*>(sinkhorn_scaling_kernel<RPW, false, 16>);      // (one column slab, up to 16 row slabs: sk_rpw)
    // The workgroups of a pair wait for each other, so all of them must become resident.  A workgroup takes a whole CU
    // (512 threads x 256 registers), and ngroups * P <= num_cu by construction: every workgroup gets a CU as soon as the
    // stragglers of earlier launches leave.  A plain launch therefore suffices when this launch has the device to itself, and
    // it starts 20-30 us sooner than hipLaunchCooperativeKernel (measured at B = 64: 430 -> 397 us for launch + 100
    // iterations).  It is NOT assumed: workgroups are dispatched in order (per XCD), so of every concurrent cluster launch at
    // most one pair per XCD is incomplete and the complete ones finish and make room - but enough concurrent launches (other
    // streams, other processes) could leave every CU with a workgroup whose partners cannot be dispatched.  The spins are
    // therefore bounded, a workgroup that gives up raises the launch's error word, and the streaming kernel launched right
    // behind (gated on that word: it leaves at once otherwise, ~3 us) redoes the launch one workgroup per pair - slow, never
    // wrong.  Without a fallback buffer (no Z and none lent) or with MDGAT_SK_COOPERATIVE=1 the launch is cooperative: the
    // runtime then checks co-residency.
    float* zfb = Z ? Z : Zfb;
    const bool can_fall_back = zfb != nullptr && streaming_supported(N, M);
    hipError_t e;
    if (cooperative || !can_fall_back || ngroups * P > num_cu) e = hipLaunchCooperativeKernel(kern, dim3(grid), dim3(SKS_THREADS), args, 0, s);
    else e = hipLaunchKernel(kern, dim3(grid), dim3(SKS_THREADS), args, 0, s);
    if (int rc = mdgat_check_hip(e, "sinkhorn scaling launch")) return rc;
    if (can_fall_back) {
        SkArgs f{scores, alpha_dev, alpha_host, zfb, N, M, iters, a.error_word, status ? status + MDGAT_STATUS_SK_FALLBACK : nullptr, a.pair_flags};
        if (int rc = launch_streaming(f, B, s)) return rc;
    }
    if (ex) {
        // (header words 2, 3 of the workspace: the all-dustbin counters of the extraction, cleared with the slots)
        ExArgs x{nullptr, N, M, ex->mode, ex->thr, ex->m0, ex->m1, ex->s0, ex->s1, can_fall_back ? a.error_word : nullptr, zfb, a.pair_flags,
                 (ex->defer_alldust || B >= 65536) ? nullptr : a.error_word + 2, B,
                 a.rbest_idx, a.rbest_val, a.cbest_idx, a.cbest_val, GR, GC, ex->matched, ex->matched_token};
        return launch_extract_impl(B, N, M, x, s, ex->defer_alldust != 0);
    }
    return MDGAT_OK;
}

size_t mdgat_sinkhorn_ws_bytes_impl(int B, int N, int M) { return sinkhorn_cluster_workspace_bytes(B, N, M); }

// ex != NULL: also extract the matches.  With the cluster kernel the arg-maxes are fused into its epilogue and Z may
// be NULL; otherwise Z must be given and is scanned by the extraction kernel.
int launch_sinkhorn(int B, int N, int M, const float* scores, const float* bin_score_dev, float bin_score_host,
                    int iters, float* Z, void* ws, size_t ws_bytes, const SkExtract* ex, hipStream_t s, unsigned* status, float* Zfb,
                    bool slots_cleared) {
    if (B <= 0) return MDGAT_OK;
    if (!Z && !ex) { mdgat_set_error("sinkhorn: nothing to compute (no Z, no extraction)"); return MDGAT_ERR_BAD_ARG; }
    if (N <= 0 || M <= 0 || iters < 0) { mdgat_set_error("sinkhorn: bad shape N=%d M=%d iters=%d", N, M, iters); return MDGAT_ERR_BAD_ARG; }
    const size_t need = sinkhorn_cluster_workspace_bytes(B, N, M);
    if (need && ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 255) == 0) {
        int dev = 0, num_cu = 0;
        if (int rc = mdgat_check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
        if (int rc = mdgat_check_hip(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev), "CU count")) return rc;
        const int rpw = sk_rpw(B, N, M);
        if (rpw == 4) return launch_scaling<4>(B, N, M, scores, bin_score_dev, bin_score_host, iters, Z, ws, num_cu, ex, status, Zfb, slots_cleared, s);
        if (rpw == 8) return launch_scaling<8>(B, N, M, scores, bin_score_dev, bin_score_host, iters, Z, ws, num_cu, ex, status, Zfb, slots_cleared, s);
        return launch_scaling<16>(B, N, M, scores, bin_score_dev, bin_score_host, iters, Z, ws, num_cu, ex, status, Zfb, slots_cleared, s);
    }
    if (!Z) { mdgat_set_error("sinkhorn: the streaming kernel needs a Z buffer"); return MDGAT_ERR_BAD_ARG; }
    SkArgs a{scores, bin_score_dev, bin_score_host, Z, N, M, iters, nullptr, nullptr, nullptr};
    const int rc = launch_streaming(a, B, s);
    if (rc || !ex) return rc;
    ExArgs xa{Z, N, M, ex->mode, ex->thr, ex->m0, ex->m1, ex->s0, ex->s1, nullptr, nullptr, nullptr, nullptr, B, nullptr, nullptr, nullptr, nullptr, 1, 1,
              ex->matched, ex->matched_token};
    return launch_extract_impl(B, N, M, xa, s, ex->defer_alldust != 0);
}

int launch_alldust_fixup(int B, int N, int M, int mode, const int64_t* m0, float* s1, hipStream_t s) {
    if (mode != MDGAT_EXTRACT_DUSTBIN && mode != MDGAT_EXTRACT_DUSTBIN_MUTUAL) return MDGAT_OK;
    const size_t n = (size_t)B * M;
    hipLaunchKernelGGL(extract_alldust_fixup, dim3((unsigned)((n + 255) / 256 < 16 ? (n + 255) / 256 : 16)), dim3(256), 0, s,
                       m0, (size_t)B * N, s1, n);
    return mdgat_check_hip(hipGetLastError(), "extract fixup launch");
}

static int launch_extract_impl(int B, int N, int M, ExArgs a, hipStream_t s, bool defer_alldust) {
    if (a.mode < 0 || a.mode > 3) { mdgat_set_error("extract: bad mode %d", a.mode); return MDGAT_ERR_BAD_ARG; }
    const size_t lds = (size_t)(2 * (N + M) + 4) * sizeof(float);
    hipLaunchKernelGGL(extract_kernel, dim3(B), dim3(1024), lds, s, a);
    if (int rc = mdgat_check_hip(hipGetLastError(), "extract launch")) return rc;
    if (defer_alldust || a.alldust_counters) return MDGAT_OK;      // (counters: the kernel applied the rule itself)
    return launch_alldust_fixup(B, N, M, a.mode, a.m0, a.s1, s);
}

// the extraction from arg-maxes another kernel has decided (sinkhorn_f64.hip: on the fp64 Z): rbest [B][N], cbest [B][M]
int launch_extract_from_bests(int B, int N, int M, const SkExtract* ex, const int* rbest_idx, const float* rbest_val, const int* cbest_idx,
                              const float* cbest_val, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    ExArgs xa{nullptr, N, M, ex->mode, ex->thr, ex->m0, ex->m1, ex->s0, ex->s1, nullptr, nullptr, nullptr, nullptr, B, rbest_idx, rbest_val, cbest_idx, cbest_val,
              1, 1, ex->matched, ex->matched_token};
    return launch_extract_impl(B, N, M, xa, s, ex->defer_alldust != 0);
}

int launch_extract(int B, int N, int M, const float* Z, int mode, float thr, int64_t* m0, int64_t* m1, float* s0,
                   float* s1, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    ExArgs a{Z, N, M, mode, thr, m0, m1, s0, s1, nullptr, nullptr, nullptr, nullptr, B, nullptr, nullptr, nullptr, nullptr, 1, 1, nullptr, 0u};
    return launch_extract_impl(B, N, M, a, s);
}
