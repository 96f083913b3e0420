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

namespace {

constexpr float NEG_BIG = -1.0e30f;   // finite stand-in for -inf (keeps a - b well defined)

struct SkArgs {
    const float* scores;      // [B][N][M]
    const float* alpha_dev;   // device scalar or nullptr
    float alpha_host;
    float* Z;                 // [B][N+1][M+1]
    int N, M, iters;
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
// Register-resident cluster kernel (N, M <= 512): the pair's score block never leaves the register file.
//
// G workgroups (1024 threads = 16 waves each, one per CU, co-resident: cooperative launch) share one pair:
// workgroup j owns rows [128 j, 128 j + 128), wave w of it 8 of those rows, lane l the 8 contiguous columns
// 8 l .. 8 l + 7 of each - a lane holds an 8 x 8 block of S (64 VGPRs) for all iterations.
//   row update   : per row, 8 in-lane terms + a 6-step DPP wave reduction for max and for sum-of-exp2;
//   column update: per column, in-lane over the wave's 8 rows -> (max, sum) pairs merged across the 16
//                  waves through LDS -> ONE float per column and workgroup (its local log-sum-exp) is
//                  exchanged with the G-1 partner workgroups through L2 as 8-byte {epoch, value} granules
//                  (single relaxed agent-scope 8-byte stores/loads: the data is the flag, no fence; two
//                  slot sets alternate by epoch parity) and merged with the closed-form dustbin terms.
// The blockIdx -> (pair group, j) map keeps the G partners on one XCD (blockIdx % 8), a pure speed choice:
// the protocol is placement independent.  Spins are bounded; a timeout poisons the error word.
struct SkcArgs {
    const float* scores;
    const float* alpha_dev;
    float alpha_host;
    float* Z;
    unsigned long long* slots;   // [ngroups][2][G][SLOT_STRIDE] granules, zeroed before every launch
    unsigned* error_word;
    int B, N, M, iters, ngroups;
};

constexpr int SLOT_STRIDE = 520;
constexpr int CL_WAVES = 8;                 // waves per workgroup (2 per SIMD: 256-VGPR budget)
constexpr int CL_THREADS = CL_WAVES * 64;
constexpr int CL_RPW = 128 / CL_WAVES;      // rows per wave
constexpr int CLUSTER_LDS_FLOATS = 516 + 2 * CL_WAVES * 512 + CL_WAVES + 4;

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

typedef __attribute__((address_space(1))) unsigned long long gu64;

template <int G>
__global__ __launch_bounds__(CL_THREADS) void sinkhorn_cluster_kernel(SkcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // CLUSTER_LDS_FLOATS
    float* v = lds;                       // [M+1] (<= 513), padded to 516
    float* pm = lds + 516;                // [CL_WAVES][512] per-wave column maxima
    float* ps = pm + CL_WAVES * 512;      // [CL_WAVES][512] per-wave column sums
    float* pu = ps + CL_WAVES * 512;      // [CL_WAVES] per-wave LSE of the row potentials
    float* misc = pu + CL_WAVES;          // [0] = u_N

    const int N = a.N, M = a.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int group, j;
    if ((a.ngroups & 7) == 0) {           // partners share blockIdx % 8 (observed: the XCD)
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        j = q % G;
        group = (q / G) * 8 + xcd;
    } else {
        group = blockIdx.x / G;
        j = blockIdx.x % G;
    }
    const float alpha = (a.alpha_dev ? *a.alpha_dev : a.alpha_host) * MDGAT_LOG2E;
    const float norm = -logf((float)(N + M));
    const float lmu = norm * MDGAT_LOG2E;
    const float lmuN = (logf((float)M) + norm) * MDGAT_LOG2E;
    const float lnu = lmu;
    const float lnuM = (logf((float)N) + norm) * MDGAT_LOG2E;
    const int row0 = j * 128 + wave * CL_RPW;   // first row of this wave
    const int col0 = lane * 8;                // first column of this lane
    gu64* slots = (gu64*)(a.slots) + (size_t)group * 2 * G * SLOT_STRIDE;
    unsigned epoch = 0;
    bool failed = false;

    for (int pair = group; pair < a.B; pair += a.ngroups) {
        const float* S = a.scores + (size_t)pair * N * M;
        // ---- the CL_RPW x 8 block of this lane, base-2 log domain (clamped addresses + selects: no branches) ----
        float s[CL_RPW][8];
        const bool vec_ok = (M & 3) == 0 && col0 + 8 <= M;
#pragma unroll
        for (int r = 0; r < CL_RPW; ++r) {
            const int i = row0 + r;
            const float* row = S + (size_t)min(i, N - 1) * M;
            if (vec_ok) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(row + col0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(row + col0 + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s[r][c] = i < N ? x0[c] * MDGAT_LOG2E : NEG_BIG;
                    s[r][4 + c] = i < N ? x1[c] * MDGAT_LOG2E : NEG_BIG;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = row[min(col0 + c, M - 1)];
                    s[r][c] = (i < N && col0 + c < M) ? x * MDGAT_LOG2E : NEG_BIG;
                }
            }
        }
        __syncthreads();                       // previous pair's readers of v are done
        for (int t = tid; t <= M; t += CL_THREADS) v[t] = 0.f;
        if (tid == 0) misc[0] = 0.f;
        float u[CL_RPW];
#pragma unroll
        for (int r = 0; r < CL_RPW; ++r) u[r] = 0.f;
        __syncthreads();

        for (int it = 0; it < a.iters; ++it) {
            ++epoch;
            // ---- row update (mdgat.py:283) ----
            float vr[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) vr[c] = (col0 + c < M) ? v[col0 + c] : 0.f;
            const float bM = alpha + v[M];
            if (wave == CL_WAVES - 1) {
                const float lse = alpha + wave_lse2(v, M, v[M], lane);
                if (lane == 0) misc[0] = lmuN - lse;
            }
            float mx[CL_RPW];
#pragma unroll
            for (int r = 0; r < CL_RPW; ++r) {
                float m = s[r][0] + vr[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) m = fmaxf(m, s[r][c] + vr[c]);
                mx[r] = m;
            }
#pragma unroll
            for (int r = 0; r < CL_RPW; ++r) mx[r] = fmaxf(wave_max_dpp(mx[r]), bM);
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(vr[c]));   // recompute s + v below: do not keep 128 sums live
            float ex[CL_RPW];
#pragma unroll
            for (int r = 0; r < CL_RPW; ++r) {
                float e = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) e += ex2(s[r][c] + vr[c] - mx[r]);
                ex[r] = e;
            }
#pragma unroll
            for (int r = 0; r < CL_RPW; ++r) {
                const float e = wave_sum_dpp(ex[r]) + ex2(bM - mx[r]);
                u[r] = lmu - (mx[r] + lg2(e));
            }
            // ---- column update, wave-local part (mdgat.py:284) ----
            float cmx[8], csm[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float m = s[0][c] + u[0];
#pragma unroll
                for (int r = 1; r < CL_RPW; ++r) m = fmaxf(m, s[r][c] + u[r]);
                asm volatile("" : "+v"(m));
                float e = 0.f;
#pragma unroll
                for (int r = 0; r < CL_RPW; ++r) e += ex2(s[r][c] + u[r] - m);
                cmx[c] = m; csm[c] = e;
            }
            {
                f32x4* pmw = reinterpret_cast<f32x4*>(pm + wave * 512 + col0);
                f32x4* psw = reinterpret_cast<f32x4*>(ps + wave * 512 + col0);
                pmw[0] = f32x4{cmx[0], cmx[1], cmx[2], cmx[3]}; pmw[1] = f32x4{cmx[4], cmx[5], cmx[6], cmx[7]};
                psw[0] = f32x4{csm[0], csm[1], csm[2], csm[3]}; psw[1] = f32x4{csm[4], csm[5], csm[6], csm[7]};
            }
            {   // LSE of this wave's valid row potentials (for the dustbin column)
                float m = NEG_BIG;
#pragma unroll
                for (int r = 0; r < CL_RPW; ++r) if (row0 + r < N) m = fmaxf(m, u[r]);
                float e = 0.f;
#pragma unroll
                for (int r = 0; r < CL_RPW; ++r) if (row0 + r < N) e += ex2(u[r] - m);
                if (lane == 0) pu[wave] = (e > 0.f) ? m + lg2(e) : NEG_BIG;
            }
            __syncthreads();
            // ---- merge the 16 waves, exchange with the partner workgroups, new column potentials ----
            for (int t = tid; t <= M; t += CL_THREADS) {
                float loc;
                if (t < M) {
                    float m = pm[t];
#pragma unroll
                    for (int w = 1; w < CL_WAVES; ++w) m = fmaxf(m, pm[w * 512 + t]);
                    float e = 0.f;
#pragma unroll
                    for (int w = 0; w < CL_WAVES; ++w) e += ps[w * 512 + t] * ex2(pm[w * 512 + t] - m);
                    loc = m + lg2(e);
                } else {
                    float m = pu[0];
#pragma unroll
                    for (int w = 1; w < CL_WAVES; ++w) m = fmaxf(m, pu[w]);
                    float e = 0.f;
#pragma unroll
                    for (int w = 0; w < CL_WAVES; ++w) e += ex2(pu[w] - m);
                    loc = m + lg2(e);
                }
                float vals[G];
                vals[0] = loc;
                if (G > 1) {
                    gu64* mine = slots + ((size_t)(epoch & 1) * G + j) * SLOT_STRIDE + t;
                    __hip_atomic_store(mine, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, loc),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int p = 0; p < G; ++p) {
                        if (p == j) { vals[p] = loc; continue; }
                        gu64* theirs = slots + ((size_t)(epoch & 1) * G + p) * SLOT_STRIDE + t;
                        unsigned long long x = 0;
                        unsigned spins = 0;
                        while (true) {
                            x = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(x >> 32) == epoch || failed) break;
                            if (++spins > (1u << 22)) { failed = true; atomicOr(a.error_word, 1u); break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        vals[p] = __builtin_bit_cast(float, (unsigned)x);
                    }
                }
                const float uN = misc[0];
                const float extra = (t < M) ? alpha + uN : uN;     // dustbin-row term
                float m = extra;
#pragma unroll
                for (int p = 0; p < G; ++p) m = fmaxf(m, vals[p]);
                float e = ex2(extra - m);
#pragma unroll
                for (int p = 0; p < G; ++p) e += ex2(vals[p] - m);
                const float lse = m + lg2(e);
                v[t] = (t < M) ? lnu - lse : lnuM - (alpha + lse);
            }
            __syncthreads();
        }

        // ---- Z = couplings + u + v - norm (mdgat.py:285, 307), natural-log units ----
        float* Zp = a.Z + (size_t)pair * (N + 1) * (M + 1);
        const float poison = (G > 1 && __hip_atomic_load(a.error_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                 ? __builtin_nanf("") : 0.f;   // a partner never arrived: make the failure loud
        const float vM = v[M] + poison;
        float vr[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) vr[c] = (col0 + c < M) ? v[col0 + c] : 0.f;
#pragma unroll
        for (int r = 0; r < CL_RPW; ++r) {
            const int i = row0 + r;
            if (i < N) {
                float* zr = Zp + (size_t)i * (M + 1);
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (col0 + c < M) zr[col0 + c] = (s[r][c] + u[r] + vr[c]) * MDGAT_LN2 - norm;
                if (lane == 0) zr[M] = (alpha + u[r] + vM) * MDGAT_LN2 - norm;
            }
        }
        if (j == G - 1) {
            // dustbin row: u_N from the final column potentials is NOT recomputed by the reference (it is the
            // value of the last row update), which is what misc[0] still holds
            const float uN = misc[0];
            float* zl = Zp + (size_t)N * (M + 1);
            for (int t = tid; t <= M; t += CL_THREADS) zl[t] = (alpha + uN + v[t]) * MDGAT_LN2 - norm;
        }
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
    int* valid_count;   // global count of valid frame-0 rows (dustbin modes; mdgat.py:465 quirk)
};

__global__ __launch_bounds__(1024) void extract_kernel(ExArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, M = a.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Z = a.Z + (size_t)blockIdx.x * (N + 1) * (M + 1);
    int* idx0 = reinterpret_cast<int*>(smem);   // [N]
    int* idx1 = idx0 + N;                       // [M]
    float* val0 = reinterpret_cast<float*>(idx1 + M);   // [N]
    float* val1 = val0 + N;                     // [M]
    int* nvalid = reinterpret_cast<int*>(val1 + M);
    const bool inner = a.mode >= MDGAT_EXTRACT_THRESHOLD;   // arg-max over the inner N x M block only
    const int ncol = inner ? M : M + 1;   // columns scanned per row
    const int nrow = inner ? N : N + 1;   // rows scanned per column
    if (tid == 0) *nvalid = 0;

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
    __syncthreads();

    int64_t* m0 = a.m0 + (size_t)blockIdx.x * N;
    int64_t* m1 = a.m1 + (size_t)blockIdx.x * M;
    float* s0 = a.s0 + (size_t)blockIdx.x * N;
    float* s1 = a.s1 + (size_t)blockIdx.x * M;

    if (a.mode == MDGAT_EXTRACT_DUSTBIN || a.mode == MDGAT_EXTRACT_DUSTBIN_MUTUAL) {
        const bool mutual = a.mode == MDGAT_EXTRACT_DUSTBIN_MUTUAL;
        int local = 0;
        for (int i = tid; i < N; i += 1024) {
            const int j = idx0[i];
            const bool valid = j < M;
            const bool keep = valid && (!mutual || idx1[j] == i);
            m0[i] = valid ? j : -1;
            s0[i] = keep ? expf(val0[i]) : 0.f;
            local += valid ? 1 : 0;
        }
        for (int j = tid; j < M; j += 1024) {
            const int i = idx1[j];
            const bool valid = i < N;
            const bool keep = valid && (!mutual || idx0[i] == j);
            m1[j] = valid ? i : -1;
            s1[j] = keep ? expf(val1[j]) : 0.f;
        }
        if (local) atomicAdd(nvalid, local);
        __syncthreads();
        if (tid == 0 && *nvalid) atomicAdd(a.valid_count, *nvalid);
    } else if (a.mode == MDGAT_EXTRACT_THRESHOLD) {
        for (int i = tid; i < N; i += 1024) {
            const float e = expf(val0[i]);
            const bool valid = e > a.thr;
            m0[i] = valid ? idx0[i] : -1;
            s0[i] = valid ? e : 0.f;
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
}

// mdgat.py:465-467: when NO frame-0 keypoint of the whole batch is matched, both score vectors are zeros
__global__ void extract_alldust_fixup(const int* valid_count, float* s1, size_t n) {
    if (*valid_count != 0) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s1[i] = 0.f;
}

int* g_valid_count[16] = {nullptr};

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

size_t sinkhorn_cluster_workspace_bytes(int B, int N, int M) {
    if (N > 512 || M > 512) return 0;
    return 256 + (size_t)64 * 2 * 4 * SLOT_STRIDE * sizeof(unsigned long long);
}

template <int G>
static int launch_cluster(int B, int N, int M, const float* scores, const float* alpha_dev, float alpha_host, int iters,
                          float* Z, void* ws, int num_cu, hipStream_t s) {
    int ngroups = num_cu / G;
    if (ngroups > 64) ngroups = 64;
    if (ngroups > B) ngroups = B;
    if (ngroups >= 8) ngroups &= ~7;
    if (ngroups < 1) return MDGAT_ERR_UNSUPPORTED;
    const size_t ws_bytes = 256 + (size_t)ngroups * 2 * G * SLOT_STRIDE * sizeof(unsigned long long);
    if (int rc = mdgat_check_hip(hipMemsetAsync(ws, 0, ws_bytes, s), "memset(sinkhorn slots)")) return rc;
    SkcArgs a{scores, alpha_dev, alpha_host, Z, reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + 256),
              static_cast<unsigned*>(ws), B, N, M, iters, ngroups};
    const size_t lds = CLUSTER_LDS_FLOATS * sizeof(float);
    static bool attr_set[3] = {false, false, false};
    const int gi = G == 1 ? 0 : (G == 2 ? 1 : 2);
    if (!attr_set[gi]) {
        if (int rc = mdgat_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_cluster_kernel<G>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                                     "sinkhorn cluster LDS attribute"))
            return rc;
        attr_set[gi] = true;
    }
    void* args[] = {&a};
    // cooperative launch: the runtime checks that all ngroups * G workgroups can be co-resident
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(sinkhorn_cluster_kernel<G>), dim3(ngroups * G),
                                              dim3(CL_THREADS), args, (unsigned)lds, s);
    return mdgat_check_hip(e, "sinkhorn cluster launch");
}

size_t mdgat_sinkhorn_ws_bytes_impl(int B, int N, int M) { return sinkhorn_cluster_workspace_bytes(B, N, M); }

int launch_sinkhorn(int B, int N, int M, const float* scores, const float* bin_score_dev, float bin_score_host,
                    int iters, float* Z, void* ws, size_t ws_bytes, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    if (N <= 0 || M <= 0 || iters < 0) { mdgat_set_error("sinkhorn: bad shape N=%d M=%d iters=%d", N, M, iters); return MDGAT_ERR_BAD_ARG; }
    const size_t need = sinkhorn_cluster_workspace_bytes(B, N, M);
    if (need && ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 255) == 0) {
        int dev = 0, num_cu = 0;
        if (int rc = mdgat_check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
        if (int rc = mdgat_check_hip(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev), "CU count")) return rc;
        if (N <= 128) return launch_cluster<1>(B, N, M, scores, bin_score_dev, bin_score_host, iters, Z, ws, num_cu, s);
        if (N <= 256) return launch_cluster<2>(B, N, M, scores, bin_score_dev, bin_score_host, iters, Z, ws, num_cu, s);
        return launch_cluster<4>(B, N, M, scores, bin_score_dev, bin_score_host, iters, Z, ws, num_cu, s);
    }
    // streaming kernel: any shape up to M = 2048, no workspace
    SkArgs a{scores, bin_score_dev, bin_score_host, Z, N, M, iters};
    if (M <= 64) return launch_sk<1, 16>(a, B, s);
    if (M <= 128) return launch_sk<2, 16>(a, B, s);
    if (M <= 256) return launch_sk<4, 16>(a, B, s);
    if (M <= 512) return launch_sk<8, 16>(a, B, s);
    if (M <= 1024 && N <= 4096) return launch_sk<16, 8>(a, B, s);
    if (M <= 2048 && N <= 4096) return launch_sk<32, 8>(a, B, s);
    mdgat_set_error("sinkhorn: M=%d > 2048 unsupported", M);
    return MDGAT_ERR_UNSUPPORTED;
}

int launch_extract(int B, int N, int M, const float* Z, int mode, float thr, int64_t* m0, int64_t* m1, float* s0,
                   float* s1, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    if (mode < 0 || mode > 3) { mdgat_set_error("extract: bad mode %d", mode); return MDGAT_ERR_BAD_ARG; }
    int dev = 0;
    if (int rc = mdgat_check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (dev >= 16) { mdgat_set_error("extract: device index %d >= 16", dev); return MDGAT_ERR_UNSUPPORTED; }
    if (!g_valid_count[dev]) {
        if (int rc = mdgat_check_hip(hipMalloc(&g_valid_count[dev], sizeof(int)), "hipMalloc(valid_count)")) return rc;
    }
    if (int rc = mdgat_check_hip(hipMemsetAsync(g_valid_count[dev], 0, sizeof(int), s), "memset(valid_count)")) return rc;
    ExArgs a{Z, N, M, mode, thr, m0, m1, s0, s1, g_valid_count[dev]};
    const size_t lds = (size_t)(2 * (N + M) + 4) * sizeof(float);
    hipLaunchKernelGGL(extract_kernel, dim3(B), dim3(1024), lds, s, a);
    if (int rc = mdgat_check_hip(hipGetLastError(), "extract launch")) return rc;
    if (mode == MDGAT_EXTRACT_DUSTBIN || mode == MDGAT_EXTRACT_DUSTBIN_MUTUAL) {
        const size_t n = (size_t)B * M;
        hipLaunchKernelGGL(extract_alldust_fixup, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, s,
                           g_valid_count[dev], s1, n);
        return mdgat_check_hip(hipGetLastError(), "extract fixup launch");
    }
    return MDGAT_OK;
}
