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

int launch_sinkhorn(int B, int N, int M, const float* scores, const float* bin_score_dev, float bin_score_host,
                    int iters, float* Z, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    if (N <= 0 || M <= 0 || iters < 0) { mdgat_set_error("sinkhorn: bad shape N=%d M=%d iters=%d", N, M, iters); return MDGAT_ERR_BAD_ARG; }
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
