// The kNN graph helper: knn() of mdgat.py:8-15 (+ get_graph_feature's adjacency, 17-32).
//
//   idx[b][n][:] = topk_k( -|x_n|^2 + 2 x_n . s_m - |s_m|^2 )  over the M source points, nearest first.
//
// Two stages, both bound by memory traffic rather than arithmetic (the north star's "HBM-bound kNN"):
//   1. the N x M matrix of 2 x.s - |s|^2 (the row-constant |x|^2 does not change a row's order) - for C = 128 feature
//      channels on the matrix cores, the score-matrix kernel of scores.hip with split-f16 operands (fp32-class
//      products: a plain f16 product would reorder neighbours), 4 N M bytes written once;
//      for other channel counts (C = 3: keypoint coordinates) nothing is stored: the selection kernel computes
//      -sum (x - s)^2 on the fly - in fp32 the difference form has no cancellation, the expanded form would lose
//      digits at |x| ~ 20 m and reorder near neighbours;
//   2. selection, one wave per query row: the row (a segment of <= 4096 values) sits in LDS, the exact k-th largest
//      value is found by bisection on the monotone integer image of a float (count(v >= t) = ballot + s_bcnt1 per
//      64 values, <= 32 probes, any distribution, any ties), the k survivors are compacted with ballot prefix sums
//      (exact ties at the k-th value: lowest indices first) and put into torch.topk's order (value descending) by a
//      bitonic sort in LDS.  Rows longer than a segment keep a running best-k list that is merged segment by segment,
//      so M is not limited; k <= 1024.
#include "common.hpp"

namespace {

constexpr int KNN_SEG = 4096;       // row values held in LDS at a time (per wave)
constexpr int KNN_WAVES = 4;        // query rows per workgroup
constexpr int KNN_MAXK = 1024;

__device__ __forceinline__ int f2ord(float f) { const int b = __builtin_bit_cast(int, f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float ord2f(int o) { return __builtin_bit_cast(float, o ^ ((o >> 31) & 0x7fffffff)); }
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ p, float* __restrict__ out, size_t rows, int C) {
    for (size_t r = blockIdx.x * (size_t)256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const float* q = p + r * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(q[c], q[c], s);
        out[r] = s;
    }
}

struct KnnArgs {
    const float* x;      // [B][N][C]
    const float* src;    // [B][M][C]
    const float* dist;   // [B][N][M] precomputed 2 x.s - |s|^2, or NULL: computed here as -sum (x - s)^2
    int64_t* idx;        // [B][N][k]
    int64_t* adj;        // [B][N][M] or NULL
    int C, N, M, k, KP;  // KP = k rounded up to a power of two
};

// descending by value, ascending by index among equal values; n = power of two; one wave, arrays in LDS
__device__ __forceinline__ void bitonic_sort_desc(float* v, int* id, int n, int lane) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                const int pos = ((t / stride) * stride << 1) + (t % stride);
                const int par = pos + stride;
                const bool desc = (pos & size) == 0;
                const float a = v[pos], b = v[par];
                const int ia = id[pos], ib = id[par];
                const bool a_first = a > b || (a == b && ia < ib);     // a belongs before b in the final order
                if (a_first != desc) { v[pos] = b; v[par] = a; id[pos] = ib; id[par] = ia; }
            }
            lds_order();
        }
    }
}

__global__ __launch_bounds__(64 * KNN_WAVES) void knn_select_kernel(KnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n = blockIdx.x * KNN_WAVES + wave;
    if (n >= a.N) return;                      // (no workgroup barrier below: every wave is on its own)
    const int M = a.M, k = a.k, KP = a.KP;
    const int seg_cap = M < KNN_SEG ? ((M + 63) & ~63) : KNN_SEG;
    float* vals = smem + (size_t)wave * (seg_cap + 4 * KP);       // [seg_cap]
    float* cv = vals + seg_cap;                                   // [2 KP] candidate values
    int* ci = reinterpret_cast<int*>(cv + 2 * KP);                // [2 KP] candidate indices
    const float NEG_INF = -__builtin_inff();
    const float* xp = a.x + ((size_t)b * a.N + n) * a.C;
    int ncand = 0;

    for (int seg0 = 0; seg0 < M; seg0 += KNN_SEG) {
        const int len = min(KNN_SEG, M - seg0);
        const int len64 = (len + 63) & ~63;
        // ---- the segment's values (pads: -inf) ----
        if (a.dist) {
            const float* row = a.dist + ((size_t)b * a.N + n) * M + seg0;
            for (int i = lane; i < len64; i += 64) vals[i] = i < len ? row[i] : NEG_INF;
        } else {
            for (int i = lane; i < len64; i += 64) {
                float d = NEG_INF;
                if (i < len) {
                    const float* sp = a.src + ((size_t)b * M + seg0 + i) * a.C;
                    float acc = 0.f;
                    for (int c = 0; c < a.C; ++c) { const float t = xp[c] - sp[c]; acc = fmaf(t, t, acc); }
                    d = -acc;
                }
                vals[i] = d;
            }
        }
        lds_order();
        const int kk = min(k, len);
        // ---- exact kk-th largest value: bisection on the ordinal; invariant count(v >= lo) >= kk > count(v >= hi) ----
        auto count_ge = [&](float t) {
            int c = 0;
            for (int i = lane; i < len64; i += 64) c += __builtin_popcountll(__ballot(vals[i] >= t));
            return c;                          // wave-uniform
        };
        float vmin = __builtin_inff(), vmax = NEG_INF;
        for (int i = lane; i < len; i += 64) { const float v = vals[i]; vmin = fminf(vmin, v); vmax = fmaxf(vmax, v); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64)); }
        float thr;
        if (count_ge(vmax) >= kk) thr = vmax;
        else {
            int lo = f2ord(vmin), hi = f2ord(vmax);
            thr = vmin;
            bool exact = false;
            while ((unsigned)hi - (unsigned)lo > 1u) {     // (lo < hi always, both finite; the distance as unsigned)
                const int mid = lo + (int)(((unsigned)hi - (unsigned)lo) >> 1);
                const float t = ord2f(mid);
                const int c = count_ge(t);
                if (c == kk) { thr = t; exact = true; break; }
                if (c > kk) lo = mid; else hi = mid;
            }
            if (!exact) thr = ord2f(lo);       // ties at the kk-th value: count(v >= thr) > kk > count(v > thr)
        }
        // ---- compact the survivors behind the candidates kept so far: all v > thr ... then v == thr, lowest index first ----
        int base = ncand;
        for (int i0 = 0; i0 < len64; i0 += 64) {
            const float v = vals[i0 + lane];
            const bool s = v > thr;
            const unsigned long long m = __ballot(s);
            if (s) {
                const int p = base + __builtin_popcountll(m & ((1ull << lane) - 1));
                cv[p] = v; ci[p] = seg0 + i0 + lane;
            }
            base += __builtin_popcountll(m);
        }
        int need = ncand + kk - base;          // tied values still to take
        for (int i0 = 0; i0 < len64 && need > 0; i0 += 64) {
            const float v = vals[i0 + lane];
            const bool s = v == thr;
            const unsigned long long m = __ballot(s);
            const int r = __builtin_popcountll(m & ((1ull << lane) - 1));
            if (s && r < need) { cv[base + r] = v; ci[base + r] = seg0 + i0 + lane; }
            const int took = min(need, (int)__builtin_popcountll(m));
            base += took; need -= took;
        }
        ncand += kk;
        lds_order();
        // ---- more than one segment: keep the best k of (kept so far + this segment's) ----
        if (seg0 + KNN_SEG < M || seg0 > 0) {
            int p2 = 1;
            while (p2 < ncand) p2 <<= 1;
            for (int i = ncand + lane; i < p2; i += 64) { cv[i] = NEG_INF; ci[i] = 0x7fffffff; }
            lds_order();
            bitonic_sort_desc(cv, ci, p2, lane);
            ncand = min(ncand, k);
        }
    }
    if (M <= KNN_SEG) {                       // single segment: not sorted yet
        for (int i = ncand + lane; i < KP; i += 64) { cv[i] = NEG_INF; ci[i] = 0x7fffffff; }
        lds_order();
        bitonic_sort_desc(cv, ci, KP, lane);
    }
    int64_t* out = a.idx + ((size_t)b * a.N + n) * k;
    for (int i = lane; i < k; i += 64) out[i] = ci[i];
    if (a.adj) {
        int64_t* row = a.adj + ((size_t)b * a.N + n) * M;
        for (int i = lane; i < M; i += 64) row[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    // the ones below land after the zeros
        for (int i = lane; i < k; i += 64) row[ci[i]] = 1;
    }
}

}  // namespace

size_t mdgat_knn_ws_bytes_impl(int B, int C, int N, int M) {
    if (C != 128 || B <= 0 || N <= 0 || M <= 0) return 0;       // other channel counts: distances on the fly, no workspace
    return ((size_t)B * N * M + (size_t)B * M) * sizeof(float);
}

int launch_knn(int B, int C, int N, int M, int k, const float* x, const float* src, int64_t* idx, int64_t* adj,
               void* ws, size_t ws_bytes, hipStream_t s) {
    if (B <= 0 || N <= 0) return MDGAT_OK;
    if (C <= 0 || M <= 0) { mdgat_set_error("knn: bad shape C=%d M=%d", C, M); return MDGAT_ERR_BAD_ARG; }
    if (k <= 0 || k > M) { mdgat_set_error("knn: k=%d out of range for %d source points", k, M); return MDGAT_ERR_BAD_ARG; }
    if (k > KNN_MAXK) { mdgat_set_error("knn: k=%d > %d unsupported", k, KNN_MAXK); return MDGAT_ERR_UNSUPPORTED; }
    int KP = 64;
    while (KP < k) KP <<= 1;
    KnnArgs a{x, src, nullptr, idx, adj, C, N, M, k, KP};
    const size_t need = mdgat_knn_ws_bytes_impl(B, C, N, M);
    if (need && ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
        // C = 128: 2 x.s - |s|^2 on the matrix cores
        float* dist = static_cast<float*>(ws);
        float* ss = dist + (size_t)B * N * M;
        const size_t rows = (size_t)B * M;
        hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((rows + 255) / 256 < 2048 ? (rows + 255) / 256 : 2048)), dim3(256), 0, s, src, ss, rows, C);
        if (int rc = mdgat_check_hip(hipGetLastError(), "knn norms launch")) return rc;
        if (int rc = launch_dots(B, N, M, x, (size_t)N * 128, src, (size_t)M * 128, dist, 2.0f, ss, s)) return rc;
        a.dist = dist;
    }
    const int seg_cap = M < KNN_SEG ? ((M + 63) & ~63) : KNN_SEG;
    const size_t lds = (size_t)KNN_WAVES * (seg_cap + 4 * KP) * sizeof(float);
    static std::atomic<unsigned long long> optin;
    if (lds > 64 * 1024)
        if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(knn_select_kernel), 160 * 1024 - 256, optin, "knn LDS attribute")) return rc;
    hipLaunchKernelGGL(knn_select_kernel, dim3((N + KNN_WAVES - 1) / KNN_WAVES, B), dim3(64 * KNN_WAVES), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "knn launch");
}
