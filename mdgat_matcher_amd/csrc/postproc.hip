// The steps on either side of the matcher that the reference does per pair on the host with numpy
// (SURVEY.md section 8f):
//
//   pose_kernel      rigid pose from the matches: solve_icp (utils/utils_test.py:73-110: centroids, 3x3 cross
//                    covariance, SVD, R = U V^T - no reflection fix, like the reference), inlier count and
//                    RTE / RRE against a ground-truth pose (calculate_error, utils_test.py:41-71)
//   gt_match_kernel  ground-truth matches of a frame pair: brute-force nearest neighbours of the world-frame
//                    keypoints in both directions under a distance threshold, optional mutual check
//                    (load_data.py:238-285)
//
// Both are tiny (3-D points, <= 2048 per frame) and latency bound: one workgroup per pair, fp64 arithmetic like the
// reference (MI355X runs fp64 at half the fp32 vector rate), so the results agree to round-off, not to 1e-4.
#include "common.hpp"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// workgroup-wide sum of NV doubles per thread -> every thread gets the totals (scratch: [16][NV])
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += scratch[w * NV + i];
        v[i] = s;
    }
}

// One-sided Jacobi SVD of a 3x3 matrix (fp64): A = U diag(s) V^T.  Returns R = U V^T, the orthogonal factor
// np.dot(U, Vh) of solve_icp (unique for a non-singular A, reflections included).
__device__ void polar_uvt(const double (&A)[3][3], double (&R)[3][3]) {
    double G[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { G[i][j] = A[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < 3; ++i) { alpha += G[i][p] * G[i][p]; beta += G[i][q] * G[i][q]; gamma += G[i][p] * G[i][q]; }
                off = fmax(off, fabs(gamma) / (sqrt(alpha * beta) + 1e-300));
                if (fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double gp = G[i][p], gq = G[i][q];
                    G[i][p] = c * gp - s * gq; G[i][q] = s * gp + c * gq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    // columns of G are s_j u_j; R = sum_j u_j v_j^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = 0.0;
    for (int k = 0; k < 3; ++k) {
        double nrm = 0.0;
        for (int i = 0; i < 3; ++i) nrm += G[i][k] * G[i][k];
        nrm = sqrt(nrm);
        const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;   // singular direction: contributes nothing (degenerate input)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] += G[i][k] * inv * V[j][k];
    }
}

struct PoseArgs {
    const float* kpts0;      // [B][N][3]
    const float* kpts1;      // [B][M][3]
    const int64_t* matches0; // [B][N], -1 = unmatched
    const double* T_gt;      // [B][4][4] or NULL
    double* T;               // [B][4][4]
    double* stats;           // [B][5]: matches, inliers, inlier ratio, translation error, rotation error (rad)
    int N, M;
    double inlier_dist;
};

__global__ __launch_bounds__(256) void pose_kernel(PoseArgs a) {
    __shared__ double scratch[4 * 16];
    __shared__ double Rt[12];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* k0 = a.kpts0 + (size_t)b * a.N * 3;
    const float* k1 = a.kpts1 + (size_t)b * a.M * 3;
    const int64_t* m0 = a.matches0 + (size_t)b * a.N;

    // ---- centroids of the matched points (utils_test.py:89-95): Q = frame-0 points, P = their frame-1 partners ----
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < a.N; i += 256) {
        const int64_t j = m0[i];
        if (j >= 0 && j < a.M) {
            acc[0] += 1.0;
            for (int c = 0; c < 3; ++c) { acc[1 + c] += (double)k0[i * 3 + c]; acc[4 + c] += (double)k1[j * 3 + c]; }
        }
    }
    {
        double v4[4] = {acc[0], acc[1], acc[2], acc[3]};
        block_sum<4>(v4, scratch);
        acc[0] = v4[0]; acc[1] = v4[1]; acc[2] = v4[2]; acc[3] = v4[3];
        double v3[4] = {acc[4], acc[5], acc[6], 0.0};
        block_sum<4>(v3, scratch);
        acc[4] = v3[0]; acc[5] = v3[1]; acc[6] = v3[2];
    }
    const double n = acc[0];
    const double inv_n = n > 0.0 ? 1.0 / n : 0.0;
    const double uq[3] = {acc[1] * inv_n, acc[2] * inv_n, acc[3] * inv_n};
    const double up[3] = {acc[4] * inv_n, acc[5] * inv_n, acc[6] * inv_n};

    // ---- H = Qc^T Pc (utils_test.py:97) ----
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < a.N; i += 256) {
        const int64_t j = m0[i];
        if (j >= 0 && j < a.M) {
            double q[3], p[3];
            for (int c = 0; c < 3; ++c) { q[c] = (double)k0[i * 3 + c] - uq[c]; p[c] = (double)k1[j * 3 + c] - up[c]; }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) h[r * 3 + c] += q[r] * p[c];
        }
    }
    {
        double v[4] = {h[0], h[1], h[2], h[3]};
        block_sum<4>(v, scratch); h[0] = v[0]; h[1] = v[1]; h[2] = v[2]; h[3] = v[3];
        double w[4] = {h[4], h[5], h[6], h[7]};
        block_sum<4>(w, scratch); h[4] = w[0]; h[5] = w[1]; h[6] = w[2]; h[7] = w[3];
        double z[4] = {h[8], 0, 0, 0};
        block_sum<4>(z, scratch); h[8] = z[0];
    }
    if (tid == 0) {
        double A[3][3], R[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = h[r * 3 + c];
        polar_uvt(A, R);                                           // R = U V^T (utils_test.py:97-98)
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Rt[r * 4 + c] = R[r][c];
            Rt[r * 4 + 3] = uq[r] - (R[r][0] * up[0] + R[r][1] * up[1] + R[r][2] * up[2]);   // t = uq - R up (99)
        }
        double* T = a.T + (size_t)b * 16;
        for (int i = 0; i < 12; ++i) T[i] = Rt[i];
        T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
    }
    __syncthreads();
    // ---- inliers: |T p1 - p0| < 1 m (utils_test.py:55-63) ----
    double cnt[4] = {0, 0, 0, 0};
    for (int i = tid; i < a.N; i += 256) {
        const int64_t j = m0[i];
        if (j >= 0 && j < a.M) {
            double d2 = 0.0;
            for (int r = 0; r < 3; ++r) {
                const double w = Rt[r * 4 + 0] * (double)k1[j * 3 + 0] + Rt[r * 4 + 1] * (double)k1[j * 3 + 1] +
                                 Rt[r * 4 + 2] * (double)k1[j * 3 + 2] + Rt[r * 4 + 3] - (double)k0[i * 3 + r];
                d2 += w * w;
            }
            if (sqrt(d2) < a.inlier_dist) cnt[0] += 1.0;
        }
    }
    block_sum<4>(cnt, scratch);
    if (tid == 0) {
        double* st = a.stats + (size_t)b * 5;
        st[0] = n; st[1] = cnt[0]; st[2] = n > 0.0 ? cnt[0] / n : 0.0;
        double rte = __builtin_nan(""), rre = __builtin_nan("");
        if (a.T_gt) {
            // T_error = inv(T) T_gt with inv(T) = [R^T | -R^T t] (utils_test.py:65-70; R is orthogonal)
            const double* G = a.T_gt + (size_t)b * 16;
            double E[3][4];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < 3; ++k) s += Rt[k * 4 + r] * (G[k * 4 + c] - (c == 3 ? Rt[k * 4 + 3] : 0.0));
                    E[r][c] = s;
                }
            rte = sqrt(E[0][3] * E[0][3] + E[1][3] * E[1][3] + E[2][3] * E[2][3]);
            rre = acos((E[0][0] + E[1][1] + E[2][2] - 1.0) * 0.5);   // unclamped, like the reference: may be NaN
        }
        st[3] = rte; st[4] = rre;
    }
}

struct GtArgs {
    const float* kpts0;   // [B][N][3] sensor-frame keypoints
    const float* kpts1;   // [B][M][3]
    const double* T0;     // [B][4][4] sensor -> world of frame 0 (pose . T_cam0_velo, load_data.py:238-242) or NULL
    const double* T1;     // [B][4][4]
    int64_t* gt0;         // [B][N]
    int64_t* gt1;         // [B][M]
    int64_t* rep;         // [B] repeatability count (load_data.py:264)
    int N, M, mutual;
    double threshold;
};

__device__ __forceinline__ void to_world(const double* T, const float* p, double (&w)[3]) {
    if (!T) { w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; return; }
#pragma unroll
    for (int r = 0; r < 3; ++r) w[r] = T[r * 4 + 0] * (double)p[0] + T[r * 4 + 1] * (double)p[1] + T[r * 4 + 2] * (double)p[2] + T[r * 4 + 3];
}

__global__ __launch_bounds__(256) void gt_match_kernel(GtArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int N = a.N, M = a.M, b = blockIdx.x, tid = threadIdx.x;
    double* w0 = sm;                 // [N][3] world-frame keypoints of frame 0
    double* w1 = w0 + 3 * N;         // [M][3]
    int* nn0 = reinterpret_cast<int*>(w1 + 3 * M);   // [N] argmin over frame 1 (min2 of load_data.py:259)
    int* nn1 = nn0 + N;                                 // [M] argmin over frame 0 (min1, 258)
    double* d0 = reinterpret_cast<double*>(nn1 + M + ((N + M) & 1));   // [N] min distance of a frame-0 point (min1v, 260)
    double* d1 = d0 + N;                                // [M] (min2v, 281)
    __shared__ int repc;
    if (tid == 0) repc = 0;
    const double* T0 = a.T0 ? a.T0 + (size_t)b * 16 : nullptr;
    const double* T1 = a.T1 ? a.T1 + (size_t)b * 16 : nullptr;
    for (int i = tid; i < N; i += 256) { double w[3]; to_world(T0, a.kpts0 + ((size_t)b * N + i) * 3, w); w0[3 * i] = w[0]; w0[3 * i + 1] = w[1]; w0[3 * i + 2] = w[2]; }
    for (int j = tid; j < M; j += 256) { double w[3]; to_world(T1, a.kpts1 + ((size_t)b * M + j) * 3, w); w1[3 * j] = w[0]; w1[3 * j + 1] = w[1]; w1[3 * j + 2] = w[2]; }
    __syncthreads();
    // cdist + argmin (first minimum, like numpy) in both directions
    for (int i = tid; i < N; i += 256) {
        double best = __builtin_inf(); int bj = 0;
        const double x = w0[3 * i], y = w0[3 * i + 1], z = w0[3 * i + 2];
        for (int j = 0; j < M; ++j) {
            const double dx = x - w1[3 * j], dy = y - w1[3 * j + 1], dz = z - w1[3 * j + 2];
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            if (d < best) { best = d; bj = j; }
        }
        nn0[i] = bj; d0[i] = best;
    }
    for (int j = tid; j < M; j += 256) {
        double best = __builtin_inf(); int bi = 0;
        const double x = w1[3 * j], y = w1[3 * j + 1], z = w1[3 * j + 2];
        for (int i = 0; i < N; ++i) {
            const double dx = w0[3 * i] - x, dy = w0[3 * i + 1] - y, dz = w0[3 * i + 2] - z;
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            if (d < best) { best = d; bi = i; }
        }
        nn1[j] = bi; d1[j] = best;
    }
    __syncthreads();
    int64_t* g0 = a.gt0 + (size_t)b * N;
    int64_t* g1 = a.gt1 + (size_t)b * M;
    int local = 0;
    if (!a.mutual) {
        // match1[min1v < thr] = min2[min1v < thr]; match2[min2v < thr] = min1[min2v < thr] (load_data.py:278-283)
        for (int i = tid; i < N; i += 256) { const bool ok = d0[i] < a.threshold; g0[i] = ok ? nn0[i] : -1; local += ok; }
        for (int j = tid; j < M; j += 256) g1[j] = d1[j] < a.threshold ? nn1[j] : -1;
    } else {
        // load_data.py:272-276: matches = {j : j = min2[i] for some i with min1v[i] < thr}  intersected with
        // {j : min2[min1[j]] == j}; match1[min1[j]] = j, match2[j] = min1[j] for those j
        for (int i = tid; i < N; i += 256) { g0[i] = -1; local += d0[i] < a.threshold; }
        for (int j = tid; j < M; j += 256) g1[j] = -1;
        __syncthreads();
        for (int j = tid; j < M; j += 256) {
            const int i = nn1[j];
            if (nn0[i] == j && d0[i] < a.threshold) { g1[j] = i; g0[i] = j; }
        }
    }
    if (local) atomicAdd(&repc, local);
    __syncthreads();
    if (tid == 0) a.rep[b] = repc;
}

}  // namespace

int launch_pose(int B, int N, int M, const float* kpts0, const float* kpts1, const int64_t* matches0, const double* T_gt,
                double inlier_dist, double* T, double* stats, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    PoseArgs a{kpts0, kpts1, matches0, T_gt, T, stats, N, M, inlier_dist};
    hipLaunchKernelGGL(pose_kernel, dim3(B), dim3(256), 0, s, a);
    return mdgat_check_hip(hipGetLastError(), "pose launch");
}

int launch_gt_match(int B, int N, int M, const float* kpts0, const float* kpts1, const double* T0, const double* T1,
                    double threshold, int mutual, int64_t* gt0, int64_t* gt1, int64_t* rep, hipStream_t s) {
    if (B <= 0) return MDGAT_OK;
    const size_t lds = (size_t)(3 * (N + M) + (N + M)) * sizeof(double) + (size_t)(N + M + 2) * sizeof(int);
    if (lds > 160 * 1024) { mdgat_set_error("gt_match: %d + %d keypoints exceed the LDS budget", N, M); return MDGAT_ERR_UNSUPPORTED; }
    GtArgs a{kpts0, kpts1, T0, T1, gt0, gt1, rep, N, M, mutual, threshold};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gt_match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gt_match_kernel, dim3(B), dim3(256), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "gt_match launch");
}
