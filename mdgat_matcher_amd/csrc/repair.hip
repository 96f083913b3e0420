// Exact re-decision of near-threshold rows of a dynamic layer (mdgat.py:196-210, `logits.topk(k)`).
//
// dynamic_attention keeps the k largest logits of a row: a discontinuous function of the logits.  The attention kernels
// compute fp32-class logits (split-f16 products of q / k planes that were themselves projected in fp32-class arithmetic);
// where the k-th and the (k+1)-th largest logit of a row are closer than that arithmetic resolves, they may keep the other
// key than exact arithmetic on the same layer input does, and the row's message then moves by ~p_k |v_a - v_b|
// (profiles/parity_r3.txt: 19 such rows in 262 144 at BASELINE configs[1]).
//
// The attention kernels list every row whose (k+1)-th largest logit lies within mdgat_near_eps() below its threshold
// (attention.hip: topk_threshold, near_append; one more counting pass per tile).  This kernel, one workgroup per listed row:
//   1. recomputes the row's logits from the q / k planes (fp32 accumulation of the exact plane products: what the matrix
//      cores compute up to their accumulation order);
//   2. takes the candidates inside the window |s - thr| < W = near_eps + 2e-5: everything above the window is kept, everything
//      below dropped, whatever the arithmetic (the logits of the attention kernel and of step 1 differ by < 1e-5);
//   3. if the candidates are not all kept or all dropped: re-projects q of the row and k of every candidate from the layer's
//      fp32 input descriptors with the fp64 weights (fp32 head + fp32 residual of each weight: 48 bits) in fp64, takes the
//      exact logits' order among the candidates (equal logits - duplicated keypoints: lowest key index first, the kernels' tie
//      rule), and REWRITES the row's message with that selection (softmax over the kept keys, P.V from the planes).
// What remains different from the reference after this are rows whose order flips with the layer INPUT (error accumulated
// by the layers before: fp32-class descriptors against the reference's fp64 ones) - nothing local can see those
// (profiles/parity_r4.txt).
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int RP_THREADS = 256;
constexpr int RP_MAXK = 2048;        // keys of a frame (the dynamic kernels' limit)
constexpr int RP_MAXC = 32;          // candidates per row (more: the row is left alone - masses of equal logits)

struct RepairArgs {
    const _Float16 *q16, *k16, *vt16;
    float* msg;
    const float* x;
    const float *w, *wlo, *b, *blo; // qkv_w [384][128] (fp32 heads of the fp64 weights), residuals of its q | k rows [256][128]; qkv_b [384], residuals [256]
    int N, M, Npad, PP, cross, topk;
    const int* count; const RepairRec* recs; int cap;
    uint32_t* sel; int selW;
    int* stats;
    int stop;    // (measurement) leave a row after step `stop`
};

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The kernel is latency-bound (a row is a chain of dependent round trips to L2 / HBM, one workgroup per row): every step
// issues all the loads of a batch before it touches the first one.
struct ProjOperands { f32x4 wh[4], wl[4], xv[4]; };
// dims 0..31 of W x + b for one point, the whole workgroup: thread (d = tid >> 3, c8 = tid & 7) takes channels
// {4 (8 j + c8) .. + 3 : j < 4} of output row `row0 + d` - eight lanes read 128 contiguous bytes of a row
__device__ __forceinline__ void project_load(const RepairArgs& a, int row0, const float* xrow, int tid, ProjOperands& o) {
    const int d = tid >> 3, c8 = tid & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = 4 * (8 * j + c8);
        o.wh[j] = *reinterpret_cast<const f32x4*>(a.w + (size_t)(row0 + d) * 128 + ch);
        o.wl[j] = *reinterpret_cast<const f32x4*>(a.wlo + (size_t)(row0 + d) * 128 + ch);
        o.xv[j] = *reinterpret_cast<const f32x4*>(xrow + ch);
    }
}
__device__ __forceinline__ void project_finish(const RepairArgs& a, int row0, int tid, const ProjOperands& o, double* out) {
    const int d = tid >> 3;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc0 = fma((double)o.wh[j][0] + (double)o.wl[j][0], (double)o.xv[j][0], acc0);
        acc1 = fma((double)o.wh[j][1] + (double)o.wl[j][1], (double)o.xv[j][1], acc1);
        acc0 = fma((double)o.wh[j][2] + (double)o.wl[j][2], (double)o.xv[j][2], acc0);
        acc1 = fma((double)o.wh[j][3] + (double)o.wl[j][3], (double)o.xv[j][3], acc1);
    }
    double acc = acc0 + acc1;
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if ((tid & 7) == 0) out[d] = acc + ((double)a.b[row0 + d] + (double)a.blo[row0 + d]);
}

__global__ __launch_bounds__(RP_THREADS, 4) void topk_repair_kernel(RepairArgs a) {
    __shared__ __attribute__((aligned(16))) float sl[RP_MAXK + 8];           // logits, then probabilities
    __shared__ double proj[RP_MAXC + 1][32];    // q of the row, k of every candidate (fp64)
    __shared__ double l64[RP_MAXC];
    __shared__ int cl[RP_MAXC];
    __shared__ float red[8][33];
    __shared__ int cnt[2];                      // candidates, logits above the window
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n = *a.count;
    if (n > a.cap) n = a.cap;
    const int P = a.N + a.M;
    const float NEG = -__builtin_inff();
    for (int rec = blockIdx.x; rec < n; rec += gridDim.x) {
        const RepairRec r = a.recs[rec];
        const int head = r.bsh & 3, side = (r.bsh >> 2) & 1, b = r.bsh >> 3;
        const int q_off = side ? a.N : 0;
        const int src = a.cross ? (1 - side) : side;
        const int nk = src ? a.M : a.N;
        const int k_off = src ? a.N : 0;
        const int nk8 = (nk + 7) & ~7;
        if (tid < 2) cnt[tid] = 0;

        // ---- 1. the row's logits from the planes (base-2 units: the q planes carry log2(e) / sqrt(32)) ----
        // eight lanes share a key: lane (key = lane >> 3, c = lane & 7) holds 16-byte chunk c of the key's row (hi dims 8 c ..,
        // c >= 4: the residual plane of dims 8 (c - 4) ..) - a load instruction reads eight whole rows
        float qf[8];
        {
            const _Float16* qp = a.q16 + (((size_t)b * P + q_off + r.q) * 4 + head) * 64 + 8 * (lane & 3);
            const f16x8 h = *reinterpret_cast<const f16x8*>(qp), l = *reinterpret_cast<const f16x8*>(qp + 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[j] = (float)h[j] + (float)l[j];      // (exact: 22 bits)
        }
        float mx = NEG;
        const int ngroups = nk8 / 8;                    // groups of 8 keys; wave w takes groups w, w + 4, ... eight per batch
        for (int g0 = wave; g0 < ngroups; g0 += 32) {
            f16x8 kc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int key = (g0 + 4 * u) * 8 + (lane >> 3);
#pragma unroll
                for (int j = 0; j < 8; ++j) kc[u][j] = (_Float16)0.f;
                if (g0 + 4 * u < ngroups && key < nk)
                    kc[u] = *reinterpret_cast<const f16x8*>(a.k16 + (((size_t)b * P + k_off + key) * 4 + head) * 64 + 8 * (lane & 7));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int key = (g0 + 4 * u) * 8 + (lane >> 3);
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(qf[j], (float)kc[u][j], s);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                if (g0 + 4 * u < ngroups) {
                    if (key >= nk) s = NEG;
                    if ((lane & 7) == 0) sl[key] = s;
                    mx = fmaxf(mx, s);
                }
            }
        }
        mx = wave_max_f(mx);
        if (lane == 0) red[wave][0] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
        if (a.stop == 1) { __syncthreads(); continue; }

        // ---- 2. the window around the threshold ----
        const float thr = r.thr;
        const float W = mdgat_near_eps(r.thr, r.m) + 2.0e-5f * fmaxf(1.0f, (fabsf(r.m) + fabsf(r.thr)) * 0.1f);
        {
            int above = 0;
            for (int j = tid; j < nk8; j += RP_THREADS) {      // (pads are -inf: neither above nor inside)
                const float s = sl[j];
                above += __popcll(__ballot(s >= thr + W));
                if (s < thr + W && s > thr - W) {
                    const int pos = atomicAdd(&cnt[0], 1);      // (list order varies from run to run; the outcome does not: step 3 ranks by (logit, key))
                    if (pos < RP_MAXC) cl[pos] = j;
                }
            }
            if (lane == 0 && above) atomicAdd(&cnt[1], above);
        }
        __syncthreads();
        const int ncand = cnt[0];
        const int need = a.topk - cnt[1];           // candidates to keep
        if (a.stats && tid == 0) {
            atomicAdd(a.stats + 0, 1);
            if (ncand > RP_MAXC) atomicAdd(a.stats + 3, 1);
        }
        // all dropped / all kept, whatever their order: the attention kernel's row stands (uniform over the workgroup)
        if (ncand > RP_MAXC || need <= 0 || need >= ncand || a.stop == 2) { __syncthreads(); continue; }

        // ---- 3. exact logits of the candidates: q and k re-projected in fp64 from the layer's input descriptors ----
        {
            const float* qrow = a.x + ((size_t)b * P + q_off + r.q) * 128;
            ProjOperands o;
#pragma unroll 1
            for (int item = 0; item <= ncand; ++item) {     // (one round trip per item; four workgroups per CU overlap theirs)
                const int row0 = item == 0 ? head * 32 : 128 + head * 32;
                project_load(a, row0, item == 0 ? qrow : a.x + ((size_t)b * P + k_off + cl[item - 1]) * 128, tid, o);
                project_finish(a, row0, tid, o, proj[item]);
            }
        }
        __syncthreads();
        if (tid < ncand) {
            double d = 0.0;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) d = fma(proj[0][c], proj[1 + tid][c], d);
            l64[tid] = d;
        }
        __syncthreads();
        bool drop = false;
        if (tid < ncand) {
            // rank among the candidates (descending; equal logits: lowest key index first)
            const double mine = l64[tid];
            const int key = cl[tid];
            const float mine32 = sl[key];
            int rank = 0, rank32 = 0;
            for (int u = 0; u < ncand; ++u) {
                rank += (l64[u] > mine) || (l64[u] == mine && cl[u] < key);
                const float o32 = sl[cl[u]];
                rank32 += (o32 > mine32) || (o32 == mine32 && cl[u] < key);
            }
            drop = rank >= need;
            // (diagnostic) the exact order keeps another set than the order of the fp32-class logits of step 1
            if (a.stats && drop != (rank32 >= need)) atomicOr(&cnt[1], 1 << 30);
        }
        __syncthreads();
        if (drop) sl[cl[tid]] = NEG;                // dropped candidates leave the row
        if (a.stats && tid == 0 && (cnt[1] >> 30)) atomicAdd(a.stats + 2, 1);
        __syncthreads();

        if (a.stop == 3) { __syncthreads(); continue; }
        // ---- 4. the row again: softmax over the kept keys (everything above thr - W now), O = P V from the planes ----
        float lsum = 0.f;
        for (int j = tid; j < nk8; j += RP_THREADS) {
            const float s = sl[j];
            const float p = s > thr - W ? __builtin_amdgcn_exp2f(s - mx) : 0.f;
            sl[j] = p;
            lsum += p;
        }
        lsum = wave_sum_f(lsum);
        if (lane == 0) red[2 + wave][32] = lsum;
        __syncthreads();
        {
            // a wave takes 16 of the 64 V^T rows (plane, dim): a row's keys are contiguous, one load instruction reads 512 of them
            const _Float16* vbase = a.vt16 + (((size_t)b * 4 + head) * 64 + wave * 16) * a.PP + (src ? a.Npad : 0);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {          // eight rows per batch of loads
                float acc[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) acc[rr] = 0.f;
                for (int c = lane; c < nk8 / 8; c += 64) {
                    f16x8 v[8];
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) v[rr] = *reinterpret_cast<const f16x8*>(vbase + (size_t)(8 * half + rr) * a.PP + 8 * c);
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(sl + 8 * c), p1 = *reinterpret_cast<const f32x4*>(sl + 8 * c + 4);
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[rr] = fmaf(p0[j], (float)v[rr][j], fmaf(p1[j], (float)v[rr][4 + j], acc[rr]));
                }
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const float t = wave_sum_f(acc[rr]);
                    const int row = wave * 16 + 8 * half + rr;      // plane * 32 + dim
                    if (lane == 0) red[row >> 5][row & 31] = t;
                }
            }
        }
        __syncthreads();
        if (tid < 32) {
            const float l = (red[2][32] + red[3][32]) + (red[4][32] + red[5][32]);
            a.msg[((size_t)b * P + q_off + r.q) * 128 + head * 32 + tid] = (red[0][tid] + red[1][tid]) / l;
        }
        if (a.sel && wave == 0) {        // parity tap: the row's final selection
            uint32_t* row = a.sel + (((size_t)b * 4 + head) * P + q_off + r.q) * a.selW;
            for (int j0 = 0; j0 < a.selW * 32; j0 += 64) {
                const int j = j0 + lane;
                const unsigned long long bits = __ballot(j < nk && sl[j] > 0.f);
                if (lane == 0) {
                    row[j0 >> 5] = (uint32_t)bits;
                    if ((j0 >> 5) + 1 < a.selW) row[(j0 >> 5) + 1] = (uint32_t)(bits >> 32);
                }
            }
        }
        if (a.stats && tid == 0) atomicAdd(a.stats + 1, 1);
        __syncthreads();
    }
}

}  // namespace

int launch_topk_repair(const RepairLaunch& p, hipStream_t s) {
    if (p.B <= 0 || p.topk <= 0 || !p.near.count) return MDGAT_OK;
    const int nk_max = p.N > p.M ? p.N : p.M;
    if (nk_max > RP_MAXK) return MDGAT_OK;              // (the dynamic kernels reject such frames before this point)
    RepairArgs a{p.qkv.q16, p.qkv.k16, p.qkv.vt16, p.msg, p.x, p.w, p.wlo, p.b, p.blo, p.N, p.M, p.qkv.Npad, p.qkv.PP, p.cross, p.topk,
                 p.near.count, p.near.recs, p.near.cap, p.sel, (nk_max + 31) / 32, p.stats, getenv("MDGAT_REPAIR_STOP") ? atoi(getenv("MDGAT_REPAIR_STOP")) : 0};
    // one workgroup per listed row (~1 row in 10^3 is listed); a launch that finds the list empty leaves at once
    const long rows = (long)p.B * (p.N + p.M) * 4;
    int blocks = (int)(rows / 256);
    if (blocks < 8) blocks = 8;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(topk_repair_kernel, dim3(blocks), dim3(RP_THREADS), 0, s, a);
    return mdgat_check_hip(hipGetLastError(), "top-k repair launch");
}
