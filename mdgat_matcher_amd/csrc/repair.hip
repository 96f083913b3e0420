// Exact re-decision of near-threshold rows of a dynamic layer (mdgat.py:196-210, `logits.topk(k)`).
//
// dynamic_attention keeps the k largest logits of a row: a discontinuous function of the logits.  The attention kernels
// compute fp32-class logits (split-f16 products of q / k planes that were themselves projected in fp32-class arithmetic);
// where the k-th and the (k+1)-th largest logit of a row are closer than that arithmetic resolves, they may keep the other
// key than exact arithmetic on the same layer input does, and the row's message then moves by ~p_k |v_a - v_b|
// (profiles/parity_r3.txt: 19 such rows in 262 144 at BASELINE configs[1]).
//
//   1. topk_threshold() (attention.hip) flags a row when more than k of its logits lie at or above thr - mdgat_near_eps (one
//      more counting pass per tile: the (k+1)-th largest logit is then within near_eps of the threshold, or tied with the
//      k-th); the row goes to a list.  ~1 row in 10^3.
//   2. topk_repair_kernel, one wave per listed row: computes the logits of the row's query tile AGAIN - the same matrix
//      instructions on the same fragments in the same order as the attention kernel, hence its logits bit for bit - and
//      collects the row's CANDIDATES: the logits inside [thr - near_eps, thr + near_eps], each with the side of the threshold
//      it is on (= whether the attention kernel kept it); everything above the window is kept and everything below dropped
//      whatever the arithmetic.
//   3. q of the row and k of every candidate are projected again from the layer's fp32 input descriptors with the fp64
//      weights (fp32 head + fp32 residual of each weight: 48 bits) in fp64; the exact logits order the candidates (equal
//      logits - duplicated keypoints: lowest key index first, the kernels' tie rule).  Where that order keeps other keys
//      than the kernel did (~1 listed row in 50), the share of the wrongly kept keys is taken out of the written message
//      row and that of the wrongly dropped ones put in:
//          o <- (o l - sum P'v [out] + sum P'v [in]) / (l - sum P' [out] + sum P' [in]).
//      Exact ties at the k-th place (a row that kept more than k logits) are resolved by the same step.
// What remains different from the reference after this are rows whose order flips with the layer INPUT (error accumulated
// by the layers before: fp32-class descriptors against the reference's fp64 ones) - nothing local can see those
// (profiles/parity_r4.txt).
//
// Why a kernel of its own: the same steps inlined into the attention kernels, or called from them out of line, cost the
// 512-key kernel its register allocation (150 -> 320-390 us per launch, whether or not a row is flagged); a list append does not.
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int RP_WAVES = 4;          // rows per workgroup pass
constexpr int NEAR_MAXC = 16;        // candidates per row (more - masses of equal logits: the row is left as the kernel wrote it)

struct NearRow {                     // LDS, one per wave
    int n;                           // candidates found (> NEAR_MAXC: given up)
    int above;                       // logits above the window
    unsigned key[NEAR_MAXC];         // key index within the source frame | (kept by the attention kernel) << 31
    float s[NEAR_MAXC];              // the attention kernel's logit
    double qk[2][32];                // q of the row, k of the candidate being evaluated (fp64)
};

struct RepairArgs {
    const _Float16 *q16, *k16, *vt16;
    float* msg;
    const float* x;
    const float *w, *wlo, *b, *blo; // qkv_w [384][128] (fp32 heads of the fp64 weights), residuals of its q | k rows [256][128]; qkv_b [384], residuals [256]
    int N, M, Npad, PP, cross, topk;
    const int* count; const RepairRec* recs; int cap;
    uint32_t* sel; int selW;
    int* stats;
    unsigned* giveup;    // optional (host-mapped status word): rows left as the attention kernel wrote them (> NEAR_MAXC candidates, or a full list)
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave: dims 0..31 of W x + b for one point in fp64 (weight = fp32 head + fp32 residual).  Lane (d8 = lane >> 3,
// c8 = lane & 7) takes channels {4 (8 j + c8) .. + 3 : j < 4} of the output rows row0 + d8 + 8 i, i < 4: eight lanes read 128
// contiguous bytes of a weight row, all 36 loads of the projection are in flight together.  out: LDS [32].
__device__ __forceinline__ void project(const RepairArgs& a, int row0, const float* xrow, int lane, double* out) {
    const int d8 = lane >> 3, c8 = lane & 7;
    f32x4 xv[4], wh[4][4], wl[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xrow + 4 * (8 * j + c8));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t off = (size_t)(row0 + d8 + 8 * i) * 128 + 4 * (8 * j + c8);
            wh[i][j] = *reinterpret_cast<const f32x4*>(a.w + off);
            wl[i][j] = *reinterpret_cast<const f32x4*>(a.wlo + off);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = fma((double)wh[i][j][0] + (double)wl[i][j][0], (double)xv[j][0], acc0);
            acc1 = fma((double)wh[i][j][1] + (double)wl[i][j][1], (double)xv[j][1], acc1);
            acc0 = fma((double)wh[i][j][2] + (double)wl[i][j][2], (double)xv[j][2], acc0);
            acc1 = fma((double)wh[i][j][3] + (double)wl[i][j][3], (double)xv[j][3], acc1);
        }
        double acc = acc0 + acc1;
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        const int row = row0 + d8 + 8 * i;
        if (c8 == 0) out[d8 + 8 * i] = acc + ((double)a.b[row] + (double)a.blo[row]);
    }
}

// one candidate of the row (called by the lanes that hold the row's logits)
__device__ __forceinline__ void offer(NearRow& R, int key, float s, float thr) {
    const int pos = atomicAdd(&R.n, 1);
    if (pos < NEAR_MAXC) { R.key[pos] = (unsigned)key | (s >= thr ? 0x80000000u : 0u); R.s[pos] = s; }
}

// The logits of row r.q's query tile again, as the attention kernel formed them (M16: frames of exactly 512 or 256 keys -
// attention_topk16_kernel, 16-query tiles, 16x16x32 products; else the 32-query tiles and 32x32x16 products of
// attention_kernel / attention_topk_wide_kernel), handed to f(key, logit) by the lanes that hold the row.
template <bool M16, typename F>
__device__ __forceinline__ void row_logits(const RepairArgs& a, const RepairRec& r, int lane, F f) {
    const int head = r.bsh & 3, side = (r.bsh >> 2) & 1, b = r.bsh >> 3;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N;
    const int q_off = side ? a.N : 0;
    const int src = a.cross ? (1 - side) : side;
    const int nk = src ? a.M : a.N;
    const int k_off = src ? a.N : 0;
    if (M16) {
        const int l15 = lane & 15, g = lane >> 4;
        const int qt = r.q & ~15;
        const bool mine = l15 == r.q - qt;
        f16x8 qh, ql;
        {
            const int qrow = min(qt + l15, nq - 1);
            const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * g;
            qh = *reinterpret_cast<const f16x8*>(p);
            ql = *reinterpret_cast<const f16x8*>(p + 32);
        }
        const _Float16* kbase = a.k16 + (((size_t)b * P + k_off + l15) * 4 + head) * 64 + 8 * g;
#pragma unroll 1
        for (int half = 0; half < nk / 256; ++half) {   // 16 blocks of 16 keys per batch of loads (nk = 512 or 256)
            f16x8 kh[16], kl[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const _Float16* kp = kbase + (size_t)(16 * (16 * half + i)) * 256;
                kh[i] = *reinterpret_cast<const f16x8*>(kp);
                kl[i] = *reinterpret_cast<const f16x8*>(kp + 32);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acx = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[i], qh, acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[i], ql, acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[i], qh, acx, 0, 0, 0);
                if (mine) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) f(16 * (16 * half + i) + 4 * g + rr, acc[rr] + acx[rr]);
                }
            }
        }
    } else {
        const int l31 = lane & 31, hi = lane >> 5;
        const int qt = r.q & ~31;
        const bool mine = l31 == r.q - qt;
        const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2 <-> 3 (attention.hip)
        f16x8 qh[2], ql[2];
        {
            const int qrow = min(qt + l31, nq - 1);
            const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * hi;
            qh[0] = *reinterpret_cast<const f16x8*>(p);
            qh[1] = *reinterpret_cast<const f16x8*>(p + 16);
            ql[0] = *reinterpret_cast<const f16x8*>(p + 32);
            ql[1] = *reinterpret_cast<const f16x8*>(p + 48);
        }
        const _Float16* kg = a.k16 + (((size_t)b * P + k_off) * 4 + head) * 64 + 8 * hi;
        const int nblk = (nk + 31) >> 5;
#pragma unroll 1
        for (int jb0 = 0; jb0 < nblk; jb0 += 4) {       // four blocks of 32 keys per batch of loads
            f16x8 k[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = min((jb0 + u) * 32 + krow, nk - 1);     // rows past the end: any finite data, skipped below
                const _Float16* kp = kg + (size_t)key * 256;
                k[u][0] = *reinterpret_cast<const f16x8*>(kp);
                k[u][1] = *reinterpret_cast<const f16x8*>(kp + 16);
                k[u][2] = *reinterpret_cast<const f16x8*>(kp + 32);
                k[u][3] = *reinterpret_cast<const f16x8*>(kp + 48);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x16 acc, acx;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) { acc[rr] = 0.f; acx[rr] = 0.f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[u][0], qh[0], acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[u][0], ql[0], acx, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[u][1], qh[1], acc, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[u][1], ql[1], acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[u][2], qh[0], acx, 0, 0, 0);
                acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[u][3], qh[1], acx, 0, 0, 0);
                if (mine) {
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        const int key = (jb0 + u) * 32 + 8 * hi + KeyLayout32::koff(0, rr);
                        if (key < nk) f(key, acc[rr] + acx[rr]);
                    }
                }
            }
        }
    }
}

template <bool M16>
__global__ __launch_bounds__(64 * RP_WAVES, 2) void topk_repair_kernel(RepairArgs a) {
    __shared__ NearRow rows[RP_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    NearRow& R = rows[wave];
    // (the record is requested before the count is known: one round trip for both)
    const int first = blockIdx.x * RP_WAVES + wave;
    RepairRec r = a.recs[first < a.cap ? first : 0];
    int n = *a.count;
    if (n > a.cap) {
        if (a.giveup && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.giveup, (unsigned)(n - a.cap));      // rows that did not fit the list
        n = a.cap;
    }
    const int P = a.N + a.M;
    for (int rec = first; rec < n; rec += gridDim.x * RP_WAVES) {
        if (rec != first) r = a.recs[rec];
        const int head = r.bsh & 3, side = (r.bsh >> 2) & 1, b = r.bsh >> 3;
        const int q_off = side ? a.N : 0;
        const int src = a.cross ? (1 - side) : side;
        const int k_off = src ? a.N : 0;
        const float thr = r.thr, m11 = r.m - 11.0f;
        const float eps = mdgat_near_eps(r.thr, r.m);
        const float lo = thr - eps, hiw = thr + eps;
        if (lane == 0) { R.n = 0; R.above = 0; }
        __builtin_amdgcn_wave_barrier();

        // ---- 2. the candidates of the row ----
        {
            int above = 0;
            row_logits<M16>(a, r, lane, [&](int key, float sv) {
                above += sv > hiw;
                if (sv >= lo && sv <= hiw) offer(R, key, sv, thr);
            });
            if (above) atomicAdd(&R.above, above);
        }
        __builtin_amdgcn_wave_barrier();
        const int nc = R.n;
        if (a.stats && lane == 0) {
            atomicAdd(a.stats + 0, 1);
            if (nc > NEAR_MAXC) atomicAdd(a.stats + 3, 1);
        }
        if (nc > NEAR_MAXC) {
            if (a.giveup && lane == 0) atomicAdd(a.giveup, 1u);
            continue;
        }
        const int need = a.topk - R.above;          // candidates to keep
        // the candidates arrived in the order of the LDS atomics: sort them by key, so that the sums below have one order
        if (lane == 0) {
            for (int i = 1; i < nc; ++i) {
                const unsigned k = R.key[i];
                const float s = R.s[i];
                int j = i - 1;
                while (j >= 0 && (R.key[j] & 0x7fffffffu) > (k & 0x7fffffffu)) { R.key[j + 1] = R.key[j]; R.s[j + 1] = R.s[j]; --j; }
                R.key[j + 1] = k; R.s[j + 1] = s;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned my_ent = lane < nc ? R.key[lane] : 0u;
        const int my_key = (int)(my_ent & 0x7fffffffu);
        const bool kept_now = (my_ent >> 31) != 0;

        // ---- 3. exact logits of the candidates (a positive multiple of them: the q scale is left out) ----
        project(a, head * 32, a.x + ((size_t)b * P + q_off + r.q) * 128, lane, R.qk[0]);
        double my_l = -__builtin_inf();
#pragma unroll 1
        for (int t = 0; t < nc; ++t) {
            const int key = (int)(R.key[t] & 0x7fffffffu);
            project(a, 128 + head * 32, a.x + ((size_t)b * P + k_off + key) * 128, lane, R.qk[1]);
            __builtin_amdgcn_wave_barrier();
            const double dot = wave_sum_d(lane < 32 ? R.qk[0][lane] * R.qk[1][lane] : 0.0);
            if (lane == t) my_l = dot;
            __builtin_amdgcn_wave_barrier();
        }
        int rank = 0;                               // descending; equal logits: lowest key index first
        for (int u = 0; u < nc; ++u) {
            const double lu = __shfl(my_l, u, 64);
            const int ku = __shfl(my_key, u, 64);
            rank += (lu > my_l) || (lu == my_l && ku < my_key);
        }
        const bool keep = lane < nc && rank < need;
        const unsigned long long cm = __ballot(lane < nc && keep != kept_now);
        if (cm == 0) continue;                      // the attention kernel kept what exact arithmetic keeps
        if (a.stats && lane == 0) atomicAdd(a.stats + 1, 1);

        // ---- take the wrongly kept keys out of the written row, put the wrongly dropped ones in (lanes 0..31: one dim each) ----
        // l: the row's sum of P' = exp2(s - m11) over the kept logits, as the pass formed it (up to the order of its additions)
        float lsum = 0.f;
        row_logits<M16>(a, r, lane, [&](int, float sv) { lsum += sv >= thr ? __builtin_amdgcn_exp2f(sv - m11) : 0.f; });
        lsum += __shfl_xor(lsum, 16, 64);           // (lanes that do not hold the row contribute nothing)
        lsum += __shfl_xor(lsum, 32, 64);
        lsum = __shfl(lsum, M16 ? (r.q & 15) : (r.q & 31), 64);
        const _Float16* vrow = a.vt16 + (((size_t)b * 4 + head) * 64 + (lane & 31)) * a.PP + (src ? a.Npad : 0);
        float acc = 0.f, el = 0.f;
        for (unsigned long long mk = cm; mk; mk &= mk - 1) {
            const int t = __builtin_ctzll(mk);
            const int key = (int)(R.key[t] & 0x7fffffffu);
            const float sign = (R.key[t] >> 31) ? -1.f : 1.f;          // kept by the kernel: it goes out
            const float e = __builtin_amdgcn_exp2f(R.s[t] - m11);      // P' as the pass computed it
            const float eh = (float)(_Float16)e, elo = (float)(_Float16)(e - eh);
            const float vh = (float)vrow[key], vl = (float)vrow[(size_t)32 * a.PP + key];
            acc = fmaf(sign, fmaf(eh, vl, fmaf(elo, vh, eh * vh)), acc);
            el = fmaf(sign, e, el);
        }
        float* o = a.msg + ((size_t)b * P + q_off + r.q) * 128 + head * 32 + lane;
        if (lane < 32) *o = (*o * lsum + acc) / (lsum + el);
        if (a.sel && lane == 0) {                   // parity tap: the row's final selection
            uint32_t* row = a.sel + (((size_t)b * 4 + head) * P + q_off + r.q) * a.selW;
            for (unsigned long long mk = cm; mk; mk &= mk - 1) {
                const int t = __builtin_ctzll(mk);
                const unsigned key = R.key[t] & 0x7fffffffu;
                if (R.key[t] >> 31) atomicAnd(row + (key >> 5), ~(1u << (key & 31)));
                else atomicOr(row + (key >> 5), 1u << (key & 31));
            }
        }
    }
}

}  // namespace

int launch_topk_repair(const RepairLaunch& p, hipStream_t s) {
    if (p.B <= 0 || p.topk <= 0 || !p.near.count) return MDGAT_OK;
    const int nk_max = p.N > p.M ? p.N : p.M;
    if (nk_max > 2048) return MDGAT_OK;                 // (the dynamic kernels reject such frames before this point)
    RepairArgs a{p.qkv.q16, p.qkv.k16, p.qkv.vt16, p.msg, p.x, p.w, p.wlo, p.b, p.blo, p.N, p.M, p.qkv.Npad, p.qkv.PP, p.cross, p.topk,
                 p.near.count, p.near.recs, p.near.cap, p.sel, (nk_max + 31) / 32, p.stats, p.giveup};
    // one wave per listed row (~1 row in 10^3 is listed); a launch that finds the list empty leaves at once
    const long rows = (long)p.B * (p.N + p.M) * 4;
    int blocks = (int)((rows / 256 + RP_WAVES - 1) / RP_WAVES);
    if (blocks < 4) blocks = 4;
    if (blocks > 1024) blocks = 1024;
    // which fragments the attention kernel formed (launch_attention: the 16-query kernel for N = M = 512 and N = M = 256)
    if ((p.N == 512 && p.M == 512) || (p.N == 256 && p.M == 256)) hipLaunchKernelGGL(topk_repair_kernel<true>, dim3(blocks), dim3(64 * RP_WAVES), 0, s, a);
    else hipLaunchKernelGGL(topk_repair_kernel<false>, dim3(blocks), dim3(64 * RP_WAVES), 0, s, a);
    return mdgat_check_hip(hipGetLastError(), "top-k repair launch");
}
