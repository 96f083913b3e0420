// Exact k-th largest of one attention row, one wave per row (f64.hip: the reference-exact mode).  dynamic_attention
// (mdgat.py:196-210) keeps the k largest logits of a row; the fp64 attention kernel parks the fp32 roundings of a tile's logits in LDS
// and lets ONE wave select on each row from registers - no exchange between waves - instead of searching 16 or 32 rows in lockstep
// across the lanes and waves that hold them in matrix-fragment layout (attention.hip's fp32-class kernels do that; three variants of
// this scheme for their 2048-key kernel were measured and lost: profiles/NOTES_r5.md section 3).
#pragma once
#include <hip/hip_runtime.h>
#include "common.hpp"
#ifndef FT
#define FT(k) do {} while (0)
#endif

// monotone image of a float in the unsigned integers (larger float <-> larger integer; -inf below every finite value)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned b = __builtin_bit_cast(unsigned, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, b);
}

// wave-wide reductions / scans on the vector ALU (DPP row shifts + row broadcasts).  After the six steps lane l holds the inclusive
// prefix over lanes 0..l, lane 63 the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int rs_dpp(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false); }
template <typename Op>
__device__ __forceinline__ int rs_wave_scan(int v, int identity, Op op) {
    v = op(v, rs_dpp<0x111, 0xf>(identity, v));   // row_shr:1
    v = op(v, rs_dpp<0x112, 0xf>(identity, v));   // row_shr:2
    v = op(v, rs_dpp<0x114, 0xf>(identity, v));   // row_shr:4
    v = op(v, rs_dpp<0x118, 0xf>(identity, v));   // row_shr:8
    v = op(v, rs_dpp<0x142, 0xa>(identity, v));   // row_bcast:15 -> rows 1, 3
    v = op(v, rs_dpp<0x143, 0xc>(identity, v));   // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ unsigned rs_wave_max_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_readlane(rs_wave_scan((int)v, 0, [](int a, int b) { return (int)max((unsigned)a, (unsigned)b); }), 63);
}
__device__ __forceinline__ unsigned rs_wave_min_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_readlane(rs_wave_scan((int)v, -1, [](int a, int b) { return (int)min((unsigned)a, (unsigned)b); }), 63);
}
// (a float sum through the same six steps; returned as bits so that the callers' lambdas stay integer-typed)
__device__ __forceinline__ int rs_wave_sum_i_bits(float v) {
    auto f = [](int x) { return __builtin_bit_cast(float, x); };
    const int r = rs_wave_scan(__builtin_bit_cast(int, v), 0, [&](int a, int b) { return __builtin_bit_cast(int, f(a) + f(b)); });
    return __builtin_amdgcn_readlane(r, 63);
}
__device__ __forceinline__ int rs_wave_sum_i(int v) { return __builtin_amdgcn_readlane(rs_wave_scan(v, 0, [](int a, int b) { return a + b; }), 63); }

struct RowSearch {
    float thr;         // the k-th largest value of the row (-inf: every key is kept).  count(>= thr) = c_ge >= k > c_gt = count(> thr)
    int c_ge, c_gt;
    float mx;          // largest value of the row
    int keylim;        // c_ge > k and more than list_cap values equal thr: of those the ones with key <= keylim are the first k - c_gt in key
                       // order (torch.topk keeps exactly k; which of several equal logits is unspecified there - here the lowest keys)
};

// row: nk floats in LDS, nk <= 64 NV (-0.0 counts as +0.0); hist: RS_HIST_INTS ints of LDS owned by this wave.  Every lane returns the same answer.
// RADIX SELECT on the monotone integer images of the floats, eight bits a level: a 256-bin histogram of the values still in play
// (LDS atomics), a DPP suffix scan that finds the bin holding the k-th largest, and down into that bin - until it holds one value or
// the digits run out (equal values: the tie).  Three levels on logits of order one, ~170 instructions each, and the SAME work for
// every row: the bracketing searches this replaces (probe a threshold, count, Newton / interpolation / bisection) average 5 probes
// but have a tail of 20-30, and a tile waits at a barrier for its slowest row (measured at 2048 keys: 36 000 cycles of search and
// 40 000 of waiting per tile; profiles/NOTES_r5.md section 3).
typedef __attribute__((address_space(3))) int rs_lds_int;
typedef __attribute__((address_space(3))) const float rs_lds_cfloat;
constexpr int RS_HIST_INTS = 320;       // 256 bins + one waste bin per lane
template <int NV>
__device__ RowSearch topk_row_search(const float* row_, int nk, int k, float zq, int lane, int list_cap, int* hist_) {
    // (both pointers are LDS: said here, or the function addresses them as flat memory)
    rs_lds_cfloat* row = (rs_lds_cfloat*)row_;
    rs_lds_int* hist = (rs_lds_int*)hist_;
    // (the arguments are the same in every lane; as a function's parameters they arrive in vector registers)
    nk = __builtin_amdgcn_readfirstlane(nk);
    k = __builtin_amdgcn_readfirstlane(k);
    list_cap = __builtin_amdgcn_readfirstlane(list_cap);
    zq = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zq)));
    unsigned o[NV];                      // pads: 0, below the image of every float (-inf is 0x007fffff)
    unsigned omn = ~0u, omx = 0u;
    float s1 = 0.f, s2 = 0.f;            // moments of the row from every STEPth register: they only choose where the select starts
    constexpr int STEP = NV >= 8 ? 4 : 1;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i;
        const float f = idx < nk ? row[idx] + 0.f : 0.f;
        o[i] = idx < nk ? f2ord(f) : 0u;
        omn = min(omn, idx < nk ? o[i] : ~0u); omx = max(omx, o[i]);
        if (i % STEP == 0) { const float g = f > -3.0e38f ? f : 0.f; s1 += g; s2 = fmaf(g, g, s2); }
    }
    omn = rs_wave_min_u(omn); omx = rs_wave_max_u(omx);
    RowSearch out{-__builtin_inff(), nk, 0, ord2f(omx), 1 << 30};
    if (k >= nk) return out;             // keep everything
    // Where the select starts.  The top byte of a float's image is sign and exponent: over the whole row [min, max] the first level's
    // digits pile most of the values into two or three bins, and 64 lanes adding to one LDS address serialise (measured on rows of
    // 2048: 25 000 cycles a row).  The k-th largest of a bell-shaped row sits near mean + zq sd: the select starts one deviation
    // below that, where the values left in play (a third of the row for k = nk / 16) spread over a hundred bins and the first
    // level already resolves 2^-6 of an octave; a row that does not hold k values above the start starts over from its minimum.
    unsigned base = omn;
    {
        s1 = __builtin_bit_cast(float, rs_wave_sum_i_bits(s1)); s2 = __builtin_bit_cast(float, rs_wave_sum_i_bits(s2));
        const float inv_n = (float)STEP / (float)nk;
        const float mu = s1 * inv_n;
        const float sd = sqrtf(fmaxf(s2 * inv_n - mu * mu, 0.f));
        const unsigned cand = f2ord(mu + (zq - 1.0f) * sd);
        if (cand > omn && cand < omx) base = cand;
    }
    int sh = 24 - __builtin_clz((omx - base) | 0xffu);       // ((omx - base) >> sh) < 256
    int above = 0, ceq = 0;
    unsigned width = 256u;               // digits in play at this level: the last step of a bin narrower than 256 values has fewer
    for (;;) {
        hist[4 * lane] = 0; hist[4 * lane + 1] = 0; hist[4 * lane + 2] = 0; hist[4 * lane + 3] = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            // (values out of play - below base, the pads among them, or beyond the 256 digits - go to this lane's waste bin: an
            // unconditional atomic instead of a branch per value, and no two lanes on one waste address)
            const unsigned d = (o[i] - base) >> sh;
            __atomic_fetch_add(hist + ((o[i] >= base && d < width) ? (int)d : 256 + lane), 1, __ATOMIC_RELAXED);
        }
        int4 h;
        h.x = hist[4 * lane]; h.y = hist[4 * lane + 1]; h.z = hist[4 * lane + 2]; h.w = hist[4 * lane + 3];
        const int pre = rs_wave_scan(h.x + h.y + h.z + h.w, 0, [](int a, int b) { return a + b; });
        const int tot = __builtin_amdgcn_readlane(pre, 63);
        if (above + tot < k) {            // (first level only: fewer than k values above the start - not a bell-shaped row)
            base = omn;
            sh = 24 - __builtin_clz((omx - omn) | 0xffu);
            continue;
        }
        // values in play above this lane's four bins, then bin by bin from the top
        const int a3 = above + tot - pre, a2 = a3 + h.w, a1 = a2 + h.z, a0 = a1 + h.y;
        int bin = -1, ab = 0, cnt = 0;
        if (a0 < k && k <= a0 + h.x) { bin = 4 * lane; ab = a0; cnt = h.x; }
        if (a1 < k && k <= a1 + h.y) { bin = 4 * lane + 1; ab = a1; cnt = h.y; }
        if (a2 < k && k <= a2 + h.z) { bin = 4 * lane + 2; ab = a2; cnt = h.z; }
        if (a3 < k && k <= a3 + h.w) { bin = 4 * lane + 3; ab = a3; cnt = h.w; }
        const int src = __builtin_ctzll(__ballot(bin >= 0));          // exactly one lane holds the bin of the k-th largest
        bin = __builtin_amdgcn_readlane(bin, src);
        above = __builtin_amdgcn_readlane(ab, src);
        ceq = __builtin_amdgcn_readlane(cnt, src);
        base += (unsigned)bin << sh;
        if (sh == 0) break;                                           // the bin is one value: `ceq` values equal it
        if (ceq == 1) {                                               // one value left in play: it is the k-th largest
            unsigned m = 0u;
#pragma unroll
            for (int i = 0; i < NV; ++i) m = max(m, (o[i] >= base && ((o[i] - base) >> sh) == 0u) ? o[i] : 0u);
            base = rs_wave_max_u(m);
            break;
        }
        width = sh >= 8 ? 256u : 1u << sh;      // (the bin just chosen spans 2^sh values: the next level must not look beyond it)
        sh = sh > 8 ? sh - 8 : 0;
    }
    out.thr = ord2f(base); out.c_gt = above; out.c_ge = above + ceq;
    if (out.c_ge == k || ceq <= list_cap) return out;
    // more than list_cap values share the k-th place: the first k - c_gt of them in key order stay (key = lane + 64 i)
    const int need = k - above;
    int seen = 0, keylim = nk;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        unsigned long long mask = __ballot(o[i] == base);
        const int c = __popcll(mask);
        if (keylim == nk && seen + c >= need) {
            for (int n = need - seen; n > 1; --n) mask &= mask - 1;
            keylim = 64 * i + __builtin_ctzll(mask);
        }
        seen += c;
    }
    out.keylim = keylim;
    return out;
}

// ---- four rows per wave at once (rows of at most 512 values) ------------------------------------------------------------------
// The same radix select with a row on SIXTEEN lanes (a DPP row) and the four rows of a wave side by side in one instruction
// stream.  One row per wave (above) is a chain of latencies - four DPP reductions, then per level LDS clear -> atomics -> wait ->
// read -> scan -> three readlanes - paid four times over, one row after the other, between the two barriers of a tile
// (profiles/NOTES_r5.md section 10: the select is 31 % of the 512-key dynamic kernel and it is latency, not issue).  Here every step
// serves four rows: the reductions are rotations inside the DPP row (every lane ends up with the row's value: no readlane), a
// level is ONE round trip for the four histograms, and rows that need another level or restart from their minimum simply stay
// active (per-lane control flow; each row's histogram is its own LDS region, so rows never meet).
// Lane s of a row holds the values s, s + 16, s + 32, ...: NV per lane, nk <= 16 NV.
template <typename Op>
__device__ __forceinline__ int rq_allreduce(int v, Op op) {      // over the 16 lanes of a DPP row, by rotations: every lane gets the result
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));      // row_ror:8
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));      // row_ror:4
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));      // row_ror:2
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));      // row_ror:1
    return v;
}
__device__ __forceinline__ int rq_sum(int v) { return rq_allreduce(v, [](int a, int b) { return a + b; }); }
__device__ __forceinline__ unsigned rq_max_u(unsigned v) { return (unsigned)rq_allreduce((int)v, [](int a, int b) { return (int)max((unsigned)a, (unsigned)b); }); }
__device__ __forceinline__ unsigned rq_min_u(unsigned v) { return (unsigned)rq_allreduce((int)v, [](int a, int b) { return (int)min((unsigned)a, (unsigned)b); }); }
__device__ __forceinline__ float rq_sum_f(float v) {
    auto f = [](int x) { return __builtin_bit_cast(float, x); };
    return f(rq_allreduce(__builtin_bit_cast(int, v), [&](int a, int b) { return __builtin_bit_cast(int, f(a) + f(b)); }));
}
// inclusive suffix sum over the row: lane s gets the sum over lanes s .. 15 (row_shl:n reads lane s + n; beyond the row: 0)
__device__ __forceinline__ int rq_suffix_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x102, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x108, 0xf, 0xf, false);
    return v;
}
// this row's 16 bits of a wave ballot
__device__ __forceinline__ unsigned rq_row_bits(unsigned long long ballot, int lane) { return (unsigned)(ballot >> (lane & 48)) & 0xffffu; }

constexpr int RQ_HIST_INTS = 272;       // per row: 256 bins + one waste bin per lane
#ifndef RQ_START_BELOW
#define RQ_START_BELOW 0.25f
#endif
// rows: the wave's four rows in LDS, `pitch` floats apart (row r of the wave = lanes 16 r .. 16 r + 15); hist: 4 x RQ_HIST_INTS ints of LDS
// owned by this wave (it may be the rows' own storage: every value is in registers before the first histogram is cleared, and the
// LDS serves a wave's accesses in order).  Every lane returns its row's answer.
// Instruction count matters here as much as latency (PMC, 512 keys: the select was 1 550 of the dynamic kernel's 2 500 vector
// instructions per wave and tile in its first form): the row is loaded without per-value branches (whole rows without any guard),
// and a level's base is ALIGNED to its digit position, so that a value's bin is (o >> sh) - (base >> sh) - two instructions, values
// below the base wrap to huge bins - and one compare against the level's width sends everything out of play to the lane's waste bin.
template <int NV>
__device__ RowSearch topk_quad_search(const float* rows_, int pitch, int nk, int k, float zq, int lane, int list_cap, int* hist_, int hist_pitch) {
    const int s = lane & 15, rw = lane >> 4;
    rs_lds_cfloat* row = (rs_lds_cfloat*)rows_ + rw * pitch;
    unsigned o[NV];                      // pads: 0, below the image of every float
    unsigned omx = 0u;
    float s1 = 0.f, s2 = 0.f;            // moments of the row from every other register: they only choose where the select starts
    if (nk == 16 * NV) {                 // (a whole row: 512 / 256 / 128 keys)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float f = row[s + 16 * i] + 0.f;
            o[i] = f2ord(f);
            omx = max(omx, o[i]);
            if (i % 2 == 0) { const float g = f > -3.0e38f ? f : 0.f; s1 += g; s2 = fmaf(g, g, s2); }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = s + 16 * i;
            const bool valid = idx < nk;
            const float f = row[min(idx, nk - 1)] + 0.f;       // (an address inside the row either way: no branch around the load)
            o[i] = valid ? f2ord(f) : 0u;
            omx = max(omx, o[i]);
            if (i % 2 == 0) { const float g = (valid && f > -3.0e38f) ? f : 0.f; s1 += g; s2 = fmaf(g, g, s2); }
        }
    }
    FT(8);
    omx = rq_max_u(omx);
    // the row's smallest value: only a row that restarts needs it (rare: computed there)
    auto row_min = [&]() {
        unsigned mn = ~0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) mn = min(mn, o[i] ? o[i] : ~0u);       // (pads are 0; no float's image is)
        return rq_min_u(mn);
    };
    RowSearch out{-__builtin_inff(), nk, 0, ord2f(omx), 1 << 30};
    if (k >= nk) return out;             // (the same for every row of the launch)
    rs_lds_int* hist = (rs_lds_int*)hist_ + rw * hist_pitch;
    unsigned base;
    int sh;
    // digit position for the values lo .. omx, the base aligned to it (eight bits a level: ((omx - base) >> sh) < 256)
    auto set_range = [&](unsigned lo) {
        int h = 24 - __builtin_clz((omx - lo) | 0xffu);
        unsigned b = lo & ~((1u << h) - 1u);
        if (((omx - b) >> h) >= 256u) { ++h; b = lo & ~((1u << h) - 1u); }
        base = b; sh = h;
    };
    {
        s1 = rq_sum_f(s1); s2 = rq_sum_f(s2);
        const float inv_n = 2.0f / (float)nk;
        const float mu = s1 * inv_n;
        const float sd = sqrtf(fmaxf(s2 * inv_n - mu * mu, 0.f));
        // Where the select starts.  A level's cost is its values IN PLAY: their atomics meet in the banks of one histogram (the values
        // out of play go to sixteen neighbouring waste bins and do not) - phase trace at 512 keys: 25 000 cycles of select for k = 128
        // (63 % in play at mean + (zq - 1) sd), 17 000 for k = 64 (44 %).  The k-th largest of a bell-shaped row sits near mean + zq
        // sd; with the moments taken over the whole row the start a quarter of a deviation below it still holds k values in all
        // but a few rows in a thousand (a row that does not starts over from its minimum), and a third of the row is in play.
        const unsigned cand = f2ord(mu + (zq - RQ_START_BELOW) * sd);
        if (cand < omx) set_range(cand); else set_range(row_min());
    }
    int above = 0, ceq = 0;
    unsigned width = 256u;
    FT(9);
    int ft_level = 0;                    // (only the -DF64_TRACE build reads it)
    (void)ft_level;
    for (;;) {
        FT(10 + 3 * min(ft_level, 2));
        typedef int rq_i4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) rq_i4 rq_lds_int4;
        rq_lds_int4* h4 = (rq_lds_int4*)hist;
        h4[4 * s] = rq_i4{0, 0, 0, 0}; h4[4 * s + 1] = rq_i4{0, 0, 0, 0}; h4[4 * s + 2] = rq_i4{0, 0, 0, 0}; h4[4 * s + 3] = rq_i4{0, 0, 0, 0};
        hist[256 + s] = 0;
        const unsigned bq = base >> sh;
        // (values out of play go to the lane's waste bin: an unconditional atomic.  Issued under the lane mask of the values in play
        // instead - a branch per value - the select was slower: 680 -> 795 us per launch at batch 64)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned d = (o[i] >> sh) - bq;               // below the base: wraps far beyond the width
            __atomic_fetch_add(hist + (d < width ? (int)d : 256 + s), 1, __ATOMIC_RELAXED);
        }
        FT(11 + 3 * min(ft_level, 2));
        // stage 1: this lane's sixteen bins (16 s .. 16 s + 15) as one count; the lane whose bins hold the k-th largest
        const rq_i4 ha = h4[4 * s], hb = h4[4 * s + 1], hc = h4[4 * s + 2], hd = h4[4 * s + 3];
        const int t = (ha.x + ha.y + ha.z + ha.w) + (hb.x + hb.y + hb.z + hb.w) + (hc.x + hc.y + hc.z + hc.w) + (hd.x + hd.y + hd.z + hd.w);
        const int suf = rq_suffix_sum(t);
        const int tot = rq_sum(t);
        if (above + tot < k) {           // (first level only: fewer than k values above the start - not a bell-shaped row)
            set_range(1u);               // (everything real is in play again: pads are 0; coarse first digits, more levels - rare)
            continue;
        }
        const int al = above + suf - t;                              // values in play above this lane's bins
        const bool own = al < k && k <= al + t;                      // exactly one lane of the row
        const int ls = __builtin_ctz(rq_row_bits(__ballot(own), lane) | 0x10000u);
        const int as = rq_sum(own ? al : 0);
        // stage 2: that lane's sixteen bins, one per lane
        const int h1 = hist[16 * ls + s];
        const int a1 = as + rq_suffix_sum(h1) - h1;
        const bool own1 = a1 < k && k <= a1 + h1;
        const int bin = 16 * ls + __builtin_ctz(rq_row_bits(__ballot(own1), lane) | 0x10000u);
        above = rq_sum(own1 ? a1 : 0);
        ceq = rq_sum(own1 ? h1 : 0);
        base += (unsigned)bin << sh;
        FT(12 + 3 * min(ft_level, 2));
        ++ft_level;
        if (sh == 0) break;                                          // the bin is one value: `ceq` values equal it
        if (ceq == 1) {                                              // one value left in play: it is the k-th largest
            const unsigned bb = base >> sh;
            unsigned m = 0u;
#pragma unroll
            for (int i = 0; i < NV; ++i) m = max(m, (o[i] >> sh) == bb ? o[i] : 0u);
            base = rq_max_u(m);
            break;
        }
        width = sh >= 8 ? 256u : 1u << sh;      // (the bin just chosen spans 2^sh values: the next level must not look beyond it)
        sh = sh > 8 ? sh - 8 : 0;
    }
    FT(20);
    out.thr = ord2f(base); out.c_gt = above; out.c_ge = above + ceq;
    if (out.c_ge == k || ceq <= list_cap) return out;
    // more than list_cap values share the k-th place: the first k - c_gt of them in key order stay (key = s + 16 i)
    const int need = k - above;
    int seen = 0, keylim = nk;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        unsigned mask = rq_row_bits(__ballot(o[i] == base), lane);
        const int c = __popc(mask);
        if (keylim == nk && seen + c >= need) {
            for (int n = need - seen; n > 1; --n) mask &= mask - 1;
            keylim = 16 * i + __builtin_ctz(mask);
        }
        seen += c;
    }
    out.keylim = keylim;
    return out;
}
