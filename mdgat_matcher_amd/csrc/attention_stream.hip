// Full attention (mdgat.py:190-194) as a STREAM of 64-key chunks, for frames whose key counts are multiples of 64.
//
// attention.hip stages all keys of a (pair, frame, head) before the first product, with one workgroup per CU: the
// chip pulls K and V at the memory rate while every matrix core waits, then computes while the memory system idles
// (measured: 25 of 76 us per launch are the staging bursts, and the QK^T products of the two waves of a SIMD run
// at the same time, so they do not hide behind the other wave's softmax either).  Here
//   * a workgroup is 4 waves (128 queries); two workgroups share a CU (67.6 KB of LDS, <= 256 registers), so the
//     prologue of one runs beside the main loop of the other;
//   * K and V^T arrive chunk by chunk through a 4-slot ring filled by global_load_lds_dwordx4 three chunks ahead;
//   * inside a wave the QK^T products of chunk c + 1 and the P.V products of chunk c are issued BETWEEN the vector
//     instructions of the softmax of chunk c, a few at a time (sched_barrier after every slot), so that the matrix
//     pipe and the vector ALU of a SIMD are both fed continuously;
//   * the output is accumulated transposed, O^T = V^T P^T: a lane then holds 16 dims of ITS OWN query, and the
//     online-softmax rescale is a plain multiply (no cross-lane traffic); the tile goes out through LDS as whole
//     128-byte head slices.
// Arithmetic is that of attention.hip: split f16 operands with unscaled residual planes (x = hi + lo, common.hpp), three
// MFMAs per product.
//
// LDS chunk = four blocks of 32 rows x 128 B (K keys 0-31, K keys 32-63, V^T plane hi, V^T plane lo; a V^T row is
// the 64 keys of one (plane, dim)).  A block is four 1 KB copy pieces (8 rows each); inside a piece the 16-byte
// unit c of row r sits at position c ^ r, and pieces 2, 3 are shifted by 128 B: the 16 lanes of every ds_read_b128
// service group ({0-3, 12-15, 20-27}, ...; MI355X guide) then hit 16 different bank groups for both fragment shapes.
#include "common.hpp"
#include <utility>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f16x8 __attribute__((may_alias)) f16x8_a;
typedef f32x4 __attribute__((may_alias)) f32x4_a;

constexpr int BLK_BYTES = 4096 + 128;
constexpr int CHUNK_BYTES = 4 * BLK_BYTES;      // 16,896
constexpr int NSLOT = 4;
constexpr int OROW = 36;                        // floats per row of the output tile (32 dims + 16 B pad)

struct StreamArgs {
    const _Float16* q16;   // [B][P][4][2][32]   pre-scaled by log2(e)/sqrt(32)
    const _Float16* k16;   // [B][P][4][2][32]
    const _Float16* vt16;  // [B][4][2][32][PP]
    float* msg;            // [B][P][128]
    int N, M, Npad, PP, cross, QT;
};

template <typename F, int... U>
__device__ __forceinline__ void for_each_unit(F&& f, std::integer_sequence<int, U...>) { (f(U), ...); }
template <int N, typename F>
__device__ __forceinline__ void for_units(F&& f) { for_each_unit(f, std::make_integer_sequence<int, N>{}); }

#define SLOT_END() __builtin_amdgcn_sched_barrier(0)

// FAST (mdgat_attention_mode F16): only the hi planes take part - one MFMA per product, no residual planes.
// QK_ONLY (bench.py's roofline_qk leg, mdgat_attention_qk_probe): the Q K^T phase in isolation - the same chunk ring,
// K fragment reads, MFMA sequence and logit combine, but no V^T copies, no softmax and no P.V; the running row maximum
// is what leaves the kernel (so that nothing is dead code).
template <bool FAST, bool QK_ONLY = false>
__global__ __launch_bounds__(256, 2) void attention_stream_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // workgroups i, i + 8, ... (same XCD under round-robin dispatch) are the query tiles of one (pair, frame, head)
    const int i0 = blockIdx.x;
    const int grp = (i0 / (8 * a.QT)) * 8 + (i0 & 7), qt = (i0 >> 3) % a.QT;
    const int head = grp & 3, side = (grp >> 2) & 1, b = grp >> 3;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N, q_off = side ? a.N : 0;
    const int src = a.cross ? 1 - side : side;
    const int nk = src ? a.M : a.N, k_off = src ? a.N : 0;
    const int q0 = qt * 128;
    if (q0 >= nq) return;
    const int NCH = nk >> 6;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- chunk copies: wave w moves block w of every chunk (waves 0, 1: K; 2, 3: V^T planes), four pieces each ----
    const int rl8 = lane >> 3, swz = (lane & 7) ^ rl8;
    const char* dbase;
    unsigned voff, pstride, cstride;
    if (wave < 2) {
        dbase = reinterpret_cast<const char*>(a.k16 + (((size_t)b * P + k_off + wave * 32) * 4 + head) * 64);
        voff = rl8 * 512 + swz * 16; pstride = 8 * 512; cstride = 64 * 512;
    } else {
        dbase = reinterpret_cast<const char*>(a.vt16 + (((size_t)b * 4 + head) * 2 + (wave - 2)) * 32 * a.PP + (src ? a.Npad : 0));
        voff = rl8 * a.PP * 2 + swz * 16; pstride = 8 * a.PP * 2; cstride = 128;
    }
    auto dma_chunk = [&](int ch) __attribute__((always_inline)) {
        if (FAST && wave == 3) return;        // the lo plane of V^T is not read (the vmcnt waits only get stricter)
        if (QK_ONLY && wave >= 2) return;
        const char* s0 = dbase + (size_t)ch * cstride;
        const unsigned d0 = lds0 + (unsigned)(ch & (NSLOT - 1)) * CHUNK_BYTES + (unsigned)wave * BLK_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(d0 + p * 1024 + ((p >> 1) & 1) * 128), "v"(voff), "s"(s0 + (size_t)p * pstride) : "memory");
    };

    // ---- fragment addresses inside a block: row (key / dim) and 16-byte unit 2 i + hi ----
    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);     // bits 2 <-> 3: see attention.hip
    unsigned kofs[4], vofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 2 * i + hi;
        kofs[i] = (krow >> 3) * 1024 + ((krow >> 4) & 1) * 128 + ((krow & 7) * 8 + (c ^ (krow & 7))) * 16;
        vofs[i] = (l31 >> 3) * 1024 + ((l31 >> 4) & 1) * 128 + ((l31 & 7) * 8 + (c ^ (l31 & 7))) * 16;
    }
    f16x8 kf[4];            // K fragments of the block in flight: hi dims 0-15, hi 16-31, lo 0-15, lo 16-31
    f16x8 vh[2], vl[2];     // V^T fragments of two consecutive steps
    auto kread = [&](int ch, int jb) __attribute__((always_inline)) {
        const char* base = smem + (ch & (NSLOT - 1)) * CHUNK_BYTES + jb * BLK_BYTES;
#pragma unroll
        for (int i = 0; i < (FAST ? 2 : 4); ++i) kf[i] = *reinterpret_cast<const f16x8_a*>(base + kofs[i]);
    };
    auto vread = [&](int ch, int i) __attribute__((always_inline)) {
        if (QK_ONLY) return;
        const char* base = smem + (ch & (NSLOT - 1)) * CHUNK_BYTES + 2 * BLK_BYTES;
        vh[i & 1] = *reinterpret_cast<const f16x8_a*>(base + vofs[i]);
        if (!FAST) vl[i & 1] = *reinterpret_cast<const f16x8_a*>(base + BLK_BYTES + vofs[i]);
    };

    // ---- this lane's query fragments: dims 16 t + 8 hi + j of query l31, planes hi / lo ----
    const int qw = q0 + wave * 32;
    f16x8 qh[2], ql[2];
    {
        const int qrow = min(qw + l31, nq - 1);
        const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * hi;
        qh[0] = *reinterpret_cast<const f16x8*>(p);
        qh[1] = *reinterpret_cast<const f16x8*>(p + 16);
        ql[0] = *reinterpret_cast<const f16x8*>(p + 32);
        ql[1] = *reinterpret_cast<const f16x8*>(p + 48);
    }
    dma_chunk(0);
    if (NCH > 1) dma_chunk(1);
    if (NCH > 2) dma_chunk(2);
    // chunk 0 (and the query loads, which are older) landed; younger copies may stay in flight
    if (NCH > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (NCH > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float NEG_INF = -__builtin_inff();
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 Sc[2];           // logits of the chunk whose softmax is running: register r of block jb = key 32 jb + 16 (r >> 3) + 8 hi + (r & 7)
    f32x16 A, X;            // products of the block in flight (main, residual terms)
    f32x16 Om = zero16;     // O^T: register r = dim 8 (r >> 2) + 4 hi + (r & 3) of query l31 (one accumulator: the residual plane of V^T is unscaled)
    f32x2 l2 = {0.f, 0.f};
    float mxa = NEG_INF, mxb = NEG_INF;

    // QK^T product k of the block whose K fragments are in kf
    auto qk = [&](int k) __attribute__((always_inline)) {
        if (k == 0) A = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qh[0], zero16, 0, 0, 0);
        else if (FAST) { if (k == 2) A = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qh[1], A, 0, 0, 0); }
        else if (k == 1) X = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], ql[0], zero16, 0, 0, 0);
        else if (k == 2) A = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qh[1], A, 0, 0, 0);
        else if (k == 3) X = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], ql[1], X, 0, 0, 0);
        else if (k == 4) X = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2], qh[0], X, 0, 0, 0);
        else X = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[3], qh[1], X, 0, 0, 0);
    };
    // combine the products into the logits of block jb and fold them into the chunk maximum, four steps
    auto comb = [&](int jb, int u) __attribute__((always_inline)) {
        if (u < 2) {
#pragma unroll
            for (int r = 8 * u; r < 8 * u + 8; r += 2) {
                if (FAST) { Sc[jb][r] = A[r]; Sc[jb][r + 1] = A[r + 1]; continue; }
                const f32x2 v = f32x2{X[r], X[r + 1]} + f32x2{A[r], A[r + 1]};      // (unscaled residual planes: common.hpp)
                Sc[jb][r] = v[0]; Sc[jb][r + 1] = v[1];
            }
        } else {
            const int r0 = 8 * (u - 2);
            mxa = fmaxf(mxa, fmaxf(Sc[jb][r0], Sc[jb][r0 + 1]));
            mxb = fmaxf(mxb, fmaxf(Sc[jb][r0 + 2], Sc[jb][r0 + 3]));
            mxa = fmaxf(mxa, fmaxf(Sc[jb][r0 + 4], Sc[jb][r0 + 5]));
            mxb = fmaxf(mxb, fmaxf(Sc[jb][r0 + 6], Sc[jb][r0 + 7]));
        }
    };

    // ---- logits of chunk 0 (nothing to overlap with yet) ----
    for_units<2>([&](int jb) __attribute__((always_inline)) {
        kread(0, jb);
        for_units<6>([&](int k) __attribute__((always_inline)) { qk(k); });
        for_units<4>([&](int u) __attribute__((always_inline)) { comb(jb, u); });
    });
    float m_run;
    {
        float mc = fmaxf(mxa, mxb);
        m_run = fmaxf(mc, __shfl_xor(mc, 32, 64));
    }

    // ---- softmax of step i (16 keys: block i >> 1, registers 8 (i & 1) .. + 7) in eight small steps ----
    float pv[8];
    f16x8 php[2], plp[2];   // split probabilities of two consecutive steps
#pragma unroll
    for (int j = 0; j < 8; ++j) { php[1][j] = (_Float16)0.f; plp[1][j] = (_Float16)0.f; vh[1][j] = (_Float16)0.f; vl[1][j] = (_Float16)0.f; }
    float m11 = 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto vs = [&](int i, int u) __attribute__((always_inline)) {
        if (QK_ONLY) return;
        const int jb = i >> 1, r0 = 8 * (i & 1), pb = i & 1;
        if (u == 0) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x2 d = f32x2{Sc[jb][r0 + j], Sc[jb][r0 + j + 1]} - f32x2{m11, m11};
                pv[j] = d[0]; pv[j + 1] = d[1];
            }
        } else if (u == 1 || u == 2) {
#pragma unroll
            for (int j = 4 * (u - 1); j < 4 * u; ++j) pv[j] = __builtin_amdgcn_exp2f(pv[j]);
        } else if (u == 3) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) l2 += f32x2{pv[j], pv[j + 1]};
        } else if (u == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) php[pb][j] = (_Float16)pv[j];
        } else if (FAST) {
        } else if (u == 5 || u == 6) {
            const u32x4 hp = __builtin_bit_cast(u32x4, php[pb]);
#pragma unroll
            for (int j = 2 * (u - 5); j < 2 * (u - 4); ++j) {
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(pv[2 * j]) : "v"(pv[2 * j]), "v"(hp[j]));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(pv[2 * j + 1]) : "v"(pv[2 * j + 1]), "v"(hp[j]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) plp[pb][j] = (_Float16)pv[j];
        }
    };
    // P.V product k of step i: O^T += V^T P^T (p' = hi + lo and v = hi + lo, both residuals unscaled: one accumulator)
    auto pvm = [&](int i, int k) __attribute__((always_inline)) {
        if (QK_ONLY) return;
        const int pb = i & 1;
        if (k == 0) Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[pb], php[pb], Om, 0, 0, 0);
        else if (FAST) {}
        else if (k == 1) Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[pb], php[pb], Om, 0, 0, 0);
        else Om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[pb], plp[pb], Om, 0, 0, 0);
    };

    // maximum over the two halves of a wave on the vector ALU (the clang builtin returns its first result twice)
    auto max_halves = [&](float v) __attribute__((always_inline)) {
        float a_ = v, b_ = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));   // (the compiler knows no hazards of an asm)
        return fmaxf(a_, b_);            // a = (lo, lo), b = (hi, hi)
    };
    // ---- one chunk: softmax of chunk c, P.V of its steps 0-2 (step 3 is issued at the top of the next chunk, where
    //      it covers the latency of the K reads); with NEXT, the logits of chunk c + 1 on the way.  `sc` is the
    //      factor that brings sums and outputs to the running maximum found at the end of the previous chunk. ----
    float sc = 1.0f;
    auto chunk_body = [&](int c, auto next_tag) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(next_tag)::value;
        // chunk c + 1 has landed (its K is read below); the copy of chunk c + 2 may stay in flight
        if (c + 2 < NCH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // every wave is done with chunk c - 1: its slot takes chunk c + 3
        if (c + 3 < NCH) dma_chunk(c + 3);
        if (NEXT) kread(c + 1, 0);
        m11 = m_run - 11.0f;
        mxa = NEG_INF; mxb = NEG_INF;
        l2 *= sc;
        SLOT_END();
        // P.V of the last step of chunk c - 1 (zeros before the first chunk), step 0 beside it
        pvm(3, 0); vs(0, 0); SLOT_END();
        pvm(3, 1); vs(0, 1); SLOT_END();
        pvm(3, 2); vread(c, 0); SLOT_END();
        for_units<6>([&](int k) __attribute__((always_inline)) {
            if (NEXT) qk(k);
            vs(0, 2 + k);
            if (k == 3) vread(c, 1);
            SLOT_END();
        });
        if (!__all(sc == 1.0f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Om[r] *= sc;
        }
        SLOT_END();
        // step 1 beside P.V of step 0, then the logits of block 0 combined (Sc[0] is free after vs(1, 0))
        if (NEXT) kread(c + 1, 1);
        for_units<8>([&](int u) __attribute__((always_inline)) {
            if (u < 3) pvm(0, u);
            vs(1, u);
            if (NEXT && u >= 3 && u < 7) comb(0, u - 3);
            if (u == 5) vread(c, 2);
            SLOT_END();
        });
        // step 2 beside P.V of step 1 and the six products of block 1
        for_units<8>([&](int u) __attribute__((always_inline)) {
            if (NEXT && u < 6) qk(u);
            SLOT_END();
            if (u >= 1 && u < 4) pvm(1, u - 1);
            vs(2, u);
            if (u == 6) vread(c, 3);
            SLOT_END();
        });
        // step 3 beside P.V of step 2, then the logits of block 1 (Sc[1] is free after vs(3, 0))
        for_units<8>([&](int u) __attribute__((always_inline)) {
            if (u < 3) pvm(2, u);
            vs(3, u);
            if (NEXT && u >= 3 && u < 7) comb(1, u - 3);
            SLOT_END();
        });
        if (NEXT) {
            // running maximum; sums and outputs follow it in the next chunk
            float mc = fmaxf(mxa, mxb);
            mc = max_halves(mc);
            // the running maximum follows the chunk maximum only when that is more than 4 octaves above it: a reference up to 4
            // octaves low keeps every p' = 2048 exp2(s - m) below 2^15 (an f16 holds it), costs no precision (the split is
            // relative) and spares the rescale of the 16 output registers and the sums in nearly every chunk after the first
            // (measured: 66.2 -> 65.2-66.2 us per launch - the rescale was not what the chunk waits for)
            const float m_new = mc > m_run + 4.0f ? mc : m_run;
            sc = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
        }
    };
    for (int c = 0; c + 1 < NCH; ++c) chunk_body(c, std::true_type{});
    chunk_body(NCH - 1, std::false_type{});
    for_units<3>([&](int k) __attribute__((always_inline)) { pvm(3, k); });
    if (QK_ONLY) {
        if (hi == 0 && qw + l31 < nq) a.msg[((size_t)b * P + q_off + qw + l31) * 128 + head * 32] = m_run;
        return;
    }

    // ---- message rows through a wave-private tile in two slots nobody touches any more ----
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.0f / l;
    float* T = reinterpret_cast<float*>(smem + ((NCH + 2 - (wave >> 1)) & (NSLOT - 1)) * CHUNK_BYTES) + (wave & 1) * 32 * OROW;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = Om[4 * g4 + j] * inv_l;
        *reinterpret_cast<f32x4_a*>(T + l31 * OROW + 8 * g4 + 4 * hi) = o;
    }
    float* out = a.msg + ((size_t)b * P + q_off) * 128 + head * 32;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), cc = lane & 7;
        const f32x4 v = *reinterpret_cast<const f32x4_a*>(T + row * OROW + cc * 4);
        const int q = qw + row;
        if (q < nq) *reinterpret_cast<f32x4*>(out + (size_t)q * 128 + cc * 4) = v;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Measurement only: the Q K^T phase with NQ = 1 or 2 query sets per wave.  Same chunk ring, K copies
// (global_load_lds), LDS layout, fragment reads and split-f16 MFMA sequence as attention_stream_kernel; per 32-key block a
// wave reads the four K fragments ONCE and feeds them to NQ sets of six products (32 NQ queries per wave, 128 NQ per
// workgroup).  NQ = 2 is the "64 queries per wave" form: half the LDS reads and ring / barrier overhead per product and four
// independent accumulator chains instead of two.  It exists here only in isolation: with the softmax state of the full
// kernel doubled as well (logits 2 x 32, products 2 x 32, O^T 2 x 16, query fragments 2 x 16, ...) a wave needs 270+
// registers, i.e. one wave per SIMD - DESIGN.md section 5, profiles/NOTES_r3.md.  Output: the row maximum of the base-2
// logits in msg[p][head * 32] (so that nothing is dead code), exactly what the QK_ONLY instance of the kernel above writes.
template <int NQ>
__global__ __launch_bounds__(256, 2) void qk_phase_probe_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int i0 = blockIdx.x;
    const int grp = (i0 / (8 * a.QT)) * 8 + (i0 & 7), qt = (i0 >> 3) % a.QT;
    const int head = grp & 3, side = (grp >> 2) & 1, b = grp >> 3;
    const int P = a.N + a.M;
    const int nq = side ? a.M : a.N, q_off = side ? a.N : 0;
    const int src = a.cross ? 1 - side : side;
    const int nk = src ? a.M : a.N, k_off = src ? a.N : 0;
    const int q0 = qt * 128 * NQ;
    if (q0 >= nq) return;
    const int NCH = nk >> 6;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // K copies: waves 0, 1 move the two 32-key blocks of every chunk (the V^T halves of the ring stay unused)
    const int rl8 = lane >> 3, swz = (lane & 7) ^ rl8;
    const char* dbase = reinterpret_cast<const char*>(a.k16 + (((size_t)b * P + k_off + (wave & 1) * 32) * 4 + head) * 64);
    const unsigned voff = rl8 * 512 + swz * 16;
    auto dma_chunk = [&](int ch) __attribute__((always_inline)) {
        if (wave >= 2) return;
        const char* s0 = dbase + (size_t)ch * (64 * 512);
        const unsigned d0 = lds0 + (unsigned)(ch & (NSLOT - 1)) * CHUNK_BYTES + (unsigned)wave * BLK_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(d0 + p * 1024 + ((p >> 1) & 1) * 128), "v"(voff), "s"(s0 + (size_t)p * (8 * 512)) : "memory");
    };
    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    unsigned kofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 2 * i + hi;
        kofs[i] = (krow >> 3) * 1024 + ((krow >> 4) & 1) * 128 + ((krow & 7) * 8 + (c ^ (krow & 7))) * 16;
    }
    const int qw = q0 + wave * 32 * NQ;
    f16x8 qh[NQ][2], ql[NQ][2];
#pragma unroll
    for (int s_ = 0; s_ < NQ; ++s_) {
        const int qrow = min(qw + 32 * s_ + l31, nq - 1);
        const _Float16* p = a.q16 + (((size_t)b * P + q_off + qrow) * 4 + head) * 64 + 8 * hi;
        qh[s_][0] = *reinterpret_cast<const f16x8*>(p);
        qh[s_][1] = *reinterpret_cast<const f16x8*>(p + 16);
        ql[s_][0] = *reinterpret_cast<const f16x8*>(p + 32);
        ql[s_][1] = *reinterpret_cast<const f16x8*>(p + 48);
    }
    dma_chunk(0);
    if (NCH > 1) dma_chunk(1);
    if (NCH > 2) dma_chunk(2);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mx[NQ];
#pragma unroll
    for (int s_ = 0; s_ < NQ; ++s_) mx[s_] = -__builtin_inff();
    for (int c = 0; c < NCH; ++c) {
        // chunk c has landed; the copies of chunks c + 1, c + 2 may stay in flight (4 per chunk in waves 0, 1)
        if (wave < 2) {
            if (c + 2 < NCH) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (c + 1 < NCH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the query loads)
        __syncthreads();                       // every wave is done with chunk c - 1: its slot takes chunk c + 3
        if (c + 3 < NCH) dma_chunk(c + 3);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const char* base = smem + (c & (NSLOT - 1)) * CHUNK_BYTES + jb * BLK_BYTES;
            f16x8 kf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kf[i] = *reinterpret_cast<const f16x8_a*>(base + kofs[i]);
            f32x16 A[NQ], X[NQ];
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_) A[s_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qh[s_][0], zero16, 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_) X[s_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], ql[s_][0], zero16, 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_) A[s_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qh[s_][1], A[s_], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_) X[s_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], ql[s_][1], X[s_], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_) X[s_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2], qh[s_][0], X[s_], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_) X[s_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[3], qh[s_][1], X[s_], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NQ; ++s_)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = f32x2{X[s_][r], X[s_][r + 1]} + f32x2{A[s_][r], A[s_][r + 1]};
                    mx[s_] = fmaxf(mx[s_], fmaxf(v[0], v[1]));
                }
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < NQ; ++s_) {
        const float m = fmaxf(mx[s_], __shfl_xor(mx[s_], 32, 64));
        const int q = qw + 32 * s_ + l31;
        if (hi == 0 && q < nq) a.msg[((size_t)b * P + q_off + q) * 128 + head * 32] = m;
    }
}

}  // namespace

bool attention_stream_supported(int N, int M) { return N % 64 == 0 && M % 64 == 0; }

int launch_attention_stream(int B, int N, int M, int cross, const Qkv16& qkv, float* msg, hipStream_t s, int mode) {
    const int nq_max = N > M ? N : M;
    StreamArgs a{qkv.q16, qkv.k16, qkv.vt16, msg, N, M, qkv.Npad, qkv.PP, cross, (nq_max + 127) / 128};
    const size_t lds = (size_t)NSLOT * CHUNK_BYTES;
    static std::atomic<unsigned long long> optin[2];
    const void* kern = mode == 1 ? reinterpret_cast<const void*>(attention_stream_kernel<true>) : reinterpret_cast<const void*>(attention_stream_kernel<false>);
    if (int rc = mdgat_lds_optin(kern, lds, optin[mode == 1], "attention LDS attribute")) return rc;
    if (mode == 1) hipLaunchKernelGGL(attention_stream_kernel<true>, dim3(B * 8 * a.QT), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(attention_stream_kernel<false>, dim3(B * 8 * a.QT), dim3(256), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "attention launch");
}

int launch_attention_qk_probe(int B, int N, int M, int cross, const Qkv16& qkv, float* msg, hipStream_t s) {
    if (!attention_stream_supported(N, M)) { mdgat_set_error("qk probe: key counts must be multiples of 64"); return MDGAT_ERR_UNSUPPORTED; }
    const int nq_max = N > M ? N : M;
    StreamArgs a{qkv.q16, qkv.k16, qkv.vt16, msg, N, M, qkv.Npad, qkv.PP, cross, (nq_max + 127) / 128};
    const size_t lds = (size_t)NSLOT * CHUNK_BYTES;
    static std::atomic<unsigned long long> optin;
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(attention_stream_kernel<false, true>), lds, optin, "qk probe LDS attribute")) return rc;
    hipLaunchKernelGGL((attention_stream_kernel<false, true>), dim3(B * 8 * a.QT), dim3(256), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "qk probe launch");
}

// nq_sets: 1 (32 queries per wave, as shipped) or 2 (64 queries per wave); see qk_phase_probe_kernel
int launch_qk_phase_probe(int B, int N, int M, int cross, int nq_sets, const Qkv16& qkv, float* msg, hipStream_t s) {
    if (!attention_stream_supported(N, M)) { mdgat_set_error("qk probe: key counts must be multiples of 64"); return MDGAT_ERR_UNSUPPORTED; }
    if (nq_sets != 1 && nq_sets != 2) { mdgat_set_error("qk probe: nq_sets must be 1 or 2"); return MDGAT_ERR_BAD_ARG; }
    const int nq_max = N > M ? N : M;
    const int per_wg = 128 * nq_sets;
    StreamArgs a{qkv.q16, qkv.k16, qkv.vt16, msg, N, M, qkv.Npad, qkv.PP, cross, (nq_max + per_wg - 1) / per_wg};
    const size_t lds = (size_t)NSLOT * CHUNK_BYTES;
    static std::atomic<unsigned long long> optin[2];
    const void* kern = nq_sets == 2 ? reinterpret_cast<const void*>(qk_phase_probe_kernel<2>) : reinterpret_cast<const void*>(qk_phase_probe_kernel<1>);
    if (int rc = mdgat_lds_optin(kern, lds, optin[nq_sets - 1], "qk phase probe LDS attribute")) return rc;
    if (nq_sets == 2) hipLaunchKernelGGL(qk_phase_probe_kernel<2>, dim3(B * 8 * a.QT), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(qk_phase_probe_kernel<1>, dim3(B * 8 * a.QT), dim3(256), lds, s, a);
    return mdgat_check_hip(hipGetLastError(), "qk phase probe launch");
}
