// log_optimal_transport + log_sinkhorn_iterations (mdgat.py:279-308) in fp64 - the reference's own arithmetic - for the
// reference-exact mode's arg-max.  Why: with the encoders and the layers in fp64 (f64.hip, layer_f64.hip) what is left of the
// exact mode's Z error is its fp32-class tail: 1.35e-6 rms, 7e-6 max (profiles/NOTES_r6.md section 11).  That is far inside the
// bar on Z (1e-4) but not below the gap of every arg-max: among 40 960 arg-maxes of a reference-held batch of 40 pairs one had its
// two candidates 1.3e-6 apart in the reference's fp64 Z, and fell the other way.  With the scores and the transport in fp64 the
// arg-maxes are the reference's down to gaps of 1e-12.
//
// The iteration, in the scaling form (as sinkhorn.hip runs it in fp32; the same map as the log-domain one the reference writes):
// with Z0 the couplings (scores, bin score on the border), r_i = max_j Z0_ij and K_ij = exp(Z0_ij - r_i),
//      a_i = mu_i / sum_j K_ij b_j,     b_j = nu_j / sum_i K_ij a_i      (b = 1 at start: v = 0, mdgat.py:281-284 updates u first)
//      Z_ij = Z0_ij - r_i + log a_i + log b_j - norm,      mu, nu, norm as mdgat.py:300-307.
// fp64 has the range for it: every K is in (e^-700, 1], the scaling factors stay within e^(+-|scores|).
//
// gfx950 mapping.  A pair's N x (M + 1) block of K stays in REGISTERS for all iterations, spread over G = ceil(N / 32) workgroups of
// eight waves: workgroup g owns rows 32 g .. 32 g + 31, wave w of it four of them, lane l the columns
// l, l + 64, ... (nine per lane at M = 512: 36 doubles); the dustbin ROW is all ones after its maximum is taken out and is carried as
// one scalar.  Row sums are in-lane dot products and one butterfly per row; column sums are in-lane over
// the wave's four rows, merged over the eight waves through LDS and over the G workgroups through memory: every workgroup publishes
// its M + 1 partials (write-through stores, acknowledged before the workgroup's flag goes out), waits for its partners' flags and adds
// the G partials of a column in slab order - the same bits in every workgroup.  Two slot sets alternate by iteration parity (a workgroup can only be two
// iterations ahead of a partner it has not heard from).  Spins are bounded: a timeout raises the error word and the call fails.
// The partners of a pair sit on one XCD under round-robin dispatch (blockIdx % 8), which only matters for speed.
//
// Frames beyond 575 keypoints (to 2175) run the STREAMING form further down: K written to memory once, one launch per iteration.
#include "common.hpp"
#include "f64.hpp"
#include "sinkhorn_f64.hpp"
#include "coop_chain.hpp"

namespace {

constexpr int S64_NC = 9;                                            // nine columns per lane: M + 1 <= 576
constexpr int S64_SLOT = 576;                                        // doubles per published vector (>= M + 1, 64-aligned)
constexpr int S64_GMAX = 18;                                         // row slabs of a pair at most (eight-wave workgroups): N <= 576

struct Sk64Args {
    const double* scores;      // [B][N][M]
    double alpha;              // the bin score (mdgat.py:359-360) ...
    const double* alpha_dev;   // ... or where it lives on the device (the forward: the weight blob), if not null
    int B, N, M, iters, G, inner;
    double* Z64;               // optional [B][N + 1][M + 1]
    float* Z32;                // optional, the fp32 rounding of the same
    int* rbest_idx; float* rbest_val;      // optional [B][N]: arg-max of every row (over the columns the extraction mode scans)
    int* cslab_idx; double* cslab_val;     // optional [B][G][M]: arg-max of every column over this workgroup's rows, value in fp64 (merged by
                                           // sinkhorn_f64_merge_kernel: a merge on fp32 roundings would undo the fp64 decision between slabs)
    double* slots;             // [B][2][G][S64_SLOT]
    unsigned* flags;           // [B][3][G], zeroed per launch: two sets of iteration flags, then the partners' XCC ids + 1
    unsigned* error_word;      // bit 0: a workgroup gave up waiting for a partner
};

// sum over the 64 lanes, the same bits in every lane: four rotations inside the DPP rows (no LDS crossbar), then the two steps across
// rows (ds_bpermute).  (Six ds_bpermute steps of two 32-bit halves each: a dependent chain of 12 crossbar round trips per sum, five
// sums per iteration.)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_d<0x128>(v);      // row_ror:8
    v += dpp_mov_d<0x124>(v);      // row_ror:4
    v += dpp_mov_d<0x122>(v);      // row_ror:2
    v += dpp_mov_d<0x121>(v);      // row_ror:1   (every lane of a row holds the row's sum)
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m, 64);
        v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    return v;
}
// Four wave-wide sums at once (the scheme of sinkhorn.hip's wave_sum16, in fp64): in: this lane's partials p[0..3] of the wave's four
// rows; out: in EVERY lane of the 16-lane DPP row q the wave total of row q.  Lane-half swap (v_permlane32_swap on two copies: lanes
// below 32 then hold rows 0 / 1, the others rows 2 / 3), 16-lane-row swap, four rotations inside the rows: ~30 instructions and one
// dependency chain instead of four chains of ~24 (wave_sum_f64 per row).  (Inline asm: the compiler inserts none of the wait states
// these instructions need around a vector write / read of their operands - sinkhorn.hip.)
__device__ __forceinline__ void swap_halves32_d(double& a, double& b) {
    unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    unsigned al = (unsigned)ua, ah = (unsigned)(ua >> 32), bl_ = (unsigned)ub, bh = (unsigned)(ub >> 32);
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(al), "+v"(bl_), "+v"(ah), "+v"(bh));
    a = __builtin_bit_cast(double, ((unsigned long long)ah << 32) | al);
    b = __builtin_bit_cast(double, ((unsigned long long)bh << 32) | bl_);
}
__device__ __forceinline__ void swap_rows16_d(double& a, double& b) {
    unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    unsigned al = (unsigned)ua, ah = (unsigned)(ua >> 32), bl_ = (unsigned)ub, bh = (unsigned)(ub >> 32);
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(al), "+v"(bl_), "+v"(ah), "+v"(bh));
    a = __builtin_bit_cast(double, ((unsigned long long)ah << 32) | al);
    b = __builtin_bit_cast(double, ((unsigned long long)bh << 32) | bl_);
}
__device__ __forceinline__ double wave_sum4_f64(const double (&p)[4]) {
    double x0 = p[0], y0 = p[2], x1 = p[1], y1 = p[3];
    swap_halves32_d(x0, y0);                 // x0 = (p0 | p2 from the lower half), y0 = (p0 | p2 from the upper half)
    swap_halves32_d(x1, y1);
    double s0 = x0 + y0, s1 = x1 + y1;       // lanes < 32: rows 0 / 1 summed over the halves; lanes >= 32: rows 2 / 3
    swap_rows16_d(s0, s1);                   // s0 = (s0 r0, s1 r0, s0 r2, s1 r2), s1 = (s0 r1, s1 r1, s0 r3, s1 r3)
    double t = s0 + s1;                      // DPP row q: row q's partials summed over the four 16-lane rows
    t += dpp_mov_d<0x128>(t);
    t += dpp_mov_d<0x124>(t);
    t += dpp_mov_d<0x122>(t);
    t += dpp_mov_d<0x121>(t);
    return t;
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// 1 / p for the scaling factors (p a positive sum, far from the ends of the exponent range): the hardware's estimate and two Newton
// steps - 2^-53 relative, not correctly rounded, the same bits wherever it is evaluated.  The IEEE division is ~35 instructions, and an
// iteration holds six of them per lane: nearly half of its vector instructions.
__device__ __forceinline__ double recip_f64(double p) {
    double r = __builtin_amdgcn_rcp(p);
    r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m, 64);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// S64_WAVES waves of four rows each per workgroup: 8 (32 rows) is what is launched - a pair of 512 keypoints is 16 workgroups on 16 CUs.
template <int S64_WAVES>
__global__ __launch_bounds__(64 * S64_WAVES, S64_WAVES / 4) void sinkhorn_f64_kernel(Sk64Args a) {
    // (one workgroup per CU either way: eight waves at 168 registers, sixteen at 128.  Two eight-wave workgroups per CU - 128 registers,
    // 152 bytes of scratch - are no faster: 32 pairs 917 against 878 us, one pair 502 against 432: a CU's iteration time is its
    // instruction count - ~450 per wave, of which the 72 products are a sixth; the rest is the five wave reductions, the exchange and
    // the reciprocals)
    constexpr int S64_THREADS = 64 * S64_WAVES, S64_ROWS = 4 * S64_WAVES;
    extern __shared__ __attribute__((aligned(16))) double s64_lds[];
    double (*colbuf)[S64_SLOT] = reinterpret_cast<double (*)[S64_SLOT]>(s64_lds);       // [waves]: column partials; in the epilogue the values of the column arg-max
    double* bl = s64_lds + (size_t)S64_WAVES * S64_SLOT;                                // the new b, for every lane to pick its columns from; in the epilogue the dustbin row of Z
    int (*cidx)[S64_SLOT] = reinterpret_cast<int (*)[S64_SLOT]>(bl + S64_SLOT);         // [waves]: row indices of the column arg-max (epilogue)
    __shared__ int dead, same_xcd_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, M = a.M, G = a.G;
    // blockIdx = (pair / 8) * (8 G) + g * 8 + pair % 8: the G workgroups of a pair share blockIdx % 8
    const int grp = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
    const int pair = grp * 8 + (rem & 7), g = rem >> 3;
    if (pair >= a.B) return;
    if (tid == 0) dead = 0;
    const double* sc = a.scores + (size_t)pair * N * M;
    const int row0 = g * S64_ROWS + wave * 4;            // this wave's rows row0 .. row0 + 3 (real rows: < N)
    const double nm = (double)(N + M);
    const double norm = -log(nm);
    const bool last = g == G - 1;                        // the slab that also carries the dustbin row
    const double alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;
    auto z0 = [&](int i, int j) -> double { return j < M ? sc[(size_t)i * M + j] : alpha; };      // (real rows only)

    // K = exp(Z0 - row maximum): 4 x 9 doubles per lane.  The DUSTBIN ROW needs no storage: its couplings are the bin score in every
    // column (mdgat.py:296-298), so K = 1 throughout - its row sum is the sum of b, its share of every column sum is a_N itself.
    double K[4][S64_NC], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + i;
        double m = -__builtin_inf();
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) {
            const int j = lane + 64 * c;
            K[i][c] = (row < N && j <= M) ? z0(row, j) : -__builtin_inf();
            m = fmax(m, K[i][c]);
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) m = fmax(m, shfl_xor_d(m, s));
        r[i] = m;
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) K[i][c] = (row < N && lane + 64 * c <= M) ? exp(K[i][c] - m) : 0.0;
    }
    const double mu = 1.0 / nm, muN = (double)M / nm, nuM = (double)N / nm;      // (mu, nu of mdgat.py:300-303; real columns: nu = mu)
    double b[S64_NC], av[4], aN = 0.0;
#pragma unroll
    for (int c = 0; c < S64_NC; ++c) b[c] = lane + 64 * c <= M ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = 0.0;
    double* slots = a.slots + (size_t)pair * 2 * G * S64_SLOT;
    unsigned* flags = a.flags + (size_t)pair * 3 * G;
    // Do the G workgroups of this pair share an XCD (they do under round-robin dispatch: blockIdx % 8)?  Then they share an L2: PLAIN
    // stores (the CU's L1 writes through to the XCD's L2 and the line stays there) and L1-bypassing loads carry the exchange inside the
    // L2.  Agent-scope write-through stores - right wherever the partners run - send every partial on to memory.  Checked once per
    // launch with an agent-scope exchange.
    if (tid == 0) {
        const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;      // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(flags + 2 * G + g, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int same = 1;
        for (int q = 0; q < G && same >= 0; ++q) {
            unsigned v = 0;
            long spins = 0;
            while ((v = __hip_atomic_load(flags + 2 * G + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u)
                if (++spins > (1L << 22)) { dead = 1; same = -1; break; }
            if (same > 0 && v != my_xcc + 1u) same = 0;
        }
        same_xcd_s = same > 0;
    }
    __syncthreads();
    const bool same_xcd = same_xcd_s != 0;

    for (int it = 0; it < a.iters; ++it) {
        // a_i = mu_i / sum_j K_ij b_j; the dustbin row: a_N = mu_N / sum_j b_j (the same bits in every wave of every workgroup)
        {                    // (only the last slab needs it, but every slab waits for the last one anyway: under `if (last)` 392 against 399 us for
            double sb = 0.0;     // one pair, 823-847 against 804-827 for 32 - nothing; and that branch once broke a since-dropped instantiation)
#pragma unroll
            for (int c = 0; c < S64_NC; ++c) sb += b[c];
            aN = muN * recip_f64(wave_sum_f64(sb));
        }
        {
            double p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[i] = 0.0;
#pragma unroll
                for (int c = 0; c < S64_NC; ++c) p[i] = __builtin_fma(K[i][c], b[c], p[i]);
            }
            // the four row sums in one reduction: DPP row q ends up with row q's total, ONE reciprocal serves the four rows, and the
            // scalings go back to every lane through scalar registers
            const double tq = wave_sum4_f64(p);
            const double aq = row0 + (lane >> 4) < N ? mu * recip_f64(tq) : 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = readlane_d(aq, 16 * i);
        }
        // column partials of this wave's rows -> LDS
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) q = __builtin_fma(K[i][c], av[i], q);
            colbuf[wave][lane + 64 * c] = q;
        }
        __syncthreads();
        // The exchange.  Stores (plain inside an XCD: they stay in its L2; write-through otherwise), acknowledged before the workgroup
        // meets and its flag goes out; the partners' flags polled by G threads; then every thread loads the G partials of its column,
        // all in flight at once (relaxed agent-scope loads: they pass the non-coherent L1).  Measured against it: release / acquire
        // FENCES by every thread (a write-back / invalidation of the L2 each): 34 us per iteration for one pair instead of 5.3; the G
        // loads in a run-time loop (a round trip each): 11.8; the data as its own flag (a tag in the lowest mantissa bit, every thread
        // polling its G granules): 7.1 - the polling rounds of 512 threads x 16 granules cost more than the two round trips they save.
        double* mine = slots + ((size_t)(it & 1) * G + g) * S64_SLOT;
        for (int j = tid; j <= M; j += S64_THREADS) {
            double p = colbuf[0][j];
#pragma unroll
            for (int w = 1; w < S64_WAVES; ++w) p += colbuf[w][j];
            if (last) p += aN;                              // the dustbin row's share of the column (K = 1), by one slab
            if (same_xcd) mine[j] = p;
            else __hip_atomic_store(mine + j, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* fl = flags + (size_t)(it & 1) * G;
        if (tid == 0) __hip_atomic_store(fl + g, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < G && tid != g && !dead) {
            long spins = 0;
            while (__hip_atomic_load(fl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) {
                if (++spins > (1L << 22)) { dead = 1; break; }
            }
        }
        __syncthreads();
        const double* all = slots + (size_t)(it & 1) * G * S64_SLOT;
        for (int j = tid; j <= M; j += S64_THREADS) {
            constexpr int GM = S64_GMAX * 8 / S64_WAVES;     // (64-row slabs: at most nine)
            double v[GM];
#pragma unroll
            for (int q = 0; q < GM; ++q) v[q] = q < G ? __hip_atomic_load(all + (size_t)q * S64_SLOT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            double tot = v[0];
#pragma unroll
            for (int q = 1; q < GM; ++q) tot += v[q];        // slab order: the same bits in every workgroup (+ 0.0 beyond G)
            bl[j] = (j < M ? mu : nuM) * recip_f64(tot);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) b[c] = lane + 64 * c <= M ? bl[lane + 64 * c] : 0.0;
    }
    if (dead && tid == 0 && a.error_word) __hip_atomic_store(a.error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.iters == 0) {         // u = v = 0 (mdgat.py:281): a = e^r in the shifted form, the dustbin row's e^alpha
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = exp(r[i]);
        aN = exp(alpha);
    }

    // Z = Z0 - r + log a + log b - norm; arg-maxes decided on the fp64 values
    double lb[S64_NC];
#pragma unroll
    for (int c = 0; c < S64_NC; ++c) lb[c] = lane + 64 * c <= M ? log(b[c]) - norm : 0.0;
    const int jlim = a.inner ? M : M + 1;                  // columns a row's arg-max scans
    double cb[S64_NC];
    int ci[S64_NC];
#pragma unroll
    for (int c = 0; c < S64_NC; ++c) { cb[c] = -__builtin_inf(); ci[c] = 0x7fffffff; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + i;
        if (row >= N) continue;                             // (wave-uniform)
        const double la = log(av[i]) - r[i];
        double best = -__builtin_inf();
        int bj = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) {
            const int j = lane + 64 * c;
            if (j > M) continue;
            const double z = z0(row, j) + la + lb[c];
            if (a.Z64) a.Z64[((size_t)pair * (N + 1) + row) * (M + 1) + j] = z;
            if (a.Z32) a.Z32[((size_t)pair * (N + 1) + row) * (M + 1) + j] = (float)z;
            if (j < jlim && z > best) { best = z; bj = j; }            // ascending j within the lane: the first maximum stays
            if (z > cb[c]) { cb[c] = z; ci[c] = row; }                 // ascending rows within the wave
        }
        if (a.rbest_idx) {
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const double ob = shfl_xor_d(best, s);
                const int oj = __shfl_xor(bj, s, 64);
                if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }      // torch.max: the first of equal maxima
            }
            if (lane == 0) { a.rbest_idx[(size_t)pair * N + row] = bj; a.rbest_val[(size_t)pair * N + row] = (float)best; }
        }
    }
    // the dustbin row of Z (row N: Z0 - r = 0), by the last slab's first wave
    __syncthreads();
    if (last && wave == 0) {
        const double laN = log(aN);
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) {
            const int j = lane + 64 * c;
            if (j > M) continue;
            const double z = laN + lb[c];
            bl[j] = z;
            if (a.Z64) a.Z64[((size_t)pair * (N + 1) + N) * (M + 1) + j] = z;
            if (a.Z32) a.Z32[((size_t)pair * (N + 1) + N) * (M + 1) + j] = (float)z;
        }
    }
    if (a.cslab_idx) {
        // the column arg-max over this workgroup's rows: waves in ascending row order, strict compare (the first of equal maxima); the
        // dustbin row - the last of all - joins unless the extraction scans the inner block only
#pragma unroll
        for (int c = 0; c < S64_NC; ++c) { colbuf[wave][lane + 64 * c] = cb[c]; cidx[wave][lane + 64 * c] = ci[c]; }
        __syncthreads();
        for (int j = tid; j < M; j += S64_THREADS) {
            double bv = colbuf[0][j];
            int bi = cidx[0][j];
#pragma unroll
            for (int w = 1; w < S64_WAVES; ++w)
                if (colbuf[w][j] > bv) { bv = colbuf[w][j]; bi = cidx[w][j]; }
            if (last && !a.inner && bl[j] > bv) { bv = bl[j]; bi = N; }
            a.cslab_val[((size_t)pair * G + g) * M + j] = bv;
            a.cslab_idx[((size_t)pair * G + g) * M + j] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

// column arg-max over the G row slabs of a pair, in fp64: ascending slabs, strict compare (the first of equal maxima)
__global__ __launch_bounds__(256) void sinkhorn_f64_merge_kernel(const int* sidx, const double* sval, int B, int G, int M, int* cbest_idx, float* cbest_val) {
    const size_t total = (size_t)B * M;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t pair = e / M, j = e % M;
        const size_t base = pair * G * M + j;
        double bv = sval[base];
        int bi = sidx[base];
        for (int g = 1; g < G; ++g)
            if (sval[base + (size_t)g * M] > bv) { bv = sval[base + (size_t)g * M]; bi = sidx[base + (size_t)g * M]; }
        cbest_idx[e] = bi;
        cbest_val[e] = (float)bv;
    }
}


// ---- the STREAMING form: frames beyond what the register-resident kernel holds (M + 1 > 576 or N > 576), up to 2175 keypoints ----
// The same iteration, K = exp(Z0 - row maximum) written to memory ONCE (fp64, rows padded to 128 columns: 35.6 MB for a pair of 2048
// keypoints) and streamed once per iteration - one launch per iteration, no workgroup waits for another inside a launch:
//   sinkhorn_f64_wide_b_kernel:    b_j = nu_j / sum over the slabs, in slab order, of the column partials the launch before left
//                                  (b = 1 at the start) - one thread per column, the G partials in flight at once;
//   sinkhorn_f64_wide_iter_kernel, workgroup (pair, slab g of 32 rows):   b -> LDS;  a_N = mu_N / sum_j b_j;  every wave takes a row at a
//   time (four of the slab's rows each, the next row's loads in flight while this one is reduced): the row's K into registers with
//   16-byte loads, the dot product with b, one wave reduction, a_i = mu / that, and K_ij a_i joins the wave's column partials in
//   registers;  the eight waves' partials are summed through LDS and leave as the slab's partials (the last slab adds the dustbin
//   row's share a_N).
// Per iteration a pair moves its K once (8 pairs of 2048: 285 MB in 50 us - 5.7 TB/s, HBM-bound; one pair: 14.5 us) and the partials
// twice (17 KB per slab).  The partial sets alternate by iteration parity.  The last launch forms Z, the arg-max of every row and per
// slab of every column.  (First form measured: slabs of 64 rows, every workgroup summing the partials itself in its prologue - one
// launch per iteration, but 0.56 MB of partials per workgroup against 1.1 MB of K: 62 / 38 us per iteration for 8 pairs / one pair
// against 55 / 19 now.)
constexpr int W64_ROWS = 32;                                          // rows per slab (four per wave)
constexpr int W64_GMAX = 68;                                          // slabs at most: N <= 2176
constexpr int W64_NC2MAX = 17;                                        // column pairs per lane at most: M + 1 <= 2176

struct Wk64Args {
    const double* scores; double alpha; const double* alpha_dev;
    int B, N, M, Mp, G, it, iters, inner;
    double* K;                 // [B][N][Mp]
    double* rmax;              // [B][N]
    double* av;                // [B][N + 1]: the row scalings of the latest iteration, a_N last
    double* P;                 // [B][2][G][Mp]: column partials per slab, by iteration parity
    double* bvec;              // [B][Mp]: the column scalings the coming launch works with (sinkhorn_f64_wide_b_kernel)
    double* Z64; float* Z32; int* rbest_idx; float* rbest_val; int* cslab_idx; double* cslab_val;
};

typedef double w64x2 __attribute__((ext_vector_type(2)));

// one wave per row: the row maximum (over the scores and the bin score of the dustbin column), K = exp(Z0 - it), zero in the padding
__global__ __launch_bounds__(512) void sinkhorn_f64_wide_init_kernel(Wk64Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blocks = (a.N + 7) / 8;
    const int pair = blockIdx.x / blocks, row = (blockIdx.x % blocks) * 8 + wave;
    if (row >= a.N) return;
    const double alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;
    const double* sc = a.scores + ((size_t)pair * a.N + row) * a.M;
    double m = alpha;
    for (int j = lane; j < a.M; j += 64) m = fmax(m, sc[j]);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmax(m, shfl_xor_d(m, s));
    double* k = a.K + ((size_t)pair * a.N + row) * a.Mp;
    for (int j = lane; j < a.Mp; j += 64) k[j] = j < a.M ? exp(sc[j] - m) : j == a.M ? exp(alpha - m) : 0.0;
    if (lane == 0) a.rmax[(size_t)pair * a.N + row] = m;
}

// b_j = nu_j / (sum over the slabs, in slab order, of the column partials launch `it` - 1 left) for launch `it` (it = iters: for the last
// launch); b = 1 before the first iteration, 0 in the padding.  One thread per column, the G partials in flight at once.
__global__ __launch_bounds__(128) void sinkhorn_f64_wide_b_kernel(Wk64Args a) {
    const int M = a.M, Mp = a.Mp, G = a.G;
    const int per = Mp >> 7;
    const int pair = blockIdx.x / per, j = (blockIdx.x % per) * 128 + threadIdx.x;
    const double nm = (double)(a.N + M);
    const double mu = 1.0 / nm, nuM = (double)a.N / nm;
    double bv = 0.0;
    if (j <= M) {
        if (a.it == 0) bv = 1.0;
        else {
            const double* Pp = a.P + (((size_t)pair * 2 + ((a.it + 1) & 1)) * G) * Mp + j;      // what launch it - 1 wrote
            double v[W64_GMAX];
#pragma unroll
            for (int q = 0; q < W64_GMAX; ++q) v[q] = q < G ? Pp[(size_t)q * Mp] : 0.0;
            double tot = v[0];
#pragma unroll
            for (int q = 1; q < W64_GMAX; ++q) tot += v[q];                                       // (+ 0.0 beyond G)
            bv = (j < M ? mu : nuM) * recip_f64(tot);
        }
    }
    a.bvec[(size_t)pair * Mp + j] = bv;
}

template <int NC2>
__global__ __launch_bounds__(512) void sinkhorn_f64_wide_iter_kernel(Wk64Args a) {
    extern __shared__ __attribute__((aligned(16))) double w64_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, M = a.M, Mp = a.Mp, G = a.G;
    const int nc2 = Mp >> 7;
    double* bl = w64_lds;                          // [Mp]
    double* colbuf = w64_lds + Mp;                 // [8][Mp]
    const int pair = blockIdx.x / G, g = blockIdx.x % G;
    const double nm = (double)(N + M);
    const double mu = 1.0 / nm, muN = (double)M / nm;
    for (int j = 2 * tid; j < Mp; j += 1024) *reinterpret_cast<w64x2*>(bl + j) = *reinterpret_cast<const w64x2*>(a.bvec + (size_t)pair * Mp + j);
    double acc[2 * NC2];
#pragma unroll
    for (int c = 0; c < 2 * NC2; ++c) acc[c] = 0.0;
    const int rbeg = g * W64_ROWS, rend = rbeg + W64_ROWS < N ? rbeg + W64_ROWS : N;
    const double* kbase = a.K + (size_t)pair * N * Mp + 2 * lane;
    auto fetch = [&](w64x2 (&kv)[NC2], int row) {
        const double* k = kbase + (size_t)row * Mp;
#pragma unroll
        for (int c = 0; c < NC2; ++c) kv[c] = (c < nc2 && row < rend) ? *reinterpret_cast<const w64x2*>(k + 128 * c) : w64x2{0.0, 0.0};
    };
    auto work = [&](const w64x2 (&kv)[NC2], int row) {
        double p = 0.0;
#pragma unroll
        for (int c = 0; c < NC2; ++c)
            if (c < nc2) {
                const w64x2 bb = *reinterpret_cast<const w64x2*>(bl + 128 * c + 2 * lane);
                p = __builtin_fma(kv[c].x, bb.x, p);
                p = __builtin_fma(kv[c].y, bb.y, p);
            }
        const double ai = mu * recip_f64(wave_sum_f64(p));
        if (lane == 0) a.av[(size_t)pair * (N + 1) + row] = ai;
#pragma unroll
        for (int c = 0; c < NC2; ++c) {
            acc[2 * c] = __builtin_fma(kv[c].x, ai, acc[2 * c]);
            acc[2 * c + 1] = __builtin_fma(kv[c].y, ai, acc[2 * c + 1]);
        }
    };
    // a row's K is in flight while the row before it is reduced (two register sets, the loop unrolled by two)
    w64x2 ka[NC2], kb[NC2];
    fetch(ka, rbeg + wave);
    __syncthreads();
    double sb = 0.0;
    for (int j = lane; j <= M; j += 64) sb += bl[j];
    const double aN = muN * recip_f64(wave_sum_f64(sb));      // (the same bits in every wave of every workgroup)
    for (int row = rbeg + wave; row < rend; row += 16) {
        fetch(kb, row + 8);
        work(ka, row);
        if (row + 8 < rend) {
            fetch(ka, row + 16);
            work(kb, row + 8);
        }
    }
#pragma unroll
    for (int c = 0; c < NC2; ++c)
        if (c < nc2) *reinterpret_cast<w64x2*>(colbuf + (size_t)wave * Mp + 128 * c + 2 * lane) = w64x2{acc[2 * c], acc[2 * c + 1]};
    __syncthreads();
    const bool last = g == G - 1;
    double* mine = a.P + (((size_t)pair * 2 + (a.it & 1)) * G + g) * Mp;
    for (int j = tid; j < Mp; j += 512) {
        double p = colbuf[j];
#pragma unroll
        for (int w = 1; w < 8; ++w) p += colbuf[(size_t)w * Mp + j];
        if (last) p += aN;                                      // the dustbin row's share (K = 1)
        mine[j] = p;
    }
    if (last && tid == 0) a.av[(size_t)pair * (N + 1) + N] = aN;
}

// Z = Z0 - r + log a + log b - norm and the arg-maxes, decided on the fp64 values (as the register-resident kernel's epilogue)
template <int NC2>
__global__ __launch_bounds__(512) void sinkhorn_f64_wide_final_kernel(Wk64Args a) {
    extern __shared__ __attribute__((aligned(16))) double w64_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, M = a.M, Mp = a.Mp, G = a.G;
    const int nc2 = Mp >> 7;
    double* bl = w64_lds;                                        // [Mp]: log b - norm
    double* cbv = w64_lds + Mp;                                  // [Mp]: column arg-max over the slab, value ...
    int* cbi = reinterpret_cast<int*>(cbv + Mp);                 // ... and row
    double* zN = reinterpret_cast<double*>(cbi + Mp);            // [Mp]: the dustbin row of Z (last slab)
    const int pair = blockIdx.x / G, g = blockIdx.x % G;
    const double norm = -log((double)(N + M));
    const double alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;
    const double* sc = a.scores + (size_t)pair * N * M;
    for (int j = tid; j < Mp; j += 512) {
        bl[j] = j <= M ? log(a.bvec[(size_t)pair * Mp + j]) - norm : 0.0;
        cbv[j] = -__builtin_inf();
        cbi[j] = 0x7fffffff;
    }
    __syncthreads();
    const int jlim = a.inner ? M : M + 1;
    double cb[2 * NC2];
    int ci[2 * NC2];
#pragma unroll
    for (int c = 0; c < 2 * NC2; ++c) { cb[c] = -__builtin_inf(); ci[c] = 0x7fffffff; }
    const int rbeg = g * W64_ROWS, rend = rbeg + W64_ROWS < N ? rbeg + W64_ROWS : N;
    for (int row = rbeg + wave; row < rend; row += 8) {
        const double la = a.iters ? log(a.av[(size_t)pair * (N + 1) + row]) - a.rmax[(size_t)pair * N + row] : 0.0;      // (no iteration: u = 0)
        double best = -__builtin_inf();
        int bj = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < 2 * NC2; ++c) {
            const int j = 128 * (c >> 1) + 2 * lane + (c & 1);
            if ((c >> 1) >= nc2 || j > M) continue;
            const double z = (j < M ? sc[(size_t)row * M + j] : alpha) + la + bl[j];
            if (a.Z64) a.Z64[((size_t)pair * (N + 1) + row) * (M + 1) + j] = z;
            if (a.Z32) a.Z32[((size_t)pair * (N + 1) + row) * (M + 1) + j] = (float)z;
            if (j < jlim && z > best) { best = z; bj = j; }            // ascending j within the lane
            if (z > cb[c]) { cb[c] = z; ci[c] = row; }                 // ascending rows within the wave
        }
        if (a.rbest_idx) {
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const double ob = shfl_xor_d(best, s);
                const int oj = __shfl_xor(bj, s, 64);
                if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
            }
            if (lane == 0) { a.rbest_idx[(size_t)pair * N + row] = bj; a.rbest_val[(size_t)pair * N + row] = (float)best; }
        }
    }
    const bool last = g == G - 1;
    if (last && wave == 0) {
        const double laN = a.iters ? log(a.av[(size_t)pair * (N + 1) + N]) : alpha;      // (no iteration: the dustbin row is the bin score)
        for (int j = lane; j <= M; j += 64) {
            const double z = laN + bl[j];
            zN[j] = z;
            if (a.Z64) a.Z64[((size_t)pair * (N + 1) + N) * (M + 1) + j] = z;
            if (a.Z32) a.Z32[((size_t)pair * (N + 1) + N) * (M + 1) + j] = (float)z;
        }
    }
    if (a.cslab_idx) {
        // the waves' rows interleave (row = slab start + wave + 8 t): the first of equal maxima is the one with the smaller row
        for (int w = 0; w < 8; ++w) {
            if (wave == w) {
#pragma unroll
                for (int c = 0; c < 2 * NC2; ++c) {
                    const int j = 128 * (c >> 1) + 2 * lane + (c & 1);
                    if ((c >> 1) >= nc2 || j >= M) continue;
                    if (cb[c] > cbv[j] || (cb[c] == cbv[j] && ci[c] < cbi[j])) { cbv[j] = cb[c]; cbi[j] = ci[c]; }
                }
            }
            __syncthreads();
        }
        for (int j = tid; j < M; j += 512) {
            double bv = cbv[j];
            int bi = cbi[j];
            if (last && !a.inner && zN[j] > bv) { bv = zN[j]; bi = N; }
            a.cslab_val[((size_t)pair * G + g) * M + j] = bv;
            a.cslab_idx[((size_t)pair * G + g) * M + j] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

}  // namespace

static size_t s64_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t s64_lds_bytes(int waves) { return ((size_t)waves * S64_SLOT + S64_SLOT) * sizeof(double) + (size_t)waves * S64_SLOT * sizeof(int); }

static bool s64_resident_supported(int N, int M) { return N >= 1 && M >= 1 && M + 1 <= 64 * S64_NC && N <= 32 * S64_GMAX; }
static bool s64_wide_supported(int N, int M) { return N >= 1 && M >= 1 && M + 1 <= 128 * W64_NC2MAX && N + 1 <= 128 * W64_NC2MAX; }

// -1 (default): the register-resident kernel wherever it holds the shape, the streaming form beyond; 1: always the streaming form (tests
// and measurements: the two agree to rounding, not bit for bit - other summation orders).  MDGAT_F64_SINKHORN_FORM in the environment.
static std::atomic<int> g_s64_form{-2};
static int s64_form() {
    const int v = g_s64_form.load(std::memory_order_relaxed);
    if (v != -2) return v;
    static const int env = [] { const char* e = getenv("MDGAT_F64_SINKHORN_FORM"); return e && atoi(e) == 1 ? 1 : -1; }();
    return env;
}
extern "C" int mdgat_set_f64_sinkhorn_form(int mode) {
    const int prev = s64_form();
    g_s64_form.store(mode == 1 ? 1 : mode == -1 ? -1 : -2, std::memory_order_relaxed);
    return prev;
}
static bool s64_use_wide(int N, int M) { return !s64_resident_supported(N, M) || (s64_form() == 1 && s64_wide_supported(N, M)); }

static size_t s64_wide_bytes(int B, int N, int M) {
    const size_t Mp = ((size_t)M + 1 + 127) & ~(size_t)127, G = ((size_t)N + W64_ROWS - 1) / W64_ROWS;
    return s64_align((size_t)B * N * Mp * sizeof(double)) + s64_align((size_t)B * N * sizeof(double)) + s64_align((size_t)B * (N + 1) * sizeof(double)) +
           s64_align((size_t)B * 2 * G * Mp * sizeof(double)) + s64_align((size_t)B * Mp * sizeof(double)) + s64_align((size_t)B * G * M * sizeof(double)) + s64_align((size_t)B * G * M * sizeof(int));
}

// (the register-resident form sized for the eight-wave workgroups: the larger slab count)
size_t sinkhorn_f64_workspace_bytes(int B, int N, int M) {
    if (B <= 0) return 0;
    if (s64_use_wide(N, M)) return s64_wide_bytes(B, N, M);
    const size_t G = (N + 31) / 32;
    return s64_align((size_t)B * 2 * G * S64_SLOT * sizeof(double)) + s64_align((size_t)B * 3 * G * sizeof(unsigned)) +
           s64_align((size_t)B * G * M * sizeof(double)) + s64_align((size_t)B * G * M * sizeof(int));
}

bool sinkhorn_f64_supported(int N, int M) { return s64_resident_supported(N, M) || s64_wide_supported(N, M); }

template <int NC2>
static int s64_wide_launches(const Wk64Args& a0, hipStream_t s) {
    Wk64Args a = a0;
    const size_t lds_it = (size_t)9 * a.Mp * sizeof(double), lds_fin = (size_t)a.Mp * (3 * sizeof(double) + sizeof(int));
    static std::atomic<unsigned long long> optin_it{0}, optin_fin{0};
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(sinkhorn_f64_wide_iter_kernel<NC2>), lds_it, optin_it, "sinkhorn_f64 (streaming) LDS")) return rc;
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(sinkhorn_f64_wide_final_kernel<NC2>), lds_fin, optin_fin, "sinkhorn_f64 (streaming, last) LDS")) return rc;
    hipLaunchKernelGGL(sinkhorn_f64_wide_init_kernel, dim3((unsigned)(a.B * ((a.N + 7) / 8))), dim3(512), 0, s, a);
    for (int it = 0; it <= a.iters; ++it) {
        a.it = it;
        hipLaunchKernelGGL(sinkhorn_f64_wide_b_kernel, dim3((unsigned)(a.B * (a.Mp >> 7))), dim3(128), 0, s, a);
        if (it < a.iters) hipLaunchKernelGGL(sinkhorn_f64_wide_iter_kernel<NC2>, dim3((unsigned)(a.B * a.G)), dim3(512), lds_it, s, a);
    }
    hipLaunchKernelGGL(sinkhorn_f64_wide_final_kernel<NC2>, dim3((unsigned)(a.B * a.G)), dim3(512), lds_fin, s, a);
    return mdgat_check_hip(hipGetLastError(), "sinkhorn_f64 (streaming) launch");
}

static int launch_sinkhorn_f64_wide(int B, int N, int M, const double* scores, double alpha, int iters, double* Z64, float* Z32, int inner, int* rbest_idx,
                                    float* rbest_val, int* cbest_idx, float* cbest_val, void* workspace, hipStream_t s, const double* alpha_dev) {
    Wk64Args a{};
    a.scores = scores; a.alpha = alpha; a.alpha_dev = alpha_dev; a.B = B; a.N = N; a.M = M; a.iters = iters; a.inner = inner;
    a.Mp = (M + 1 + 127) & ~127; a.G = (N + W64_ROWS - 1) / W64_ROWS;
    a.Z64 = Z64; a.Z32 = Z32; a.rbest_idx = rbest_idx; a.rbest_val = rbest_val;
    char* w = static_cast<char*>(workspace);
    a.K = reinterpret_cast<double*>(w); w += s64_align((size_t)B * N * a.Mp * sizeof(double));
    a.rmax = reinterpret_cast<double*>(w); w += s64_align((size_t)B * N * sizeof(double));
    a.av = reinterpret_cast<double*>(w); w += s64_align((size_t)B * (N + 1) * sizeof(double));
    a.P = reinterpret_cast<double*>(w); w += s64_align((size_t)B * 2 * a.G * a.Mp * sizeof(double));
    a.bvec = reinterpret_cast<double*>(w); w += s64_align((size_t)B * a.Mp * sizeof(double));
    double* sval = reinterpret_cast<double*>(w); w += s64_align((size_t)B * a.G * M * sizeof(double));
    int* sidx = reinterpret_cast<int*>(w);
    a.cslab_idx = cbest_idx ? sidx : nullptr; a.cslab_val = cbest_idx ? sval : nullptr;
    const int nc2 = a.Mp >> 7;
    int rc = nc2 <= 5 ? s64_wide_launches<5>(a, s) : nc2 <= 9 ? s64_wide_launches<9>(a, s) : s64_wide_launches<W64_NC2MAX>(a, s);
    if (rc) return rc;
    if (cbest_idx) {
        const size_t total = (size_t)B * M;
        hipLaunchKernelGGL(sinkhorn_f64_merge_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, s, sidx, sval, B, a.G, M,
                           cbest_idx, cbest_val);
        if (int rc2 = mdgat_check_hip(hipGetLastError(), "sinkhorn_f64 merge launch")) return rc2;
    }
    return MDGAT_OK;
}

int launch_sinkhorn_f64(int B, int N, int M, const double* scores, double alpha, int iters, double* Z64, float* Z32, int inner, int* rbest_idx,
                        float* rbest_val, int* cbest_idx, float* cbest_val, void* workspace, size_t workspace_bytes, unsigned* error_word,
                        hipStream_t s, const double* alpha_dev) {
    if (B <= 0) return MDGAT_OK;
    if (!sinkhorn_f64_supported(N, M)) { mdgat_set_error("fp64 Sinkhorn: %d x %d keypoints > %d supported", N, M, 128 * W64_NC2MAX - 1); return MDGAT_ERR_UNSUPPORTED; }
    if (!workspace || workspace_bytes < sinkhorn_f64_workspace_bytes(B, N, M) || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        mdgat_set_error("fp64 Sinkhorn: workspace too small or not 256-byte aligned");
        return MDGAT_ERR_BAD_ARG;
    }
    if (s64_use_wide(N, M))
        return launch_sinkhorn_f64_wide(B, N, M, scores, alpha, iters, Z64, Z32, inner, rbest_idx, rbest_val, cbest_idx, cbest_val, workspace, s, alpha_dev);
    // Eight-wave workgroups at every launch size.  Sixteen-wave ones (64 rows, half the partners, one per CU) were 7 % faster once the
    // launch no longer gives every workgroup a CU (32 pairs of 512: 818 against 878 us per 100 iterations; one pair 602 against 432) -
    // at 128 registers with spills, a second grouping of the column sums (so results that depend on the batch a pair travels in), and a
    // code generation that turned to garbage under a harmless edit (profiles/NOTES_r6.md section 12): not kept.
    constexpr int waves = 8;
    const int G = (N + 4 * waves - 1) / (4 * waves);
    char* w = static_cast<char*>(workspace);
    Sk64Args a{};
    a.scores = scores; a.alpha = alpha; a.alpha_dev = alpha_dev; a.B = B; a.N = N; a.M = M; a.iters = iters; a.G = G; a.inner = inner;
    a.Z64 = Z64; a.Z32 = Z32; a.rbest_idx = rbest_idx; a.rbest_val = rbest_val;
    const size_t Gmax = (N + 31) / 32;
    a.slots = reinterpret_cast<double*>(w); w += s64_align((size_t)B * 2 * Gmax * S64_SLOT * sizeof(double));
    a.flags = reinterpret_cast<unsigned*>(w); w += s64_align((size_t)B * 3 * Gmax * sizeof(unsigned));
    double* sval = reinterpret_cast<double*>(w); w += s64_align((size_t)B * Gmax * M * sizeof(double));
    int* sidx = reinterpret_cast<int*>(w);
    a.cslab_idx = cbest_idx ? sidx : nullptr; a.cslab_val = cbest_idx ? sval : nullptr;
    a.error_word = error_word;
    // ONE launch of this kernel at a time per device: a launch waits (on its own stream) for the previous one, whatever stream that was on.
    // The workgroups of a pair wait for each other, and a workgroup that waits holds its CU.  Within one launch that is safe - an XCD gets
    // its pairs in order, each as a contiguous run of workgroups, so at most one pair per XCD is partly resident and the other slots hold
    // a complete pair that finishes - and other kernels that occupy CUs are finite.  But three launches in flight can each hold a partly
    // resident pair on an XCD (3 x 15 of its 32 slots) with no slot left for any of them: measured - four streams, launches of 20 and
    // 40 pairs: every spin ran into its bound and the results were garbage.  (Streams being captured into a graph are left alone.)
    int dev = 0;
    (void)hipGetDevice(&dev);
    CoopChain& sr = coop_chain_of(dev);                 // (shared with the clustered fp64 layer tail: coop_chain.hpp)
    std::lock_guard<std::recursive_mutex> lock(sr.m);
    const bool chain = !coop_stream_capturing(s);
    if (chain)
        if (int rc = mdgat_check_hip(coop_chain_wait(sr, s), "fp64 Sinkhorn: wait for the previous launch")) return rc;
    if (int rc = mdgat_check_hip(hipMemsetAsync(a.flags, 0, (size_t)B * 3 * G * sizeof(unsigned), s), "memset(fp64 Sinkhorn flags)")) return rc;
    const int groups = (B + 7) / 8;
    const size_t lds = s64_lds_bytes(waves);
    static std::atomic<unsigned long long> optin8{0};
    if (int rc = mdgat_lds_optin(reinterpret_cast<const void*>(sinkhorn_f64_kernel<8>), lds, optin8, "sinkhorn_f64 LDS")) return rc;
    hipLaunchKernelGGL(sinkhorn_f64_kernel<8>, dim3(groups * 8 * G), dim3(512), lds, s, a);
    if (int rc = mdgat_check_hip(hipGetLastError(), "sinkhorn_f64 launch")) return rc;
    if (chain)
        if (int rc = mdgat_check_hip(coop_chain_record(sr, s), "fp64 Sinkhorn: record")) return rc;
    if (cbest_idx) {
        const size_t total = (size_t)B * M;
        hipLaunchKernelGGL(sinkhorn_f64_merge_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, s, sidx, sval, B, G, M,
                           cbest_idx, cbest_val);
        if (int rc = mdgat_check_hip(hipGetLastError(), "sinkhorn_f64 merge launch")) return rc;
    }
    return MDGAT_OK;
}

// ---- per-op entry points (include/mdgat_hip.h) ----
size_t sinkhorn_f64_bests_bytes(int B, int N, int M) { return s64_align((size_t)B * N * 4) * 2 + s64_align((size_t)B * M * 4) * 2; }

extern "C" size_t mdgat_sinkhorn_f64_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return sinkhorn_f64_workspace_bytes(B, N, M) + sinkhorn_f64_bests_bytes(B, N, M);
}

extern "C" int mdgat_sinkhorn_f64(int B, int N, int M, const double* scores, double bin_score, int iters, double* Z, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!scores || !Z) { mdgat_set_error("mdgat_sinkhorn_f64: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (N <= 0 || M <= 0 || iters < 0) { mdgat_set_error("mdgat_sinkhorn_f64: bad shape N=%d M=%d iters=%d", N, M, iters); return MDGAT_ERR_BAD_ARG; }
    return launch_sinkhorn_f64(B, N, M, scores, bin_score, iters, Z, nullptr, 0, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, nullptr,
                               static_cast<hipStream_t>(stream));
}

extern "C" int mdgat_sinkhorn_f64_extract(int B, int N, int M, const double* scores, double bin_score, int iters, int mode, float match_threshold,
                                          int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* Z_or_null, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    if (!scores || !matches0 || !matches1 || !mscores0 || !mscores1) { mdgat_set_error("mdgat_sinkhorn_f64_extract: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (N <= 0 || M <= 0 || iters < 0) { mdgat_set_error("mdgat_sinkhorn_f64_extract: bad shape N=%d M=%d iters=%d", N, M, iters); return MDGAT_ERR_BAD_ARG; }
    if (!workspace || workspace_bytes < mdgat_sinkhorn_f64_workspace_bytes(B, N, M) || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        mdgat_set_error("mdgat_sinkhorn_f64_extract: workspace too small or not 256-byte aligned");
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(workspace) + sinkhorn_f64_workspace_bytes(B, N, M);
    int* ri = reinterpret_cast<int*>(w); w += s64_align((size_t)B * N * 4);
    float* rv = reinterpret_cast<float*>(w); w += s64_align((size_t)B * N * 4);
    int* ci = reinterpret_cast<int*>(w); w += s64_align((size_t)B * M * 4);
    float* cv = reinterpret_cast<float*>(w);
    const int inner = mode >= MDGAT_EXTRACT_THRESHOLD;
    if (int rc = launch_sinkhorn_f64(B, N, M, scores, bin_score, iters, nullptr, Z_or_null, inner, ri, rv, ci, cv, workspace, sinkhorn_f64_workspace_bytes(B, N, M),
                                     nullptr, s))
        return rc;
    const SkExtract ex{mode, match_threshold, matches0, matches1, mscores0, mscores1, 0, nullptr, 0u};
    return launch_extract_from_bests(B, N, M, &ex, ri, rv, ci, cv, s);
}
