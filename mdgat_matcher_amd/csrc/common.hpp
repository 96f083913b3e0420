// Shared declarations for libmdgat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/mdgat_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define MDGAT_LOG2E 1.4426950408889634f
#define MDGAT_LN2 0.6931471805599453f

// Split-f16 operands: x = hi + lo / 2048 with hi = f16(x), lo = f16((x - hi) * 2048) - 22 mantissa bits, so
// hi.hi + (hi.lo + lo.hi) / 2048 on the f16 matrix cores is an fp32-class product.
#define MDGAT_SPLIT_SCALE 2048.0f
#define MDGAT_SPLIT_INV 0.00048828125f
#ifdef __HIPCC__
__device__ __forceinline__ void mdgat_split(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * MDGAT_SPLIT_SCALE);
}
// The q / k / v operands of the attention kernels carry their residual UNSCALED, x = hi + lo with lo = f16(x - hi): for
// |x| < 0.25 that is an f16 denormal, which the matrix cores honour exactly (tools/ubench/mfma_denorm.hip), its rounding
// is at most 2^-25 absolute (the class of an fp32 product for operands of order one: tools/precision_probe.py, UNSCALED=1),
// and hi.hi + hi.lo + lo.hi then accumulate in ONE register set - no second accumulator, no combine per logit.
__device__ __forceinline__ void mdgat_split_unscaled(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}
#endif

// Key index (within the source frame) of logit register (jb, r) of a lane, relative to the lane's offset, in the S^T = K Q^T
// fragments of the attention kernels (attention.hip).
struct KeyLayout32 { static __device__ constexpr int koff(int jb, int r) { return jb * 32 + 16 * (r >> 3) + (r & 7); } };       // 32x32 fragments, + 8 (lane >> 5)
struct KeyLayout16 { static __device__ constexpr int koff(int jb, int r) { return 16 * (4 * jb + (r >> 2)) + (r & 3); } };      // 16x16 fragments, + 4 (lane >> 4)

// row of the 32x32 MFMA C/D fragment held in accumulator register r by a lane of half `hi`
// (cdna_hip_programming.md section 3: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31).
__device__ __forceinline__ constexpr int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

void mdgat_set_error(const char* fmt, ...);
int mdgat_check_hip(hipError_t e, const char* what);
// Opt a kernel in to `lds` bytes of dynamic LDS, once per device (`done` = the caller's static bitmap of devices;
// one process may drive several GPUs from several threads: torch.nn.DataParallel).
inline int mdgat_lds_optin(const void* kern, size_t lds, std::atomic<unsigned long long>& done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;       // (bit 63: never cached)
    if (dev < 63 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return MDGAT_OK;
    if (int rc = mdgat_check_hip(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), what)) return rc;
    if (dev < 63) done.fetch_or(1ull << dev, std::memory_order_release);
    return MDGAT_OK;
}

// ---- packed-weight blob layout (must match mdgat_matcher_amd/pack.py) --------------------------
struct BlobLayout {
    // encoders (BN folded)
    size_t kenc0_w, kenc0_b;   // [32][4], [32]
    size_t denc0_w, denc0_b;   // [64][33], [64]
    size_t kenc1_w, kenc1_b;   // [64][32], [64]
    size_t kenc2_w, kenc2_b;   // [128][64], [128]
    size_t denc1_w, denc1_b;   // [128][64], [128]
    size_t encl_w, encl_b;     // [128][256] = [denc.6 | kenc.9], [128]
    size_t layer0;             // start of layer 0
    size_t layer_stride;
    // per layer (relative to layer start)
    size_t qkv_w, qkv_b;       // [384][128] rows = which*128 + head*32 + dim, [384]
    size_t mlp1_w, mlp1_b;     // [256][256] cols = [x | head-major message (merge folded)], [256]
    size_t mlp2_w, mlp2_b;     // [128][256], [128]
    size_t final_w, final_b;   // [128][128], [128]
    size_t bin_score;          // [1] (+3 pad)
    size_t total;
};
BlobLayout mdgat_blob_layout(int L);

// ---- kernel launchers (all asynchronous on `stream`) --------------------------------------------
struct GemmArgs {
    const float* A0; int lda0; int K0;   // columns [0, K0) of the input come from A0
    const float* A1; int lda1;           // columns [K0, K) from A1 (unused when K0 == K)
    const float* W; int ldw;             // [N][K]
    const float* bias;                   // [N] or nullptr
    const float* R; int ldr;             // residual [M][N] or nullptr
    float* C; int ldc;
    int M, N, K;
    int relu;
    float scale;                         // applied to the accumulator before bias
    int batch;                           // grid.z
    long long sA, sW, sC;                // batch strides (elements) of A0, W, C
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
// score matrix [B][N][M] = mdesc0 . mdesc1^T * scale from mdesc [B][N + M][128] (scores.hip)
// zero / zero_bytes (optional, 16-byte granular): memory the kernel clears on the side - the forward hands it the exchange
// slots of the Sinkhorn kernel that runs next (sinkhorn_slots_clear_bytes) instead of a memset launch
// guard (optional, host-mapped): raised when an operand is outside the f16 operand range or not finite
int launch_scores(int B, int N, int M, const float* mdesc, float* scores, float scale, hipStream_t s, void* zero = nullptr, size_t zero_bytes = 0,
                  unsigned* guard = nullptr);

// fused encoders (encoder.hip).  es = split weights [kenc.3 64x2x32 | kenc.6 128x2x64 | denc.0 64x2x48 | denc.3 128x2x64 |
// last convs 128x2x256]; inputs either as separate arrays or as raw 37-float frame records
struct EncoderLaunch {
    const float *kpts0, *sigma0, *fpfh0, *kpts1, *sigma1, *fpfh1;
    const float *rec0, *rec1;
    int normalize;
    const float* w; const BlobLayout* bl;
    const _Float16* es;
    float* x;
    int B, N, M;
};
constexpr size_t MDGAT_ENC_SPLIT_HALVES = 64 * 64 + 128 * 128 + 64 * 96 + 128 * 128 + 128 * 512;
int launch_encoder(const EncoderLaunch& p, hipStream_t s);
int launch_split_rows_pad(const float* w, _Float16* out, int rows, int Kin, int Kpad, hipStream_t s);

// q/k/v of every point in the split-f16 operand layouts of the attention kernel (attention.hip)
struct Qkv16 {
    _Float16* q16;    // [B][P][4 heads][2 planes][32 dims], pre-scaled by log2(e) / sqrt(32); planes = (hi, unscaled residual)
    _Float16* k16;    // [B][P][4][2][32]
    _Float16* vt16;   // [B][4][2][32][PP]: V transposed, keys contiguous; frame 1 starts at column Npad
    int Npad, PP;     // Npad = N rounded up to 32, PP = Npad + (M rounded up to 32)
};
size_t mdgat_qkv16_halves(int B, int N, int M);
Qkv16 mdgat_qkv16_carve(_Float16* base, int B, int N, int M);
int launch_qkv_split(int B, int N, int M, const float* qkv, const Qkv16& out, hipStream_t s);
// mode: mdgat_attention_mode (1 = single-f16 products where implemented)
// sel (parity tap, may be NULL): the kept keys of a dynamic layer as bit masks [B][4][P][W], W = ceil(max(N, M) / 32)
int launch_attention(int B, int N, int M, int cross, int topk, const Qkv16& qkv, float* msg, hipStream_t s, int mode = 0, uint32_t* sel = nullptr);
// full attention as a stream of 64-key chunks (attention_stream.hip); frames with key counts that are multiples of 64
bool attention_stream_supported(int N, int M);
int launch_attention_stream(int B, int N, int M, int cross, const Qkv16& qkv, float* msg, hipStream_t s, int mode = 0);
// measurement only: the Q K^T phase of the streamed kernel in isolation (msg[row][head * 32] receives the row maximum)
int launch_attention_qk_probe(int B, int N, int M, int cross, const Qkv16& qkv, float* msg, hipStream_t s);
int launch_qk_phase_probe(int B, int N, int M, int cross, int nq_sets, const Qkv16& qkv, float* msg, hipStream_t s);

// fused layer tail (layer.hip): [mlp.0 -> mlp.3 -> residual] of one layer + q|k|v projection of the next
struct LayerLaunch {
    float* x; const float* msg;
    const _Float16 *w1s, *w2s, *w3s;   // split weight images [rows][hi K | lo K | 8 pad]
    const _Float16 *w1f, *w2f, *w3f;   // the same matrices in fragment order (launch_frag_image), read by layer_split.hip
    const float *b1, *b2, *b3;
    Qkv16 out;                         // mode3 == 1
    float* mdesc;                      // mode3 == 2
    int R, N, M;
    int do_mlp;                        // 0: projection only
    int mode3;                         // 1: q|k|v, 2: final_proj
    unsigned* guard;                   // optional (host-mapped): set to 1 when an input row holds a value outside the f16
                                       // operand range (|v| >= MDGAT_F16_GUARD) or a non-finite one
};
int launch_layer(const LayerLaunch& p, hipStream_t s);
// rowh >= 2 K: row pitch (halves); the first nperm rows are written in the P/Q row order of layer.hip
int launch_split_rows(const float* w, _Float16* out, int rows, int K, int rowh, int nperm, hipStream_t s);
// row image [rows][rowh] (launch_split_rows) -> fragment order [rows / 16][K / 32][plane][lane = row l15 + 16 g][8 halves] (layer_split.hip)
int launch_frag_image(const _Float16* img, _Float16* out, int rows, int K, int rowh, hipStream_t s);

struct SkExtract {   // match extraction to run after (or fused into) the Sinkhorn kernel
    int mode; float thr;
    int64_t *m0, *m1;
    float *s0, *s1;
    int defer_alldust;   // the batch-wide "no keypoint of frame 0 matched" rule (mdgat.py:465-467) is applied later by
                         // launch_alldust_fixup() over the whole batch (the forward runs large batches in slices)
    unsigned* matched;   // optional (host-mapped): receives matched_token when some frame-0 keypoint of the launch is matched
    unsigned matched_token;
};
int launch_alldust_fixup(int B, int N, int M, int mode, const int64_t* m0, float* s1, hipStream_t s);
// Asynchronous status words of a handle (host-mapped memory the kernels write; read by the host after a synchronisation).
constexpr int MDGAT_STATUS_SK_FALLBACK = 0;   // the Sinkhorn cluster kernel lost a partner workgroup: the launch was redone by the streaming kernel
constexpr int MDGAT_STATUS_RANGE = 1;         // an activation left the f16 operand range or is not finite: the outputs are invalid
constexpr int MDGAT_STATUS_MATCHED = 4;       // first of MDGAT_MATCH_SLOTS words: slot (token % slots) receives the token of a forward whose extraction matched
                                              // at least one frame-0 keypoint (mdgat.py:465: the reference tests valid0.sum() on the host; mdgat_matched_any
                                              // reads the call's slot instead of a reduction + copy; a slot per call: concurrent callers of one handle)
constexpr int MDGAT_MATCH_SLOTS = 256;
constexpr int MDGAT_STATUS_WORDS = MDGAT_STATUS_MATCHED + MDGAT_MATCH_SLOTS;
constexpr float MDGAT_F16_GUARD = 6.0e4f;     // f16 max is 65504; the split's hi plane must stay finite
// status (optional, device pointer to MDGAT_STATUS_WORDS host-mapped words).  Zfb: where the streaming fallback puts Z when
// the cluster kernel lost a partner and the caller wanted no Z (NULL: the cluster launch is cooperative instead).
// slots_cleared: the first sinkhorn_slots_clear_bytes(N, M) bytes of ws are already zero on the stream (no memset launch).
int launch_sinkhorn(int B, int N, int M, const float* scores, const float* bin_score_dev, float bin_score_host,
                    int iters, float* Z, void* ws, size_t ws_bytes, const SkExtract* ex, hipStream_t s, unsigned* status = nullptr,
                    float* Zfb = nullptr, bool slots_cleared = false);
size_t sinkhorn_slots_clear_bytes(int B, int N, int M);   // 0: the shape does not use the cluster kernel
size_t mdgat_sinkhorn_ws_bytes_impl(int B, int N, int M);

int launch_extract(int B, int N, int M, const float* Z, int mode, float thr, int64_t* m0, int64_t* m1,
                   float* s0, float* s1, hipStream_t s);
int launch_extract_from_bests(int B, int N, int M, const SkExtract* ex, const int* rbest_idx, const float* rbest_val, const int* cbest_idx,
                              const float* cbest_val, hipStream_t s);

int launch_pose(int B, int N, int M, const float* kpts0, const float* kpts1, const int64_t* matches0, const double* T_gt,
                double inlier_dist, double* T, double* stats, hipStream_t s);
int launch_gt_match(int B, int N, int M, const float* kpts0, const float* kpts1, const double* T0, const double* T1,
                    double threshold, int mutual, int64_t* gt0, int64_t* gt1, int64_t* rep, hipStream_t s);

// out[b][i][j] = scale <A[b][i], Bm[b][j]> - col_bias[b][j] over 128 channels, split-f16 products (scores.hip)
int launch_dots(int B, int N, int M, const float* A, size_t strideA, const float* Bm, size_t strideB, float* out, float scale,
                const float* col_bias, hipStream_t s, void* zero = nullptr, size_t zero_bytes = 0, unsigned* guard = nullptr);
int launch_knn(int B, int C, int N, int M, int k, const float* x, const float* src, int64_t* idx, int64_t* adj,
               void* ws, size_t ws_bytes, hipStream_t s);
size_t mdgat_knn_ws_bytes_impl(int B, int C, int N, int M);
