/*
 * mdgat_hip.h - C ABI of libmdgat_hip.so: the MI355X (gfx950) implementation of the inference
 * hot path of MDGAT-matcher.
 *
 * The reference (nubot-nudt/MDGAT-matcher) is pure Python/PyTorch: it has NO native boundary.  The
 * seam this library sits behind is the Python class models.mdgat.MDGAT
 * (/root/reference/models/mdgat.py:315-603) as called from test.py:201 and
 * test_registration_metric.py:202.  Each entry point below names the reference code it replaces.
 * The Python host side (mdgat_matcher_amd/mdgat.py) binds these with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is DEVICE memory unless stated otherwise;
 *   - all tensors are owned by the caller (PyTorch); the library keeps only its packed weights;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 on success, a negative mdgat_status otherwise; mdgat_last_error() gives the text;
 *   - a handle is bound to one device.  Thread safety: mdgat_forward / mdgat_forward_frames on ONE handle may be called from
 *     several host threads (and streams) - the library serialises the enqueue of a call internally (a mutex in the handle: the
 *     second lane's stream, its fork / join events and the profiling state are per handle); the device work of calls issued on
 *     different streams overlaps as far as those streams allow, but the halves that run on the handle's second lane queue on
 *     that one stream in call order.  Each call needs its own workspace while it is in flight.  mdgat_set_lanes,
 *     mdgat_load_weights, mdgat_profile and mdgat_destroy must not race with a forward on the same handle.
 *
 * Layouts (fp32, row-major, innermost last)
 *   keypoints  kpts  [B][N][3]      saliency sigma [B][N]      FPFH fpfh [B][N][33]
 *   descriptors x    [B][P][128]    P = N + M, frame-0 points first, then frame-1 points
 *   q/k/v      qkv   [B][P][3][4][32]  (which, head, dim) - heads de-interleaved, see pack.py
 *   scores           [B][N][M]      couplings / Z  [B][N+1][M+1]
 *   matches    int64 [B][N] / [B][M]    matching scores fp32 [B][N] / [B][M]
 */
#ifndef MDGAT_HIP_H
#define MDGAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDGAT_MAX_LAYERS 64 /* 2L <= 64 */
#define MDGAT_D 128         /* descriptor_dim (mdgat.py:317) */
#define MDGAT_HEADS 4       /* mdgat.py:255 */
#define MDGAT_FPFH 33       /* mdgat.py:148 */

typedef enum {
    MDGAT_OK = 0,
    MDGAT_ERR_BAD_ARG = -1,     /* e.g. top-k larger than the number of keys (torch.topk raises there) */
    MDGAT_ERR_HIP = -2,         /* a HIP runtime call failed */
    MDGAT_ERR_UNSUPPORTED = -3, /* shape / config outside what the kernels implement */
    MDGAT_ERR_NO_WEIGHTS = -4
} mdgat_status;

/* match-extraction variants of mdgat.py:441-483 */
typedef enum {
    MDGAT_EXTRACT_DUSTBIN = 0,        /* loss_method != 'superglue', mutual_check False (461-464, 480-483) */
    MDGAT_EXTRACT_DUSTBIN_MUTUAL = 1, /* loss_method != 'superglue', mutual_check True  (469-478) */
    MDGAT_EXTRACT_THRESHOLD = 2,      /* loss_method == 'superglue', mutual_check False (444, 455-458) */
    MDGAT_EXTRACT_THRESHOLD_MUTUAL = 3 /* loss_method == 'superglue', mutual_check True (447-453) */
} mdgat_extract_mode;

/* Arithmetic of the attention products (q.k and p.v).  The default carries every operand as two f16 halves and is
 * fp32-faithful (the parity bar); MDGAT_ATTENTION_F16 rounds q, k, v and the probabilities to ONE f16 each (fp32
 * accumulation, fp32 softmax statistics): a throughput mode outside the parity bar (max|dZ| ~ 2e-3), applied where a
 * kernel implements it (full attention with key counts that are multiples of 64, dynamic attention at 512 keys)
 * and ignored elsewhere. */
typedef enum { MDGAT_ATTENTION_FP32 = 0, MDGAT_ATTENTION_F16 = 1 } mdgat_attention_mode;

/* Arithmetic of the forward.  MDGAT_ARITH_FP32 (default): fp32-class products everywhere (split-f16 on the matrix cores) -
 * Z within 1e-4 of the fp64 reference wherever the top-k selections of the dynamic layers agree with the reference's, which
 * near-ties below fp32 resolution do not always (about 1.5 rows per pair at 512 keypoints).
 * MDGAT_ARITH_FP64: the reference's own arithmetic (the reference runs net.double(), test.py:193) where it decides anything
 * discontinuous: the encoders (mdgat.py:392-393) and the propagation layers up to and including the LAST dynamic layer
 * (mdgat.py:259-276 with dynamic_attention 196-210) run in fp64 on v_mfma_f64_16x16x4_f64 from fp64 inputs and fp64 weights
 * (mdgat_load_weights_f64, mdgat_forward_f64; csrc/f64.hip), so that logits.topk(k) selects what the reference selects; the
 * layers behind it, final_proj, the score matrix and Sinkhorn are continuous and stay on the fp32-class kernels. */
typedef enum { MDGAT_ARITH_FP32 = 0, MDGAT_ARITH_FP64 = 1 } mdgat_arithmetic;

/* Replaces the config dict of MDGAT.__init__ (mdgat.py:325-367) for descriptor == 'FPFH'. */
typedef struct {
    int32_t L;                         /* config['L']: 2L alternating self/cross layers (352-353) */
    int32_t sinkhorn_iters;            /* config['sinkhorn_iterations'] (321) */
    int32_t topk[MDGAT_MAX_LAYERS];    /* per layer 0..2L-1: 0 = attention() (190-194), k > 0 =
                                          dynamic_attention(k) (196-210); schedule of 268-272 is
                                          resolved by the host */
    int32_t extract_mode;              /* mdgat_extract_mode */
    float match_threshold;             /* config['match_threshold'] (322) */
    int32_t attention_mode;            /* mdgat_attention_mode; not a reference key (BASELINE configs[2]) */
    int32_t arithmetic;                /* mdgat_arithmetic; not a reference key (the reference IS fp64) */
    int32_t f64_layers;                /* MDGAT_ARITH_FP64: how many leading propagation layers run in fp64.  0 (what a
                                          zero-initialised config holds): automatic - up to and including the last layer with
                                          topk > 0 (the encoders only when no layer has one); n > 0: exactly n (2L: all of them);
                                          MDGAT_F64_ENCODERS_ONLY (-1): none, the encoders only */
    int32_t f64_sinkhorn;              /* MDGAT_ARITH_FP64: the tail as well - EVERY layer, final_proj, the score matrix and the optimal
                                          transport in fp64, the extraction's arg-maxes decided on the fp64 Z (csrc/sinkhorn_f64.hip).
                                          0 (a zero-initialised config): automatic - on for frames of at most 2175 keypoints (and
                                          f64_layers == 0; up to 575 the Sinkhorn holds the couplings in registers, beyond that it
                                          streams them from memory), else the fp32-class tail; 1: required (larger frames are refused);
                                          MDGAT_F64_SINKHORN_OFF (-1): the fp32-class tail behind the last dynamic layer (rounds 5 / 6:
                                          Z good to 7e-6, an arg-max whose candidates lie closer than that may fall the other way) */
} mdgat_config;
#define MDGAT_F64_ENCODERS_ONLY (-1)
#define MDGAT_F64_SINKHORN_OFF (-1)

typedef struct mdgat_handle mdgat_handle;

/* Optional taps for parity tests: any pointer may be NULL. */
typedef struct {
    float* x_enc;    /* [B][P][128]        encoder sum (mdgat.py:392-393) */
    float* x_layers; /* [2L][B][P][128]    descriptors after every layer (mdgat.py:274) */
    float* mdesc;    /* [B][P][128]        final_proj output (mdgat.py:397) */
    float* scores;   /* [B][N][M]          pre-OT scores (mdgat.py:430-431) */
    uint32_t* topk_sel; /* [2L][mdgat_topk_sel_words(B, N, M)]  the keys every dynamic layer kept (the index set of
                           mdgat.py:202 `logits.topk(k)`) as bit masks [B][4 heads][P queries][W words], W = ceil(max(N, M) / 32):
                           bit j of word w = key 32 w + j of the query's source frame; slices of full-attention layers are
                           left untouched.  Parity tests feed this selection to the oracle to separate near-tie flips of
                           the discontinuous top-k from arithmetic error. */
} mdgat_taps;
size_t mdgat_topk_sel_words(int B, int N, int M);

/* ---- lifetime ------------------------------------------------------------------------------ */

/* MDGAT(config).to(device) (test.py:156, 172).  Fails with MDGAT_ERR_UNSUPPORTED unless the
 * device is gfx950. */
int mdgat_create(const mdgat_config* cfg, int device, mdgat_handle** out);

/* net.load_state_dict(...) (test.py:159) after host-side packing (BN folded into the convs, heads
 * de-interleaved, merge folded into mlp.0; mdgat_matcher_amd/pack.py).  Every weight is the fp32
 * rounding of the folded fp64 value.  `blob` holds
 * mdgat_blob_floats(L) fp32 values; `on_device` != 0 when it is device memory (e.g. received by
 * an RCCL broadcast), else host memory. */
int mdgat_load_weights(mdgat_handle* h, const float* blob, size_t n_floats, int on_device);
size_t mdgat_blob_floats(int L);

/* MDGAT_ARITH_FP64: the same blob in fp64 - the folded weights BEFORE their rounding to fp32 (pack.py computes them in fp64;
 * same layout, mdgat_blob_floats(L) doubles).  Needed in addition to mdgat_load_weights (the layers behind the last dynamic
 * one run on the fp32 blob).  Fails with MDGAT_ERR_BAD_ARG on a handle created with another arithmetic. */
int mdgat_load_weights_f64(mdgat_handle* h, const double* blob, size_t n_doubles, int on_device);
/* Device pointer to the handle's fp64 blob (NULL before mdgat_load_weights_f64; for an RCCL broadcast). */
double* mdgat_weights_f64_device_ptr(mdgat_handle* h);

/* Device pointer to the handle's packed weights (for an RCCL broadcast from rank 0). */
float* mdgat_weights_device_ptr(mdgat_handle* h);

void mdgat_destroy(mdgat_handle* h);
const char* mdgat_last_error(void);

/* ---- whole forward ---------------------------------------------------------------------------- */

/* Scratch the caller must provide to mdgat_forward for a batch of B pairs with N / M keypoints. */
size_t mdgat_workspace_bytes(const mdgat_handle* h, int B, int N, int M);

/* MDGAT.forward (mdgat.py:369-483) for descriptor == 'FPFH', loss excluded.
 * Z (optional, may be NULL) receives log_optimal_transport's output [B][N+1][M+1]. */
int mdgat_forward(mdgat_handle* h, int B, int N, int M,
                  const float* kpts0, const float* sigma0, const float* fpfh0,
                  const float* kpts1, const float* sigma1, const float* fpfh1,
                  int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1,
                  float* Z, const mdgat_taps* taps,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The same forward in the reference's arithmetic (handle created with MDGAT_ARITH_FP64, both blobs loaded): inputs in fp64 as
 * the reference receives them (test.py:194-199 moves the loader's float64 tensors to the device; layouts as above), outputs as
 * mdgat_forward (Z and the matching scores come from the fp32 Sinkhorn: within 1e-4 of the reference's fp64 values).  The taps
 * receive fp32 roundings of the fp64 stages. */
int mdgat_forward_f64(mdgat_handle* h, int B, int N, int M,
                      const double* kpts0, const double* sigma0, const double* fpfh0,
                      const double* kpts1, const double* sigma1, const double* fpfh1,
                      int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1,
                      float* Z, const mdgat_taps* taps,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The same forward fed with the loader's raw frame records instead of separate arrays: frames [B][N][37] fp32,
 * one record per keypoint = xyz(3) | saliency(1) | FPFH(33), the layout of the KITTI keypoint files that
 * SparseDataset.__getitem__ reads (load_data.py:146-165).  normalize_fpfh != 0 applies the loader's L2
 * normalisation of the FPFH part (load_data.py:290-292) inside the encoder kernel.  On a MDGAT_ARITH_FP64 handle the records are
 * taken through the loader's own sequence - the normalisation in float32 exactly as numpy computes it (its pairwise summation
 * order, correctly rounded square root / reciprocal / products), then widened to double (load_data.py:294-295) - so that the
 * forward sees bit for bit the inputs the reference's forward sees. */
int mdgat_forward_frames(mdgat_handle* h, int B, int N, int M, const float* frames0, const float* frames1,
                         int normalize_fpfh,
                         int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1,
                         float* Z, const mdgat_taps* taps,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Asynchronous status of the handle's forwards since the last call with clear != 0 (read it after synchronising the
 * stream: the forward itself never synchronises).  *range_violation != 0: an activation left the f16 operand range
 * (|v| >= 6e4) or a non-finite value reached a kernel - every operand is carried as f16 head + f16 residual (DESIGN.md
 * section 3), so the outputs of those calls are invalid.  The fp64 layers of MDGAT_ARITH_FP64 raise it as well: every fp64
 * product tests its outputs by their exponent bits (not finite, or |v| >= 2^500 - beyond which a logit could overflow and the
 * softmax meet inf - inf; csrc/f64.hip: f64_out_of_range), between the layers, not only at the hand-over; the function then returns MDGAT_ERR_UNSUPPORTED (and the next
 * mdgat_forward on the handle does, if nobody asked before).  *sinkhorn_fallback != 0: informational - a Sinkhorn cluster
 * launch lost a partner workgroup (device shared with other work) and was redone by the streaming kernel inside the same
 * call; the results are valid.  The reference has no counterpart (ATen raises nothing either: it returns inf / NaN). */
int mdgat_async_status(mdgat_handle* h, int clear, unsigned* sinkhorn_fallback, unsigned* range_violation);

/* After the caller's synchronisation: *matched = 1 when the forward that carried `token` matched at least one frame-0 keypoint
 * anywhere in its batch (matches0 >= 0), else 0.  This is the reference's host-side test `valid0.sum() == 0`
 * (models/mdgat.py:464-467: integer zero scores when nothing matched) without a reduction kernel and a device-to-host copy: the
 * extraction kernels write the call's token into a host-mapped slot (256 slots, indexed by the token: calls of several threads
 * and streams on one handle do not disturb each other).  mdgat_last_token returns the token of the most recent
 * mdgat_forward / mdgat_forward_f64 / mdgat_forward_frames enqueued on the handle - read it before another thread can enqueue
 * (the Python wrapper does so under its per-handle lock); token 0 stands for that most recent call. */
unsigned mdgat_last_token(mdgat_handle* h);
int mdgat_matched_any(mdgat_handle* h, unsigned token, unsigned* matched);

/* Per-kernel-class timing of mdgat_forward, measured with HIP events on the launch stream (bench.py's
 * roofline leg).  mdgat_profile(h, enable, ms, launches) returns the time (ms) and launch count
 * accumulated per class since the previous call in ms[MDGAT_PROF_CLASSES] / launches[...] (either may
 * be NULL), clears them, and switches the instrumentation on or off.  While it is on, mdgat_forward
 * ends with a stream synchronisation; an interval covers the launch(es) of the class and the gap
 * before them on that lane's stream (with two lanes in flight the launches of the two lanes share the device: an interval is
 * then a launch's duration under that sharing, as a kernel trace reports it). */
enum {
    MDGAT_PROF_ENCODER = 0,        /* both encoders up to the summed descriptor */
    MDGAT_PROF_LAYER = 1,          /* fused mlp + residual + next q/k/v projection (or final_proj) */
    MDGAT_PROF_ATTENTION_FULL = 2,
    MDGAT_PROF_ATTENTION_TOPK = 3,
    MDGAT_PROF_SCORES = 4,         /* score matrix */
    MDGAT_PROF_SINKHORN = 5,
    MDGAT_PROF_EXTRACT = 6,
    MDGAT_PROF_LAYER_FIRST = 7,    /* the launch before layer 0: q/k/v projection only (a third of a layer launch's work) */
    MDGAT_PROF_LAYER_LAST = 8,     /* the launch after the last layer: mlp + residual + final_proj */
    MDGAT_PROF_F64_GEMM = 9,       /* MDGAT_ARITH_FP64: the Conv1d(k=1) products of the fp64 stages (csrc/f64.hip) */
    MDGAT_PROF_F64_ATTENTION_FULL = 10,
    MDGAT_PROF_F64_ATTENTION_TOPK = 11,
    MDGAT_PROF_F64_OTHER = 12,     /* input assembly, hand-over conversion to fp32 */
    MDGAT_PROF_CLASSES = 13
};
int mdgat_profile(mdgat_handle* h, int enable, double* ms, long long* launches);

/* How a batch is executed (no counterpart in the reference, which hands the whole batch to every ATen kernel,
 * models/mdgat.py:369-483).  lanes = 2 (default; MDGAT_FORWARD_LANES=1 in the environment selects 1): a batch of more than
 * 32 768 keypoints is cut into an even number of balanced slices which alternate between the caller's stream and a second
 * stream owned by the handle - forked from and joined to the caller's stream with events, so the call stays asynchronous
 * and ordered on the caller's stream - because two half-size forwards in flight fill each other's idle phases (csrc/api.hip:
 * forward_batched).  lanes = 1: everything on the caller's stream, slices of 65 536 keypoints beyond 1.5 x that.  Results do
 * not depend on the setting (pairs are independent).  mdgat_workspace_bytes() follows the handle's setting: size the workspace
 * after changing it. */
int mdgat_set_lanes(mdgat_handle* h, int lanes);

/* Which kernel runs the layer tail (mlp + residual + next q|k|v; models/mdgat.py:227-232, 246-248, 274, 397) of a launch.
 * Launches of at most `tiles` tiles of 128 keypoints (default 64; MDGAT_LAYER_SPLIT_TILES in the environment) run the
 * channel-split kernel of csrc/layer_split.hip (32-keypoint workgroups, the eight waves share the output channels: the
 * latency shape, test.py:132 runs batch_size = 1), larger ones the keypoint-split kernels of csrc/layer.hip.  Process-wide;
 * results are bit-identical either way.  tiles = 0: never; tiles < 0: back to the default.  Returns the previous value. */
int mdgat_set_layer_split_tiles(int tiles);

/* MDGAT_ARITH_FP64: how the tail of an fp64 layer (mlp.0 + ReLU, mlp.3 + residual; models/mdgat.py:246-248, 274) and the next
 * layer's q | k | v projection (227-232) are launched.  1 (default; MDGAT_F64_LAYER_FUSION=0 in the environment selects 0): one
 * launch per layer with the hidden activation kept on chip (csrc/layer_f64.hip) - launches of at most a quarter of the compute
 * units in 16-keypoint blocks (one pair per call) with FOUR workgroups per block that split the output channels and exchange the
 * hidden layer and the new x through L2; 2: the same, one workgroup per block at every launch size; 0: three launches of the fp64
 * product kernel (csrc/f64.hip); 16 / 32 / 64: one launch with that many keypoints per workgroup (default: by launch size);
 * mode < 0: back to the default.  Process-wide; the results are bit-identical either way.  Returns the previous value. */
int mdgat_set_f64_layer_fusion(int mode);

/* MDGAT_ARITH_FP64: how full attention (models/mdgat.py:190-194) is launched.  -1 (default; MDGAT_F64_ATTENTION_FORM in the
 * environment selects another): by launch size - launches of at least four 128-query workgroups per compute unit (32 pairs of 512
 * keypoints) give every wave 32 queries and all the keys of the frame, smaller ones split the keys of a 16- / 32-query tile over
 * the four waves of a workgroup and combine them through LDS.  The two forms sum a row's terms in different orders: they agree to
 * rounding (1e-15 relative), not bit for bit, so what a pair returns can differ in the last bits with the size of the batch it
 * travels in.  0: always the split-key form - a pair's result is then bit-identical whatever the batch; 1: always one wave per
 * 32 queries; mode < -1: back to the environment's / default.  Process-wide.  Returns the previous value. */
int mdgat_set_f64_attention_form(int mode);

/* fp64 Sinkhorn: -1 (default; MDGAT_F64_SINKHORN_FORM in the environment selects another): the register-resident kernel wherever it
 * holds the frames (<= 575 keypoints), the streaming form beyond; 1: the streaming form at every size (tests, measurements: the two sum
 * in different orders and agree to rounding); anything else: back to the environment's / default.  Process-wide; sizes of workspaces
 * asked for before a change are not valid after it.  Returns the previous value. */
int mdgat_set_f64_sinkhorn_form(int mode);

/* ---- per-op entry points (unit parity; the forward uses the same kernels) ---------------------- */

/* log_optimal_transport + log_sinkhorn_iterations (mdgat.py:279-308): scores [B][N][M] -> Z. */
int mdgat_sinkhorn(int B, int N, int M, const float* scores, float bin_score, int iters,
                   float* Z, void* workspace, size_t workspace_bytes, void* stream);
size_t mdgat_sinkhorn_workspace_bytes(int B, int N, int M);

/* The same in fp64 - the reference's own arithmetic (mdgat.py:279-308 run in float64, test.py:193) - on fp64 scores: Z [B][N+1][M+1]
 * fp64; N, M <= 2175.  Frames of at most 575 keypoints run in one launch with the couplings held in registers; larger ones stream them
 * from memory, one launch per iteration (csrc/sinkhorn_f64.hip).  workspace: mdgat_sinkhorn_f64_workspace_bytes, 256-byte aligned. */
int mdgat_sinkhorn_f64(int B, int N, int M, const double* scores, double bin_score, int iters,
                       double* Z, void* workspace, size_t workspace_bytes, void* stream);
size_t mdgat_sinkhorn_f64_workspace_bytes(int B, int N, int M);
/* ... followed by the match extraction (mdgat.py:441-483) with every arg-max decided on the fp64 Z (a near tie of 1e-6 between two
 * candidates is below what an fp32 Z resolves); Z_or_null: optional fp32 rounding of Z [B][N+1][M+1]. */
int mdgat_sinkhorn_f64_extract(int B, int N, int M, const double* scores, double bin_score, int iters, int mode, float match_threshold,
                               int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* Z_or_null,
                               void* workspace, size_t workspace_bytes, void* stream);

/* match extraction (mdgat.py:441-483) from Z [B][N+1][M+1]. */
int mdgat_extract(int B, int N, int M, const float* Z, int mode, float match_threshold,
                  int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1,
                  void* stream);

/* attention / dynamic_attention (mdgat.py:190-210) on projected q/k/v in the library layout
 * [B][P][3][4][32]; `cross` selects the other frame as source (mdgat.py:263-266); topk 0 = full.
 * msg [B][P][128] with channel = head*32 + dim.  The kernel consumes q/k/v as split-f16 operands
 * (DESIGN.md section 3); this entry point converts the fp32 input into `workspace` first (16-byte
 * aligned, mdgat_attention_workspace_bytes), the forward writes that layout from its projection. */
int mdgat_attention(int B, int N, int M, int cross, int topk, const float* qkv, float* msg,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t mdgat_attention_workspace_bytes(int B, int N, int M);
/* The same, also returning the kept keys of a dynamic layer (topk > 0) in sel [mdgat_topk_sel_words(B, N, M)]
 * (layout as mdgat_taps.topk_sel). */
int mdgat_attention_sel(int B, int N, int M, int cross, int topk, const float* qkv, float* msg, uint32_t* sel,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Measurement only (bench.py, roofline_qk): the Q K^T contraction of the streamed full-attention kernel in isolation -
 * the same chunk ring, LDS reads and split-f16 MFMA sequence as mdgat_attention, without V^T traffic, softmax and P.V.
 * msg [B][P][128]: element [p][head * 32] receives the row maximum of the base-2 logits, the rest is left alone.
 * N and M must be multiples of 64.  Not part of the matching path (mdgat.py:192 is its subject). */
int mdgat_attention_qk_probe(int B, int N, int M, int cross, const float* qkv, float* msg,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Measurement only (bench.py, roofline_qk.standalone_*): the same Q K^T phase as a standalone kernel with nq_sets = 1 (32
 * queries per wave, the shipped arrangement) or 2 (64 queries per wave: every K fragment read from LDS feeds two sets of
 * products) - the lever VERDICT r2 names; the full kernel cannot hold the registers for it.  Arguments
 * and output as mdgat_attention_qk_probe.  Not part of the matching path. */
int mdgat_attention_qk_probe_sets(int B, int N, int M, int cross, int nq_sets, const float* qkv, float* msg,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* Measurement only (bench.py, roofline.sustained_*): the rate this device SUSTAINS on v_mfma_f32_16x16x32_f16 with random
 * operands resident in registers (two waves per SIMD, nothing but MFMAs, one workgroup pair per CU, `reps` x 24 MFMAs per
 * wave) - the chip clocks to its power budget under such a load, well below the 2.4 GHz the dense peak is quoted at.
 * workspace: >= 136 KB of device memory, 256-byte aligned.  ms_out: duration of the timed launch (HIP events on `stream`, synchronises),
 * flops_out: FLOPs it executed, ticks_out: shader cycles (s_memtime) one wave spent in its loop.  Not part of the
 * matching path; replaces nothing in the reference. */
int mdgat_mfma_probe(int reps, void* workspace, size_t workspace_bytes, float* ms_out, double* flops_out,
                     long long* ticks_out, void* stream);

/* Measurement only (bench.py, exact_mode.roofline): the rate this device sustains on v_mfma_f64_16x16x4_f64 (two waves per
 * SIMD, operands in registers, `reps` x 16 instructions per wave).  workspace: >= 72 KB, 256-byte aligned.  Outputs as
 * mdgat_mfma_probe.  Not part of the matching path. */
int mdgat_mfma_f64_probe(int reps, void* workspace, size_t workspace_bytes, float* ms_out, double* flops_out,
                         long long* ticks_out, void* stream);

/* Unit parity of the fp64 kernels (csrc/f64.hip; the fp64 forward uses the same launches):
 * mdgat_pointwise_f64: as mdgat_pointwise on doubles, no alignment or K granularity required.
 * mdgat_attention_f64: attention / dynamic_attention (mdgat.py:190-210) on fp64 q | k | v rows [B][P][384] ([which][head][dim]),
 * msg [B][P][128]; sel (optional): the kept keys of a dynamic layer, layout of mdgat_taps.topk_sel. */
int mdgat_pointwise_f64(int M, int N, int K, const double* A, int lda, const double* W, int ldw, const double* bias,
                        int relu, const double* R, int ldr, double* C, int ldc, void* stream);
int mdgat_attention_f64(int B, int N, int M, int cross, int topk, const double* qkv, double* msg, uint32_t* sel, void* stream);

/* Conv1d(k=1)(+folded BN)(+ReLU) over points: C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+R).
 * (MLP of mdgat.py:34-46 after folding.)  K must be a multiple of 32; lda/ldw/ldc multiples of 4. */
int mdgat_pointwise(int M, int N, int K, const float* A, int lda, const float* W, int ldw,
                    const float* bias, int relu, const float* R, int ldr, float* C, int ldc,
                    void* stream);

/* knn() / get_graph_feature() (mdgat.py:8-32; dead code there, named by the north star):
 * x [B][N][C], src [B][M][C] (point-major) -> idx int64 [B][N][k] in topk order (nearest first; exact ties in the
 * order of their indices), and, if adj != NULL, the dense 0/1 int64 adjacency [B][N][M].  Any M; k <= 1024.
 * C == 128 (feature space) runs the inner products on the matrix cores when `workspace` holds
 * mdgat_knn_workspace_bytes(B, C, N, M) bytes (16-byte aligned; the N x M matrix of 2 x.s - |s|^2); other channel
 * counts - or workspace == NULL - compute -sum (x - s)^2 inside the selection kernel and need none. */
int mdgat_knn(int B, int C, int N, int M, int k, const float* x, const float* src, int64_t* idx,
              int64_t* adj, void* workspace, size_t workspace_bytes, void* stream);
size_t mdgat_knn_workspace_bytes(int B, int C, int N, int M);

/* ---- the steps either side of the matcher, fp64 like the reference's numpy ---------- */

/* Pose from matches, per pair: solve_icp (utils/utils_test.py:73-110: centroids, cross covariance, SVD,
 * R = U V^T without reflection fix) on {kpts0[i], kpts1[matches0[i]] : matches0[i] >= 0}, as test.py:213-216
 * selects them; T [B][4][4] maps frame-1 points onto frame 0.  stats [B][5] = number of matches, inliers
 * (|T p1 - p0| < inlier_dist, utils_test.py:55-63), inlier ratio, and - when T_gt [B][4][4] is given -
 * translation / rotation error of inv(T) T_gt (utils_test.py:65-70; the arccos is not clamped there either). */
int mdgat_pose(int B, int N, int M, const float* kpts0, const float* kpts1, const int64_t* matches0,
               const double* T_gt, double inlier_dist, double* T, double* stats, void* stream);

/* Ground-truth matches of SparseDataset.__getitem__ (load_data.py:238-285): nearest neighbours of the
 * world-frame keypoints (T0/T1 [B][4][4] = pose . T_cam0_velo per frame, NULL = identity) in both
 * directions under `threshold`, optional mutual check; -1 = no match; rep [B] = number of frame-0
 * keypoints with a frame-1 keypoint within the threshold (load_data.py:264). */
int mdgat_gt_matches(int B, int N, int M, const float* kpts0, const float* kpts1, const double* T0, const double* T1,
                     double threshold, int mutual, int64_t* gt0, int64_t* gt1, int64_t* rep, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDGAT_HIP_H */
