// fp64 kernels of the reference-exact mode (f64.hip): declarations shared with api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// C[M][N] = act(A W^T + bias) (+ R); the K input columns come from A0 (columns [0, K0)) and A1 (columns [K0, K)).  No alignment
// beyond 8 bytes is assumed of any operand (the FPFH rows are 33 doubles).
struct GemmF64Args {
    const double* A0; int lda0; int K0;
    const double* A1; int lda1;
    const double* W; int ldw;          // [N][K]
    const double* bias;                // [N] or nullptr
    const double* R; int ldr;          // residual [M][N] or nullptr (may alias C)
    double* C; int ldc;
    int M, N, K;
    int relu;
    unsigned* guard;                   // host-mapped status word (MDGAT_STATUS_RANGE) or nullptr: raised when an output is not finite or
                                       // beyond 2^500 in magnitude (f64_out_of_range in f64.hip)
    double scale = 1.0;                // C = act(scale A W^T + bias) (+ R)   (the score matrix: 1 / sqrt(128), mdgat.py:431)
    int batch = 1;                     // independent products; A0, W, C of product z at + z sA, + z sW, + z sC (no second source, bias, residual shared)
    long long sA = 0, sW = 0, sC = 0;
};
int launch_gemm_f64(const GemmF64Args& a, hipStream_t s);

struct AttnF64Args {
    const double* qkv;     // [B][P][384]: q | k | v, each [4 heads][32 dims] (the rows of pack.py's qkv_w)
    double* msg;           // [B][P][128]: channel = head * 32 + dim
    int N, M, cross, topk;
    float zq;              // standard-normal quantile of the top-k fraction (first probe of the threshold search)
    uint32_t* sel;         // parity tap (mdgat_taps.topk_sel layout) or nullptr
    int selW;
    int units, tiles;      // B * 2 * 4 (pair, frame, head) units; query tiles per unit
    int hist_ints;         // dynamic attention: ints of LDS for the radix-select histograms (attn_hist_ints in f64.hip)
    unsigned* guard;       // as GemmF64Args::guard, for the message rows
};
// attention (topk == 0) / dynamic_attention (mdgat.py:190-210) on fp64 q / k / v; sel: optional tap of the kept keys
// mdgat_set_f64_attention_form / MDGAT_F64_ATTENTION_FORM: -1 full attention by launch size; 0 always the split-key form (results do not
// depend on the batch a pair travels in); 1 the big-launch form at every size
int f64_attention_form();
int launch_attention_f64(int B, int N, int M, int cross, int topk, const double* qkv, double* msg, uint32_t* sel, hipStream_t s, unsigned* guard = nullptr);
// in4 [R][4] = x y z saliency, in33 [R][33] = FPFH; rows pair-major, frame 0 then frame 1
int launch_assemble_f64(int B, int N, int M, const double* kpts0, const double* sigma0, const double* fpfh0, const double* kpts1,
                        const double* sigma1, const double* fpfh1, double* in4, double* in33, unsigned* guard, hipStream_t s);
// the same from raw float32 records [B][N][37] (load_data.py:146-165; FPFH normalised as numpy does it in float32, 290-292)
int launch_assemble_frames_f64(int B, int N, int M, const float* rec0, const float* rec1, int normalize, double* in4, double* in33, unsigned* guard,
                               hipStream_t s);
// guard (all three): host-mapped status word raised when a value is not finite (tested by its bits), or nullptr
int launch_f64_to_f32(const double* in, float* out, size_t n, unsigned* guard, hipStream_t s);

// ---- layer_f64.hip: the tail of a propagation layer (mlp.0 + ReLU, mlp.3 + residual, the next layer's q | k | v) as ONE launch ----
struct LayerF64Args {
    double* x;                 // [R][128] residual stream, updated in place
    const double* msg;         // [R][128] the layer's message (attention output; merge is folded into w1)
    const double *w1f, *b1;    // mlp.0 [256][256] in fragment order (launch_frag64), bias [256]
    const double *w2f, *b2;    // mlp.3 [128][256]
    const double *w3f, *b3;    // the NEXT layer's q | k | v projection [384][128], or nullptr (last fp64 layer)
    double* qkv;               // [R][384] (w3f != nullptr)
    float* x32;                // optional: the new x rounded to fp32 as well (the hand-over to the fp32-class layers)
    int R;
    unsigned* guard;           // as GemmF64Args::guard
    double* hid = nullptr;     // optional scratch [R][256]: with it, launches of few 16-row blocks may run the clustered kernel (layer_f64.hip)
};
int launch_layer_tail_f64(const LayerF64Args& a, hipStream_t s);
// both encoders, their sum and layer 0's q | k | v as one launch (mdgat.py:184-188, 152-155, 392-393, 227-232); weights in fragment order
struct EncoderF64Args {
    const double *in4, *in33;                  // [R][4] x y z saliency, [R][33] FPFH (launch_assemble_f64)
    const double *wk0, *bk0, *wd0, *bd0;       // kenc.0 [32][4], denc.0 [64][33] (BN folded)
    const double *wk1, *bk1, *wk2, *bk2;       // kenc.3 [64][32], kenc.6 [128][64]
    const double *wd1, *bd1;                   // denc.3 [128][64]
    const double *wl, *bl;                     // the encoders' last layers as one product over [hd2 ; hk3]: [128][256]
    const double *wq, *bq;                     // layer 0's q | k | v [384][128], or nullptr (no fp64 layer follows)
    double* x; double* qkv;                    // [R][128], [R][384]
    float* x32;                                // optional fp32 rounding of x (the hand-over when no fp64 layer follows)
    int R;
    unsigned* guard;
};
int launch_encoder_f64(const EncoderF64Args& a, hipStream_t s);
size_t encoder_f64_frag_doubles();
size_t frag64_doubles(int N, int K);
// W [N][K] row-major -> the fragment order the kernels of layer_f64.hip load ([N / 16][ceil(K / 8)][64 lanes][2], zero beyond K); N % 16 == 0
int launch_frag64(const double* W, double* out, int N, int K, hipStream_t s);
size_t layer_f64_frag_doubles();      // per layer: mlp.0 | mlp.3 | q|k|v
bool layer_f64_fused();               // mdgat_set_f64_layer_fusion / MDGAT_F64_LAYER_FUSION
