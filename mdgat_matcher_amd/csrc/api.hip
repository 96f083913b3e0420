// C ABI of libmdgat_hip.so (see include/mdgat_hip.h) and the launch sequence of one forward.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.hpp"
#include "f64.hpp"
#include "sinkhorn_f64.hpp"
#include "coop_chain.hpp"

// ---------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void mdgat_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mdgat_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return MDGAT_OK;
    mdgat_set_error("%s: %s", what, hipGetErrorString(e));
    return MDGAT_ERR_HIP;
}

extern "C" const char* mdgat_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------- blob layout
BlobLayout mdgat_blob_layout(int L) {
    BlobLayout b{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~size_t(3); return r; };
    b.kenc0_w = take(32 * 4);    b.kenc0_b = take(32);
    b.denc0_w = take(64 * 33);   b.denc0_b = take(64);
    b.kenc1_w = take(64 * 32);   b.kenc1_b = take(64);
    b.kenc2_w = take(128 * 64);  b.kenc2_b = take(128);
    b.denc1_w = take(128 * 64);  b.denc1_b = take(128);
    b.encl_w = take(128 * 256);  b.encl_b = take(128);
    b.layer0 = o;
    size_t lo = 0;
    auto ltake = [&](size_t n) { size_t r = lo; lo += (n + 3) & ~size_t(3); return r; };
    b.qkv_w = ltake(384 * 128);  b.qkv_b = ltake(384);
    b.mlp1_w = ltake(256 * 256); b.mlp1_b = ltake(256);
    b.mlp2_w = ltake(128 * 256); b.mlp2_b = ltake(128);
    b.layer_stride = lo;
    o += lo * (size_t)(2 * L);
    b.final_w = take(128 * 128); b.final_b = take(128);
    b.bin_score = take(1);
    b.total = o;
    return b;
}

extern "C" size_t mdgat_blob_floats(int L) { return mdgat_blob_layout(L).total; }

// ---------------------------------------------------------------------------------- handle
struct mdgat_handle {
    mdgat_config cfg;
    int device;
    BlobLayout bl;
    float* weights;      // device, fp32 blob (pack.py layout)
    _Float16* wsplit;    // device, split-f16 copies of the GNN / final_proj matrices (layer.hip)
    double* weights64;   // device, the blob in fp64 (MDGAT_ARITH_FP64: f64.hip), else nullptr
    double* wfrag64;     // device, per layer mlp.0 | mlp.3 | q|k|v once more in MFMA fragment order (layer_f64.hip), else nullptr
    bool loaded, loaded64;
    unsigned* host_error; // MDGAT_STATUS_WORDS host-mapped words the kernels set (common.hpp): Sinkhorn fallback taken, f16 range guard,
                          // token of the last forward that matched a frame-0 keypoint
    unsigned match_token; // the running forward's token (a new one per mdgat_forward / mdgat_forward_frames call, never 0)
    // Two lanes (forward_batched): the second lane's stream and the events that fork it off the caller's stream and join it again
    int lanes;            // 1 or 2 (mdgat_set_lanes; default 2, MDGAT_FORWARD_LANES)
    hipStream_t lane_stream;
    hipEvent_t ev_fork, ev_join, ev_mid;     // ev_mid: staggered lanes (forward_batched)
    // optional per-kernel-class timing of mdgat_forward (mdgat_profile): HIP events on the launch stream of each lane
    bool prof_on;
    struct ProfLane { std::vector<hipEvent_t> ev; std::vector<int> cls; size_t n = 0; } prof[2];
    double prof_ms[MDGAT_PROF_CLASSES];
    long long prof_launches[MDGAT_PROF_CLASSES];
    // One forward at a time per handle: the fork / join events, the second lane's stream and the profiling event lists are
    // per-handle state.  Two host threads calling mdgat_forward on one handle from two streams are serialised HERE (the enqueue
    // only - microseconds; the device work of the two calls still overlaps as far as their streams allow), so that one call's
    // lane can never fork off the other call's event record.  (The Python wrapper holds its own lock as well.)
    std::mutex enqueue;
};

// split-weight buffer: per layer the LDS images of layer.hip [w1 256 rows x 528 | w2 128 x 528 | qkv 384 x 272]
// (row = hi plane | lo plane | 32 B pad, zero filled; output rows in the P/Q order of layer.hip except the v rows), then final_proj 128 x 272, then the encoder matrices.
// The stage copies of layer.hip read whole KB: up to 512 B past a K = 256 block, hence the slack after final_proj.
static constexpr size_t WS_ROW256 = 528, WS_ROW128 = 272;
static constexpr size_t WS_W1 = 0, WS_W2 = 256 * WS_ROW256, WS_QKV = WS_W2 + 128 * WS_ROW256, WS_LAYER = WS_QKV + 384 * WS_ROW128;
static constexpr size_t WS_FINAL = 128 * WS_ROW128 + 512;
static size_t wsplit_halves(int L) { return WS_LAYER * (size_t)(2 * L) + WS_FINAL + MDGAT_ENC_SPLIT_HALVES; }
// behind them, the same GNN / final_proj matrices once more in FRAGMENT order for layer_split.hip, whose waves load their
// slice of the weights straight into registers: [row block of 16][k-step of 32][plane hi / lo][lane (row l15, column g)] x 16 B,
// so that a wave's load instruction reads one contiguous KB (launch_frag_image; no pads)
static constexpr size_t WF_W1 = 0, WF_W2 = 256 * 512, WF_QKV = WF_W2 + 128 * 512, WF_LAYER = WF_QKV + 384 * 256, WF_FINAL = 128 * 256;
static size_t wfrag_halves(int L) { return WF_LAYER * (size_t)(2 * L) + WF_FINAL; }
// fp64 fragment copies (layer_f64.hip: launch_frag64), per layer
static constexpr size_t WF64_W1 = 0, WF64_W2 = 256 * 256, WF64_QKV = WF64_W2 + 128 * 256;

extern "C" int mdgat_create(const mdgat_config* cfg, int device, mdgat_handle** out) {
    if (!cfg || !out) { mdgat_set_error("mdgat_create: null argument"); return MDGAT_ERR_BAD_ARG; }
    if (cfg->L < 0 || 2 * cfg->L > MDGAT_MAX_LAYERS) { mdgat_set_error("mdgat_create: L=%d out of range", cfg->L); return MDGAT_ERR_BAD_ARG; }
    if (cfg->attention_mode != MDGAT_ATTENTION_FP32 && cfg->attention_mode != MDGAT_ATTENTION_F16) { mdgat_set_error("mdgat_create: bad attention_mode %d", cfg->attention_mode); return MDGAT_ERR_BAD_ARG; }
    if (cfg->arithmetic != MDGAT_ARITH_FP32 && cfg->arithmetic != MDGAT_ARITH_FP64) { mdgat_set_error("mdgat_create: bad arithmetic %d", cfg->arithmetic); return MDGAT_ERR_BAD_ARG; }
    if (cfg->arithmetic == MDGAT_ARITH_FP64 && cfg->attention_mode != MDGAT_ATTENTION_FP32) { mdgat_set_error("mdgat_create: MDGAT_ARITH_FP64 and MDGAT_ATTENTION_F16 exclude each other"); return MDGAT_ERR_BAD_ARG; }
    if (cfg->arithmetic == MDGAT_ARITH_FP64 && cfg->f64_layers > 2 * cfg->L) { mdgat_set_error("mdgat_create: f64_layers=%d > 2L", cfg->f64_layers); return MDGAT_ERR_BAD_ARG; }
    if (cfg->extract_mode < 0 || cfg->extract_mode > 3) { mdgat_set_error("mdgat_create: bad extract_mode %d", cfg->extract_mode); return MDGAT_ERR_BAD_ARG; }
    for (int i = 0; i < 2 * cfg->L; ++i)
        if (cfg->topk[i] < 0) { mdgat_set_error("mdgat_create: topk[%d] < 0", i); return MDGAT_ERR_BAD_ARG; }
    hipDeviceProp_t prop;
    if (int rc = mdgat_check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return rc;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mdgat_set_error("mdgat_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return MDGAT_ERR_UNSUPPORTED;
    }
    mdgat_handle* h = new (std::nothrow) mdgat_handle();
    if (!h) { mdgat_set_error("mdgat_create: out of host memory"); return MDGAT_ERR_HIP; }
    h->cfg = *cfg;
    h->device = device;
    h->bl = mdgat_blob_layout(cfg->L);
    h->weights = nullptr;
    h->wsplit = nullptr;
    h->weights64 = nullptr;
    h->wfrag64 = nullptr;
    h->loaded = false;
    h->loaded64 = false;
    h->host_error = nullptr;
    h->match_token = 0;
    h->prof_on = false;
    h->lane_stream = nullptr;
    h->ev_fork = h->ev_join = h->ev_mid = nullptr;
    {
        const char* e = getenv("MDGAT_FORWARD_LANES");
        h->lanes = (e && atoi(e) == 1) ? 1 : 2;
    }
    for (int c = 0; c < MDGAT_PROF_CLASSES; ++c) { h->prof_ms[c] = 0.0; h->prof_launches[c] = 0; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = mdgat_check_hip(hipSetDevice(device), "hipSetDevice");
    if (!rc) rc = mdgat_check_hip(hipMalloc(&h->weights, h->bl.total * sizeof(float)), "hipMalloc(weights)");
    if (!rc && cfg->arithmetic == MDGAT_ARITH_FP64) rc = mdgat_check_hip(hipMalloc(&h->weights64, h->bl.total * sizeof(double)), "hipMalloc(fp64 weights)");
    if (!rc && cfg->arithmetic == MDGAT_ARITH_FP64)
        rc = mdgat_check_hip(hipMalloc(&h->wfrag64, (layer_f64_frag_doubles() * (size_t)(2 * cfg->L) + encoder_f64_frag_doubles()) * sizeof(double)),
                             "hipMalloc(fp64 weight fragments)");
    if (!rc) rc = mdgat_check_hip(hipMalloc(&h->wsplit, (wsplit_halves(cfg->L) + wfrag_halves(cfg->L)) * sizeof(_Float16)), "hipMalloc(split weights)");
    if (!rc) rc = mdgat_check_hip(hipMemset(h->wsplit, 0, (wsplit_halves(cfg->L) + wfrag_halves(cfg->L)) * sizeof(_Float16)), "hipMemset(split weights)");
    if (!rc) rc = mdgat_check_hip(hipHostMalloc(reinterpret_cast<void**>(&h->host_error), MDGAT_STATUS_WORDS * sizeof(unsigned), hipHostMallocMapped), "hipHostMalloc(status words)");
    if (!rc) for (int i = 0; i < MDGAT_STATUS_WORDS; ++i) h->host_error[i] = 0;
    if (!rc) rc = mdgat_check_hip(hipStreamCreateWithFlags(&h->lane_stream, hipStreamNonBlocking), "hipStreamCreate(lane)");
    if (!rc) rc = mdgat_check_hip(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming), "hipEventCreate");
    if (!rc) rc = mdgat_check_hip(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming), "hipEventCreate");
    if (!rc) rc = mdgat_check_hip(hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming), "hipEventCreate");
    (void)hipSetDevice(prev);
    if (rc) {
        mdgat_destroy(h);
        return rc;
    }
    *out = h;
    return MDGAT_OK;
}

// largest integer image of |w| over the blob (NaN / inf on top): the weights become f16 head + f16 residual like every operand
__global__ __launch_bounds__(256) void blob_absmax_kernel(const float* w, size_t n, unsigned* out) {
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, __builtin_bit_cast(unsigned, w[i]) & 0x7fffffffu);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

extern "C" int mdgat_load_weights(mdgat_handle* h, const float* blob, size_t n_floats, int on_device) {
    if (!h || !blob) { mdgat_set_error("mdgat_load_weights: null argument"); return MDGAT_ERR_BAD_ARG; }
    if (n_floats != h->bl.total) {
        mdgat_set_error("mdgat_load_weights: blob has %zu floats, expected %zu for L=%d", n_floats, h->bl.total, h->cfg.L);
        return MDGAT_ERR_BAD_ARG;
    }
    if (blob != h->weights) {
        if (int rc = mdgat_check_hip(hipMemcpy(h->weights, blob, n_floats * sizeof(float),
                                               on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice),
                                     "hipMemcpy(weights)"))
            return rc;
    }
    // split-f16 copies of the matrices the MFMA kernels consume (synchronous, like the copy above)
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = mdgat_check_hip(hipSetDevice(h->device), "hipSetDevice");
    const BlobLayout& bl = h->bl;
    h->loaded = false;
    if (!rc) {
        // f16 operand range of the WEIGHTS (the kernels guard activations): a weight beyond 6e4 would become an infinite f16 head,
        // and what that makes of the activations a ReLU can turn back into finite garbage (max(NaN, 0) = 0).  The packer brings
        // every rescalable channel to unit scale (pack.py: gauge fixing), so this only fires for checkpoints that are broken or
        // hold non-finite values.
        unsigned* dmax = reinterpret_cast<unsigned*>(h->wsplit);       // (scratch: the split images are written below)
        rc = mdgat_check_hip(hipMemset(dmax, 0, sizeof(unsigned)), "hipMemset(weight range)");
        unsigned hmax = 0;
        if (!rc) {
            hipLaunchKernelGGL(blob_absmax_kernel, dim3(256), dim3(256), 0, nullptr, h->weights, n_floats, dmax);
            rc = mdgat_check_hip(hipMemcpy(&hmax, dmax, sizeof(unsigned), hipMemcpyDeviceToHost), "hipMemcpy(weight range)");
        }
        if (!rc && hmax >= __builtin_bit_cast(unsigned, MDGAT_F16_GUARD)) {
            (void)hipSetDevice(prev);
            mdgat_set_error("mdgat_load_weights: a packed weight is not finite or beyond the f16 operand range (|w| >= 6e4; largest image 0x%08x): "
                            "this checkpoint does not fit the split-f16 arithmetic", hmax);
            return MDGAT_ERR_UNSUPPORTED;
        }
        if (!rc) rc = mdgat_check_hip(hipMemset(dmax, 0, sizeof(unsigned)), "hipMemset(weight range)");
    }
    for (int i = 0; i < 2 * h->cfg.L && !rc; ++i) {
        const float* lw = h->weights + bl.layer0 + (size_t)i * bl.layer_stride;
        _Float16* ls = h->wsplit + WS_LAYER * (size_t)i;
        rc = launch_split_rows(lw + bl.mlp1_w, ls + WS_W1, 256, 256, WS_ROW256, 256, nullptr);
        if (!rc) rc = launch_split_rows(lw + bl.mlp2_w, ls + WS_W2, 128, 256, WS_ROW256, 128, nullptr);
        if (!rc) rc = launch_split_rows(lw + bl.qkv_w, ls + WS_QKV, 384, 128, WS_ROW128, 256, nullptr);
        _Float16* lf = h->wsplit + wsplit_halves(h->cfg.L) + WF_LAYER * (size_t)i;
        if (!rc) rc = launch_frag_image(ls + WS_W1, lf + WF_W1, 256, 256, WS_ROW256, nullptr);
        if (!rc) rc = launch_frag_image(ls + WS_W2, lf + WF_W2, 128, 256, WS_ROW256, nullptr);
        if (!rc) rc = launch_frag_image(ls + WS_QKV, lf + WF_QKV, 384, 128, WS_ROW128, nullptr);
    }
    if (!rc) rc = launch_split_rows(h->weights + bl.final_w, h->wsplit + WS_LAYER * (size_t)(2 * h->cfg.L), 128, 128, WS_ROW128, 128, nullptr);
    if (!rc) rc = launch_frag_image(h->wsplit + WS_LAYER * (size_t)(2 * h->cfg.L), h->wsplit + wsplit_halves(h->cfg.L) + WF_LAYER * (size_t)(2 * h->cfg.L),
                                    128, 128, WS_ROW128, nullptr);
    {
        _Float16* es = h->wsplit + WS_LAYER * (size_t)(2 * h->cfg.L) + WS_FINAL;
        if (!rc) rc = launch_split_rows(h->weights + bl.kenc1_w, es, 64, 32, 64, 0, nullptr);
        if (!rc) rc = launch_split_rows(h->weights + bl.kenc2_w, es + 64 * 64, 128, 64, 128, 0, nullptr);
        if (!rc) rc = launch_split_rows_pad(h->weights + bl.denc0_w, es + 64 * 64 + 128 * 128, 64, 33, 48, nullptr);
        if (!rc) rc = launch_split_rows(h->weights + bl.denc1_w, es + 64 * 64 + 128 * 128 + 64 * 96, 128, 64, 128, 0, nullptr);
        if (!rc) rc = launch_split_rows(h->weights + bl.encl_w, es + 64 * 64 + 128 * 128 + 64 * 96 + 128 * 128, 128, 256, 512, 0, nullptr);
    }
    if (!rc) rc = mdgat_check_hip(hipDeviceSynchronize(), "split weights");
    (void)hipSetDevice(prev);
    if (rc) return rc;
    h->loaded = true;
    return MDGAT_OK;
}

extern "C" float* mdgat_weights_device_ptr(mdgat_handle* h) { return h ? h->weights : nullptr; }
extern "C" double* mdgat_weights_f64_device_ptr(mdgat_handle* h) { return h ? h->weights64 : nullptr; }

extern "C" int mdgat_load_weights_f64(mdgat_handle* h, const double* blob, size_t n_doubles, int on_device) {
    if (!h || !blob) { mdgat_set_error("mdgat_load_weights_f64: null argument"); return MDGAT_ERR_BAD_ARG; }
    if (!h->weights64) { mdgat_set_error("mdgat_load_weights_f64: the handle was not created with MDGAT_ARITH_FP64"); return MDGAT_ERR_BAD_ARG; }
    if (n_doubles != h->bl.total) {
        mdgat_set_error("mdgat_load_weights_f64: blob has %zu doubles, expected %zu for L=%d", n_doubles, h->bl.total, h->cfg.L);
        return MDGAT_ERR_BAD_ARG;
    }
    h->loaded64 = false;
    if (blob != h->weights64)
        if (int rc = mdgat_check_hip(hipMemcpy(h->weights64, blob, n_doubles * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice),
                                     "hipMemcpy(fp64 weights)"))
            return rc;
    // the layer-tail kernel's copies in fragment order (synchronous, like the copy above)
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = mdgat_check_hip(hipSetDevice(h->device), "hipSetDevice");
    const BlobLayout& bl = h->bl;
    for (int i = 0; i < 2 * h->cfg.L && !rc; ++i) {
        const double* lw = h->weights64 + bl.layer0 + (size_t)i * bl.layer_stride;
        double* lf = h->wfrag64 + layer_f64_frag_doubles() * (size_t)i;
        rc = launch_frag64(lw + bl.mlp1_w, lf + WF64_W1, 256, 256, nullptr);
        if (!rc) rc = launch_frag64(lw + bl.mlp2_w, lf + WF64_W2, 128, 256, nullptr);
        if (!rc) rc = launch_frag64(lw + bl.qkv_w, lf + WF64_QKV, 384, 128, nullptr);
    }
    {
        // the encoder matrices behind the layers': kenc.0 | denc.0 | kenc.3 | kenc.6 | denc.3 | last layers summed (EncFrag below)
        double* ef = h->wfrag64 + layer_f64_frag_doubles() * (size_t)(2 * h->cfg.L);
        const double* w = h->weights64;
        const struct { size_t src; int n, k; } enc[6] = {{bl.kenc0_w, 32, 4}, {bl.denc0_w, 64, 33}, {bl.kenc1_w, 64, 32}, {bl.kenc2_w, 128, 64},
                                                         {bl.denc1_w, 128, 64}, {bl.encl_w, 128, 256}};
        for (int j = 0; j < 6 && !rc; ++j) {
            rc = launch_frag64(w + enc[j].src, ef, enc[j].n, enc[j].k, nullptr);
            ef += frag64_doubles(enc[j].n, enc[j].k);
        }
    }
    if (!rc) rc = mdgat_check_hip(hipDeviceSynchronize(), "fp64 weight fragments");
    (void)hipSetDevice(prev);
    if (rc) return rc;
    h->loaded64 = true;
    return MDGAT_OK;
}

extern "C" void mdgat_destroy(mdgat_handle* h) {
    if (!h) return;
    if (h->weights) (void)hipFree(h->weights);
    if (h->wsplit) (void)hipFree(h->wsplit);
    if (h->weights64) (void)hipFree(h->weights64);
    if (h->wfrag64) (void)hipFree(h->wfrag64);
    if (h->host_error) (void)hipHostFree(h->host_error);
    if (h->lane_stream) { (void)hipStreamSynchronize(h->lane_stream); (void)hipStreamDestroy(h->lane_stream); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
    for (auto& pl : h->prof)
        for (hipEvent_t e : pl.ev) (void)hipEventDestroy(e);
    delete h;
}

// ---------------------------------------------------------------------------------- workspace
namespace {
struct Workspace {
    float *x, *qkv, *hid, *msg, *scores, *Z, *sk;
    double *x64, *qkv64, *hid64, *msg64;   // MDGAT_ARITH_FP64 only: the residual stream, q|k|v, hidden layer and message of the fp64 layers
    double* scores64;                      // ... the fp64 score matrix [B][N][M]: q | k | v's room where it fits, its own beyond (frames past ~770 keypoints)
    float* sk64; size_t sk64_bytes;        // ... and the workspace of the fp64 Sinkhorn + its arg-max arrays (0: the shape is beyond that kernel)
    _Float16* qkv16;
    size_t sk_bytes;
    size_t total;   // floats
};
Workspace carve(float* base, int B, int N, int M, bool f64) {
    const size_t R = (size_t)B * (N + M);
    Workspace w{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~size_t(63); return base ? base + r : nullptr; };
    w.x = take(R * 128);
    w.qkv = take(R * 384);   // qkv and hid are contiguous: the encoder uses them as one scratch area
    w.hid = take(R * 256);
    w.msg = take(R * 128);
    w.scores = take((size_t)B * N * M);
    w.Z = take((size_t)B * (N + 1) * (M + 1));
    w.sk_bytes = mdgat_sinkhorn_ws_bytes_impl(B, N, M);
    w.sk = take((w.sk_bytes + 3) / 4);
    w.qkv16 = reinterpret_cast<_Float16*>(take((mdgat_qkv16_halves(B, N, M) + 1) / 2));
    if (f64) {
        w.x64 = reinterpret_cast<double*>(take(R * 128 * 2));
        w.qkv64 = reinterpret_cast<double*>(take(R * 384 * 2));     // qkv64 and hid64 are contiguous: the encoder stages live there
        w.hid64 = reinterpret_cast<double*>(take(R * 256 * 2));
        w.msg64 = reinterpret_cast<double*>(take(R * 128 * 2));
        w.sk64_bytes = sinkhorn_f64_supported(N, M) ? sinkhorn_f64_workspace_bytes(B, N, M) + sinkhorn_f64_bests_bytes(B, N, M) : 0;
        w.sk64 = take((w.sk64_bytes + 3) / 4);
        w.scores64 = (size_t)N * M <= (size_t)384 * (N + M) || !w.sk64_bytes ? w.qkv64 : reinterpret_cast<double*>(take((size_t)B * N * M * 2));
    }
    w.total = o;
    return w;
}
}  // namespace


// ---------------------------------------------------------------------------------- forward
static GemmArgs pointwise(const float* A, int lda, int K, const float* W, const float* bias, int relu, float* C, int ldc,
                          int rows, int cout) {
    GemmArgs g{};
    g.A0 = A; g.lda0 = lda; g.K0 = K; g.A1 = nullptr; g.lda1 = 0;
    g.W = W; g.ldw = K; g.bias = bias; g.R = nullptr; g.ldr = 0;
    g.C = C; g.ldc = ldc; g.M = rows; g.N = cout; g.K = K; g.relu = relu; g.scale = 1.f;
    g.batch = 1; g.sA = g.sW = g.sC = 0;
    return g;
}

// inputs of a forward: six fp32 arrays, or raw 37-float frame records, or (MDGAT_ARITH_FP64) six fp64 arrays
struct FwdIn {
    const float *kpts0, *sigma0, *fpfh0, *kpts1, *sigma1, *fpfh1;
    const float *rec0, *rec1;
    int normalize_fpfh;
    const double *dk0, *ds0, *df0, *dk1, *ds1, *df1;
    // the same inputs from pair c on (slices of a batch)
    FwdIn from(size_t c, int N, int M) const {
        auto o = [](auto* q, size_t n) { return q ? q + n : q; };
        return FwdIn{o(kpts0, c * N * 3), o(sigma0, c * N), o(fpfh0, c * N * 33), o(kpts1, c * M * 3), o(sigma1, c * M), o(fpfh1, c * M * 33),
                     o(rec0, c * N * 37), o(rec1, c * M * 37), normalize_fpfh,
                     o(dk0, c * N * 3), o(ds0, c * N), o(df0, c * N * 33), o(dk1, c * M * 3), o(ds1, c * M), o(df1, c * M * 33)};
    }
};

// MDGAT_ARITH_FP64: the number of leading propagation layers that run in fp64 (mdgat_config.f64_layers)
static int f64_layer_count(const mdgat_config& cfg) {
    if (cfg.f64_layers > 0) return cfg.f64_layers;
    if (cfg.f64_layers < 0) return 0;       // MDGAT_F64_ENCODERS_ONLY
    int n = 0;
    for (int i = 0; i < 2 * cfg.L; ++i)
        if (cfg.topk[i] > 0) n = i + 1;
    return n;
}

static int forward_impl(mdgat_handle* h, int B, int N, int M, const FwdIn& in,
                        int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* Z,
                        const mdgat_taps* taps, void* workspace, size_t workspace_bytes, void* stream, int defer_alldust = 0, int lane = 0,
                        int mid_layer = -1) {
    const float *kpts0 = in.kpts0, *sigma0 = in.sigma0, *fpfh0 = in.fpfh0, *kpts1 = in.kpts1, *sigma1 = in.sigma1, *fpfh1 = in.fpfh1;
    const float *rec0 = in.rec0, *rec1 = in.rec1;
    const int normalize_fpfh = in.normalize_fpfh;
    if (!h) { mdgat_set_error("mdgat_forward: null handle"); return MDGAT_ERR_BAD_ARG; }
    // fp64 inputs, or raw float32 records on a handle that computes in fp64 (the loader's own sequence: float32 records, FPFH
    // normalised in float32, widened to double - load_data.py:146-165, 290-295)
    const bool f64 = in.dk0 != nullptr || (in.rec0 != nullptr && h->cfg.arithmetic == MDGAT_ARITH_FP64);
    if (!h->loaded) { mdgat_set_error("mdgat_forward: weights not loaded"); return MDGAT_ERR_NO_WEIGHTS; }
    if (f64 && (h->cfg.arithmetic != MDGAT_ARITH_FP64 || !h->loaded64)) {
        mdgat_set_error("mdgat_forward_f64: the handle needs MDGAT_ARITH_FP64 and mdgat_load_weights_f64");
        return h->cfg.arithmetic != MDGAT_ARITH_FP64 ? MDGAT_ERR_BAD_ARG : MDGAT_ERR_NO_WEIGHTS;
    }
    if (!f64 && h->cfg.arithmetic == MDGAT_ARITH_FP64) {
        mdgat_set_error("mdgat_forward: this handle computes in fp64 (MDGAT_ARITH_FP64): call mdgat_forward_f64 with fp64 inputs");
        return MDGAT_ERR_BAD_ARG;
    }
    if (static_cast<volatile unsigned*>(h->host_error)[MDGAT_STATUS_RANGE]) {
        // the forward is asynchronous: what an earlier launch found surfaces here unless the caller asked first
        // (mdgat_async_status after its own synchronisation - MDGAT.forward does)
        h->host_error[MDGAT_STATUS_RANGE] = 0;
        mdgat_set_error("mdgat_forward: a previous call on this handle met activations outside the f16 operand range (|v| >= 6e4; fp64 "
                        "layers of the exact mode: |v| >= 2^500) or non-finite values; its outputs are invalid");
        return MDGAT_ERR_UNSUPPORTED;
    }
    if (B <= 0 || N <= 0 || M <= 0) { mdgat_set_error("mdgat_forward: empty batch/keypoints (B=%d N=%d M=%d) must be handled by the caller", B, N, M); return MDGAT_ERR_BAD_ARG; }
    const bool arrays = kpts0 && sigma0 && fpfh0 && kpts1 && sigma1 && fpfh1;
    const bool arrays64 = in.dk0 && in.ds0 && in.df0 && in.dk1 && in.ds1 && in.df1;
    if ((!arrays && !(rec0 && rec1) && !arrays64) || !matches0 || !matches1 || !mscores0 || !mscores1 || !workspace) {
        mdgat_set_error("mdgat_forward: null pointer argument");
        return MDGAT_ERR_BAD_ARG;
    }
    const size_t need = carve(nullptr, B, N, M, h->cfg.arithmetic == MDGAT_ARITH_FP64).total * sizeof(float);
    if (workspace_bytes < need) { mdgat_set_error("mdgat_forward: workspace %zu < %zu bytes", workspace_bytes, need); return MDGAT_ERR_BAD_ARG; }
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) { mdgat_set_error("mdgat_forward: workspace must be 256-byte aligned"); return MDGAT_ERR_BAD_ARG; }
    const int L2 = 2 * h->cfg.L;
    for (int i = 0; i < L2; ++i) {
        const int k = h->cfg.topk[i];
        if (k > 0 && (k > N || k > M)) {   // torch.topk raises (mdgat.py:202)
            mdgat_set_error("layer %d: dynamic attention k=%d exceeds the number of keys (N=%d, M=%d)", i, k, N, M);
            return MDGAT_ERR_BAD_ARG;
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const BlobLayout& bl = h->bl;
    const float* w = h->weights;
    Workspace ws = carve(static_cast<float*>(workspace), B, N, M, h->cfg.arithmetic == MDGAT_ARITH_FP64);
    const int P = N + M;
    const int R = B * P;
    int rc;
    unsigned* status_dev = nullptr;
    if ((rc = mdgat_check_hip(hipHostGetDevicePointer(reinterpret_cast<void**>(&status_dev), h->host_error, 0), "hipHostGetDevicePointer"))) return rc;

    // profiling (off by default): an event after every launch on this lane's stream; the intervals are attributed to the
    // kernel classes after the whole batch has been enqueued (prof_collect), which then ends with a synchronisation
    mdgat_handle::ProfLane& pl = h->prof[lane];
    auto mark = [&](int cls) {
        if (!h->prof_on) return;
        if (pl.n == pl.ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            pl.ev.push_back(e);
        }
        (void)hipEventRecord(pl.ev[pl.n++], s);
        pl.cls.push_back(cls);
    };
    mark(-1);

    const Qkv16 q16 = mdgat_qkv16_carve(ws.qkv16, B, N, M);
    if ((N & 31) || (M & 31))   // the attention kernel reads V^T in whole 32-key blocks: pad columns must be zero
        if ((rc = mdgat_check_hip(hipMemsetAsync(q16.vt16, 0, (size_t)B * 256 * q16.PP * sizeof(_Float16), s), "memset(V^T pads)"))) return rc;
    int first = 0;              // the first propagation layer the fp32-class kernels run
    bool tail64 = false;        // MDGAT_ARITH_FP64: final_proj, scores, Sinkhorn and the extraction's arg-maxes in fp64 too
    if (!f64) {
        // ---- encoders (mdgat.py:392-393), one fused launch ----
        EncoderLaunch e{};
        e.kpts0 = kpts0; e.sigma0 = sigma0; e.fpfh0 = fpfh0; e.kpts1 = kpts1; e.sigma1 = sigma1; e.fpfh1 = fpfh1;
        e.rec0 = rec0; e.rec1 = rec1; e.normalize = normalize_fpfh;
        e.w = w; e.bl = &bl; e.es = h->wsplit + WS_LAYER * (size_t)L2 + WS_FINAL;
        e.x = ws.x; e.B = B; e.N = N; e.M = M;
        if ((rc = launch_encoder(e, s))) return rc;
        mark(MDGAT_PROF_ENCODER);
        if (taps && taps->x_enc)
            if ((rc = mdgat_check_hip(hipMemcpyAsync(taps->x_enc, ws.x, (size_t)R * 128 * sizeof(float), hipMemcpyDeviceToDevice, s), "tap x_enc"))) return rc;
    } else {
        // ---- MDGAT_ARITH_FP64 (f64.hip): encoders and the layers up to the last dynamic one in the reference's arithmetic ----
        const double* w64 = h->weights64;
        const size_t Rz = (size_t)R;
        // (the forward's launches of kernels whose workgroups wait for each other - clustered layer tails, the resident fp64 Sinkhorn - as
        // one group of the device's chain: coop_chain.hpp)
        CoopGroup coop_group(h->device, s, true);
        auto gemm = [&](const double* A0, int lda0, int K0, const double* A1, int lda1, size_t wofs, size_t bofs, int relu, const double* Rs, double* C, int ldc,
                        int cout, int K) {
            GemmF64Args g{A0, lda0, K0, A1, lda1, w64 + wofs, K, w64 + bofs, Rs, ldc, C, ldc, R, cout, K, relu, status_dev + MDGAT_STATUS_RANGE};
            return launch_gemm_f64(g, s);
        };
        // the assembled inputs at the END of the hidden area (the fused encoder writes layer 0's q | k | v while other workgroups still
        // read their inputs); the stages of the one-product-per-launch form in the (contiguous) q|k|v + hidden area in front of them:
        // 32 + 64 + 128 + 64 + 128 = 416 of the 603 doubles per point there
        double* in4 = ws.hid64 + Rz * (256 - 37);
        double* in33 = in4 + Rz * 4;
        double* hk1 = ws.qkv64;
        double* hk2 = hk1 + Rz * 32;
        double* hk3 = hk2 + Rz * 64;
        double* hd1 = hk3 + Rz * 128;
        double* hd2 = hd1 + Rz * 64;
        if (in.rec0) rc = launch_assemble_frames_f64(B, N, M, in.rec0, in.rec1, normalize_fpfh, in4, in33, status_dev + MDGAT_STATUS_RANGE, s);
        else rc = launch_assemble_f64(B, N, M, in.dk0, in.ds0, in.df0, in.dk1, in.ds1, in.df1, in4, in33, status_dev + MDGAT_STATUS_RANGE, s);
        if (rc) return rc;
        mark(MDGAT_PROF_F64_OTHER);
        first = f64_layer_count(h->cfg);
        // The TAIL in fp64 as well (mdgat_config.f64_sinkhorn; round 6): every layer, final_proj, the score matrix and the optimal
        // transport in the reference's own arithmetic, every arg-max of the extraction decided on the fp64 Z (sinkhorn_f64.hip).  With
        // the fp32-class tail Z is good to 7e-6 - inside the bar of 1e-4, but among 40 960 arg-maxes of a reference-held batch one had
        // its two candidates 1.3e-6 apart and fell the other way (profiles/NOTES_r6.md section 11).
        if (h->cfg.f64_sinkhorn > 0 && !ws.sk64_bytes) {
            mdgat_set_error("mdgat_forward_f64: f64_sinkhorn = 1 and %d x %d keypoints are beyond the fp64 Sinkhorn kernels (2175)", N, M);
            return MDGAT_ERR_UNSUPPORTED;
        }
        tail64 = h->cfg.f64_sinkhorn >= 0 && ws.sk64_bytes != 0 && h->cfg.f64_layers == 0;
        if (tail64) first = L2;
        // The tail of a layer - mlp.0 + ReLU, mlp.3 + residual (mdgat.py:246-248, 274) - and the NEXT layer's q | k | v projection
        // (227-232) run as one launch (layer_f64.hip), the hidden activation never leaving the chip, and so do the two encoders with
        // layer 0's projection; the last fp64 launch also writes the fp32 rounding of x, the hand-over.  mdgat_set_f64_layer_fusion(0)
        // keeps the one-product-per-launch form (bit-identical).
        const bool fused = layer_f64_fused() && h->wfrag64;
        bool handed_over = false;
        // KeypointEncoder (mdgat.py:184-188), DescriptorEncoder (152-155), their sum (392-393) as one product over [hd ; hk]
        if (fused) {
            const double* ef = h->wfrag64 + layer_f64_frag_doubles() * (size_t)L2;
            const double* f_k0 = ef;
            const double* f_d0 = f_k0 + frag64_doubles(32, 4);
            const double* f_k1 = f_d0 + frag64_doubles(64, 33);
            const double* f_k2 = f_k1 + frag64_doubles(64, 32);
            const double* f_d1 = f_k2 + frag64_doubles(128, 64);
            const double* f_l = f_d1 + frag64_doubles(128, 64);
            const EncoderF64Args e{in4, in33, f_k0, w64 + bl.kenc0_b, f_d0, w64 + bl.denc0_b, f_k1, w64 + bl.kenc1_b, f_k2, w64 + bl.kenc2_b,
                                   f_d1, w64 + bl.denc1_b, f_l, w64 + bl.encl_b,
                                   first > 0 ? h->wfrag64 + WF64_QKV : nullptr, first > 0 ? w64 + bl.layer0 + bl.qkv_b : nullptr,
                                   ws.x64, ws.qkv64,
                                   first == 0 ? ws.x : nullptr, R, status_dev + MDGAT_STATUS_RANGE};
            if ((rc = launch_encoder_f64(e, s))) return rc;
            handed_over = first == 0;
        } else {
            if ((rc = gemm(in4, 4, 4, nullptr, 0, bl.kenc0_w, bl.kenc0_b, 1, nullptr, hk1, 32, 32, 4))) return rc;
            if ((rc = gemm(hk1, 32, 32, nullptr, 0, bl.kenc1_w, bl.kenc1_b, 1, nullptr, hk2, 64, 64, 32))) return rc;
            if ((rc = gemm(hk2, 64, 64, nullptr, 0, bl.kenc2_w, bl.kenc2_b, 1, nullptr, hk3, 128, 128, 64))) return rc;
            if ((rc = gemm(in33, 33, 33, nullptr, 0, bl.denc0_w, bl.denc0_b, 1, nullptr, hd1, 64, 64, 33))) return rc;
            if ((rc = gemm(hd1, 64, 64, nullptr, 0, bl.denc1_w, bl.denc1_b, 1, nullptr, hd2, 128, 128, 64))) return rc;
            if ((rc = gemm(hd2, 128, 128, hk3, 128, bl.encl_w, bl.encl_b, 0, nullptr, ws.x64, 128, 128, 256))) return rc;
        }
        mark(MDGAT_PROF_F64_GEMM);
        if (taps && taps->x_enc)
            if ((rc = launch_f64_to_f32(ws.x64, taps->x_enc, Rz * 128, nullptr, s))) return rc;
        for (int i = 0; i < first; ++i) {
            const size_t lo = bl.layer0 + (size_t)i * bl.layer_stride;
            // MultiHeadedAttention (mdgat.py:223-237; merge is folded into mlp.0 by pack.py), attention / dynamic_attention (190-210)
            if (!fused) {
                if ((rc = gemm(ws.x64, 128, 128, nullptr, 0, lo + bl.qkv_w, lo + bl.qkv_b, 0, nullptr, ws.qkv64, 384, 384, 128))) return rc;
                mark(MDGAT_PROF_F64_GEMM);
            }
            uint32_t* sel = (taps && taps->topk_sel) ? taps->topk_sel + (size_t)i * mdgat_topk_sel_words(B, N, M) : nullptr;
            if ((rc = launch_attention_f64(B, N, M, i & 1, h->cfg.topk[i], ws.qkv64, ws.msg64, sel, s, status_dev + MDGAT_STATUS_RANGE))) return rc;
            mark(h->cfg.topk[i] > 0 ? MDGAT_PROF_F64_ATTENTION_TOPK : MDGAT_PROF_F64_ATTENTION_FULL);
            // AttentionalPropagation + residual (mdgat.py:246-248, 274)
            if (fused) {
                const double* lf = h->wfrag64 + layer_f64_frag_doubles() * (size_t)i;
                const bool last = i + 1 == first;
                const LayerF64Args t{ws.x64, ws.msg64, lf + WF64_W1, w64 + lo + bl.mlp1_b, lf + WF64_W2, w64 + lo + bl.mlp2_b,
                                     last ? nullptr : lf + layer_f64_frag_doubles() + WF64_QKV, last ? nullptr : w64 + lo + bl.layer_stride + bl.qkv_b,
                                     ws.qkv64, last ? ws.x : nullptr, R, status_dev + MDGAT_STATUS_RANGE, ws.hid64};
                if ((rc = launch_layer_tail_f64(t, s))) return rc;
                handed_over = last;
            } else {
                if ((rc = gemm(ws.x64, 128, 128, ws.msg64, 128, lo + bl.mlp1_w, lo + bl.mlp1_b, 1, nullptr, ws.hid64, 256, 256, 256))) return rc;
                if ((rc = gemm(ws.hid64, 256, 256, nullptr, 0, lo + bl.mlp2_w, lo + bl.mlp2_b, 0, ws.x64, ws.x64, 128, 128, 256))) return rc;
            }
            mark(MDGAT_PROF_F64_GEMM);
            if (taps && taps->x_layers)
                if ((rc = launch_f64_to_f32(ws.x64, taps->x_layers + (size_t)i * Rz * 128, Rz * 128, nullptr, s))) return rc;
        }
        if (tail64) {
            // final_proj (mdgat.py:397), the score matrix (430-431), the optimal transport (434-436) and the extraction (441-483)
            double* mdesc64 = ws.msg64;          // (the message and q | k | v of the last layer are dead)
            double* scores64 = ws.scores64;      // [B][N][M]: in q | k | v's room while N M <= 384 (N + M)
            if ((rc = gemm(ws.x64, 128, 128, nullptr, 0, bl.final_w, bl.final_b, 0, nullptr, mdesc64, 128, 128, 128))) return rc;
            if (taps && taps->mdesc)
                if ((rc = launch_f64_to_f32(mdesc64, taps->mdesc, Rz * 128, nullptr, s))) return rc;
            GemmF64Args sg{mdesc64, 128, 128, nullptr, 0, mdesc64 + (size_t)N * 128, 128, nullptr, nullptr, 0, scores64, M, N, M, 128, 0, status_dev + MDGAT_STATUS_RANGE};
            sg.scale = 0.08838834764831845;      // 1 / sqrt(128)
            sg.batch = B; sg.sA = sg.sW = (long long)(N + M) * 128; sg.sC = (long long)N * M;
            if ((rc = launch_gemm_f64(sg, s))) return rc;
            mark(MDGAT_PROF_F64_GEMM);
            if (taps && taps->scores)
                if ((rc = launch_f64_to_f32(scores64, taps->scores, (size_t)B * N * M, nullptr, s))) return rc;
            const size_t kb = sinkhorn_f64_workspace_bytes(B, N, M);
            char* bw = reinterpret_cast<char*>(ws.sk64) + kb;
            auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
            int* ri = reinterpret_cast<int*>(bw); bw += al((size_t)B * N * 4);
            float* rv = reinterpret_cast<float*>(bw); bw += al((size_t)B * N * 4);
            int* ci = reinterpret_cast<int*>(bw); bw += al((size_t)B * M * 4);
            float* cv = reinterpret_cast<float*>(bw);
            const SkExtract ex64{h->cfg.extract_mode, h->cfg.match_threshold, matches0, matches1, mscores0, mscores1, defer_alldust,
                                 status_dev + MDGAT_STATUS_MATCHED + (h->match_token % MDGAT_MATCH_SLOTS), h->match_token};
            if ((rc = launch_sinkhorn_f64(B, N, M, scores64, 0.0, h->cfg.sinkhorn_iters, nullptr, Z, h->cfg.extract_mode >= MDGAT_EXTRACT_THRESHOLD, ri, rv, ci, cv,
                                          ws.sk64, kb, status_dev + MDGAT_STATUS_RANGE, s, w64 + bl.bin_score))) return rc;
            if ((rc = launch_extract_from_bests(B, N, M, &ex64, ri, rv, ci, cv, s))) return rc;
            mark(MDGAT_PROF_SINKHORN);
            return MDGAT_OK;
        }
        // hand-over: nothing behind the last dynamic layer is discontinuous
        if (!handed_over) {
            if ((rc = launch_f64_to_f32(ws.x64, ws.x, Rz * 128, status_dev + MDGAT_STATUS_RANGE, s))) return rc;
            mark(MDGAT_PROF_F64_OTHER);
        }
    }

    // ---- 2L attentional propagation layers (mdgat.py:259-276) ----
    // launch i: [attention of layer i] -> [mlp + residual of layer i | q/k/v of layer i + 1 (or final_proj)]
    float* mdesc = ws.hid;
    const _Float16* wfinal = h->wsplit + WS_LAYER * (size_t)L2;
    const _Float16* wfrag = h->wsplit + wsplit_halves(h->cfg.L);      // fragment-order copies (layer_split.hip)
    const _Float16* ffinal = wfrag + WF_LAYER * (size_t)L2;
    {
        LayerLaunch p{};
        p.x = ws.x; p.R = R; p.N = N; p.M = M; p.out = q16; p.mdesc = mdesc; p.do_mlp = 0; p.guard = status_dev + MDGAT_STATUS_RANGE;
        if (first < L2) { p.mode3 = 1; p.w3s = h->wsplit + WS_LAYER * (size_t)first + WS_QKV; p.w3f = wfrag + WF_LAYER * (size_t)first + WF_QKV; p.b3 = w + bl.layer0 + (size_t)first * bl.layer_stride + bl.qkv_b; }
        else { p.mode3 = 2; p.w3s = wfinal; p.w3f = ffinal; p.b3 = w + bl.final_b; }
        if ((rc = launch_layer(p, s))) return rc;
        mark(MDGAT_PROF_LAYER_FIRST);
    }
    for (int i = first; i < L2; ++i) {
        const float* lw = w + bl.layer0 + (size_t)i * bl.layer_stride;
        const _Float16* ls = h->wsplit + WS_LAYER * (size_t)i;
        const int cross = i & 1;   // names = ['self', 'cross'] * L (mdgat.py:352-353)
        uint32_t* sel = (taps && taps->topk_sel) ? taps->topk_sel + (size_t)i * mdgat_topk_sel_words(B, N, M) : nullptr;
        const int kk = h->cfg.topk[i];
        if ((rc = launch_attention(B, N, M, cross, kk, q16, ws.msg, s, h->cfg.attention_mode, sel))) return rc;
        mark(kk > 0 ? MDGAT_PROF_ATTENTION_TOPK : MDGAT_PROF_ATTENTION_FULL);
        LayerLaunch p{};
        p.x = ws.x; p.msg = ws.msg; p.R = R; p.N = N; p.M = M; p.out = q16; p.mdesc = mdesc; p.do_mlp = 1; p.guard = status_dev + MDGAT_STATUS_RANGE;
        p.w1s = ls + WS_W1; p.b1 = lw + bl.mlp1_b; p.w2s = ls + WS_W2; p.b2 = lw + bl.mlp2_b;
        const _Float16* lf = wfrag + WF_LAYER * (size_t)i;
        p.w1f = lf + WF_W1; p.w2f = lf + WF_W2;
        if (i + 1 < L2) { p.mode3 = 1; p.w3s = ls + WS_LAYER + WS_QKV; p.w3f = lf + WF_LAYER + WF_QKV; p.b3 = lw + bl.layer_stride + bl.qkv_b; }
        else { p.mode3 = 2; p.w3s = wfinal; p.w3f = ffinal; p.b3 = w + bl.final_b; }
        if ((rc = launch_layer(p, s))) return rc;
        mark(i + 1 < L2 ? MDGAT_PROF_LAYER : MDGAT_PROF_LAYER_LAST);
        if (i == mid_layer) (void)hipEventRecord(h->ev_mid, s);      // staggered lanes: the other lane starts here
        if (taps && taps->x_layers)
            if ((rc = mdgat_check_hip(hipMemcpyAsync(taps->x_layers + (size_t)i * R * 128, ws.x, (size_t)R * 128 * sizeof(float), hipMemcpyDeviceToDevice, s), "tap x_layers"))) return rc;
    }

    // ---- final projection (mdgat.py:397, computed by the last launch above) and score matrix (430-431) ----
    if (taps && taps->mdesc)
        if ((rc = mdgat_check_hip(hipMemcpyAsync(taps->mdesc, mdesc, (size_t)R * 128 * sizeof(float), hipMemcpyDeviceToDevice, s), "tap mdesc"))) return rc;
    // (the score kernel also clears the exchange slots of the Sinkhorn kernel that follows: no memset launch in between)
    const size_t sk_clear = ws.sk_bytes ? sinkhorn_slots_clear_bytes(B, N, M) : 0;
    if ((rc = launch_scores(B, N, M, mdesc, ws.scores, 0.08838834764831845f /* 1 / sqrt(128) */, s, sk_clear ? ws.sk : nullptr, sk_clear,
                            status_dev + MDGAT_STATUS_RANGE))) return rc;
    mark(MDGAT_PROF_SCORES);
    if (taps && taps->scores)
        if ((rc = mdgat_check_hip(hipMemcpyAsync(taps->scores, ws.scores, (size_t)B * N * M * sizeof(float), hipMemcpyDeviceToDevice, s), "tap scores"))) return rc;

    // ---- optimal transport (mdgat.py:434-436) and match extraction (441-483) ----
    // (Z is only materialised when the caller asks for it or the streaming Sinkhorn needs it for the extraction)
    const bool fused = ws.sk_bytes != 0;   // N, M <= 2048: the cluster kernel, arg-maxes fused
    float* Zout = Z ? Z : (fused ? nullptr : ws.Z);
    const SkExtract ex{h->cfg.extract_mode, h->cfg.match_threshold, matches0, matches1, mscores0, mscores1, defer_alldust,
                       status_dev + MDGAT_STATUS_MATCHED + (h->match_token % MDGAT_MATCH_SLOTS), h->match_token};
    if ((rc = launch_sinkhorn(B, N, M, ws.scores, w + bl.bin_score, 0.f, h->cfg.sinkhorn_iters, Zout, ws.sk, ws.sk_bytes, &ex, s, status_dev,
                              Z ? Z : ws.Z, sk_clear != 0))) return rc;
    mark(MDGAT_PROF_SINKHORN);
    return MDGAT_OK;
}

// profiling: wait for the events of both lanes and add the intervals between consecutive ones to their kernel classes
// (an interval that starts at a -1 mark - the beginning of a forward_impl call - is counted, one that ends there is not)
static int prof_collect(mdgat_handle* h) {
    if (!h->prof_on) return MDGAT_OK;
    for (auto& pl : h->prof) {
        if (pl.n > 1) {
            if (int rc = mdgat_check_hip(hipEventSynchronize(pl.ev[pl.n - 1]), "profile sync")) return rc;
            for (size_t i = 1; i < pl.n; ++i) {
                float ms = 0.f;
                if (pl.cls[i] >= 0 && hipEventElapsedTime(&ms, pl.ev[i - 1], pl.ev[i]) == hipSuccess) {
                    h->prof_ms[pl.cls[i]] += ms;
                    h->prof_launches[pl.cls[i]] += 1;
                }
            }
        }
        pl.n = 0;
        pl.cls.clear();
    }
    return MDGAT_OK;
}

// Batches run in slices on two lanes.  Pairs are independent.  (i) A slice of ~64 pairs fills the part (512 tiles of the
// layer kernel, 2048 workgroups of the attention kernel, one Sinkhorn workgroup per CU at N = M = 512) while its working set
// (q / k / v: 1.6 MB per pair and layer) still fits the 256 MB Infinity Cache between the kernel that writes it and the one
// that reads it, which a batch of 128 no longer does (20 400 pairs/s at B = 64 against 19 100-19 500 at B = 128 ... 512).
// (ii) Round 3: the kernels of ONE forward run back to back with synchronised phases (every layer workgroup loads its tile
// at the same moment, every launch has a tail, the Sinkhorn kernel mostly waits for its partners); two half-size forwards
// on two streams fill each other's gaps: 2 x 32 pairs in flight 21 300 pairs/s against 20 500 for 1 x 64 on the same box
// (tools/overlap_streams.py; 2 x 20 against 1 x 40: 20 300 / 16 350 - a single launch of 1.25 tile rounds has a long tail).
// So: a batch of more than MDGAT_FORWARD_SLICE_POINTS (32 768) keypoints is cut into an EVEN number of balanced slices of at
// most that many, which alternate between the caller's stream and the handle's second stream (forked from and joined to the
// caller's stream by events: the call stays asynchronous and ordered on the caller's stream); each lane has its own half of
// the workspace.  The one batch-wide rule of the reference, mdgat.py:465-467, is applied over the whole batch afterwards.
// Taps (whole-batch layouts) run unsliced; mdgat_set_lanes(h, 1) / MDGAT_FORWARD_LANES=1 keeps everything on the caller's stream
// (slices of 65 536 keypoints beyond 1.5 x that).
struct LanePlan { int nslices, per, lanes; size_t lane_bytes; };
static LanePlan lane_plan(int lanes, int B, int N, int M, bool f64) {
    static const long env_points = [] { const char* e = getenv("MDGAT_FORWARD_SLICE_POINTS"); return e ? atol(e) : -1L; }();   // unset: defaults; 0: never slice
    LanePlan p{1, B, 1, 0};
    const long per_pair = (long)N + M;
    if (B <= 1 || per_pair <= 0 || env_points == 0) return p;
    const long total = (long)B * per_pair;
    if (lanes >= 2) {
        const long pts = env_points > 0 ? env_points : 32768L;
        if (total <= pts) return p;
        long n = 2 * ((total + 2 * pts - 1) / (2 * pts));
        if (n > B) n = B;
        p.nslices = (int)n;
        p.per = (int)((B + n - 1) / n);
        p.nslices = (B + p.per - 1) / p.per;
        p.lanes = p.nslices >= 2 ? 2 : 1;
    } else {
        const long pts = env_points > 0 ? env_points : 65536L;
        long slice = pts / per_pair;
        if (slice < 1) slice = 1;
        if ((long)B <= slice + slice / 2) return p;
        const long n = (B + slice - 1) / slice;
        p.per = (int)((B + n - 1) / n);
        p.nslices = (B + p.per - 1) / p.per;
    }
    p.lane_bytes = (carve(nullptr, p.per, N, M, f64).total * sizeof(float) + 255) & ~size_t(255);
    return p;
}

extern "C" size_t mdgat_workspace_bytes(const mdgat_handle* h, int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    // (taps run unsliced: the whole batch's workspace is the lower bound in every case)
    const bool f64 = h && h->cfg.arithmetic == MDGAT_ARITH_FP64;
    const size_t whole = carve(nullptr, B, N, M, f64).total * sizeof(float);
    const LanePlan p = lane_plan(h ? h->lanes : 2, B, N, M, f64);
    const size_t laned = p.lane_bytes * (size_t)p.lanes;
    return whole > laned ? whole : laned;
}

extern "C" int mdgat_set_lanes(mdgat_handle* h, int lanes) {
    if (!h || lanes < 1 || lanes > 2) { mdgat_set_error("mdgat_set_lanes: lanes must be 1 or 2"); return MDGAT_ERR_BAD_ARG; }
    h->lanes = lanes;
    return MDGAT_OK;
}

static int forward_batched(mdgat_handle* h, int B, int N, int M, const FwdIn& in,
                           int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* Z,
                           const mdgat_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
    LanePlan p{1, B, 1, 0};
    if (!h) { mdgat_set_error("mdgat_forward: null handle"); return MDGAT_ERR_BAD_ARG; }
    std::lock_guard<std::mutex> serialise(h->enqueue);
    if (++h->match_token == 0) h->match_token = 1;      // this call's token (mdgat_matched_any): every slice / lane of the call writes the same one
    if (!taps && B > 0 && N > 0 && M > 0 && matches0 && matches1 && mscores0 && mscores1) p = lane_plan(h->lanes, B, N, M, h->cfg.arithmetic == MDGAT_ARITH_FP64);
    if (p.nslices <= 1) {
        const int rc = forward_impl(h, B, N, M, in, matches0, matches1, mscores0, mscores1, Z, taps, workspace, workspace_bytes, stream);
        return rc ? rc : prof_collect(h);
    }
    if (!workspace || workspace_bytes < p.lane_bytes * (size_t)p.lanes) {
        mdgat_set_error("mdgat_forward: workspace %zu < %zu bytes", workspace_bytes, p.lane_bytes * (size_t)p.lanes);
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s0 = static_cast<hipStream_t>(stream);
    int rc = MDGAT_OK;
    if (p.lanes == 2) {
        // (both lanes start together: a second lane started one to five launches behind the first - complementary kernels
        // side by side - pays the delay as a tail: 20 900 -> 20 300 ... 19 700 pairs/s at B = 64)
        if ((rc = mdgat_check_hip(hipEventRecord(h->ev_fork, s0), "fork record"))) return rc;
        if ((rc = mdgat_check_hip(hipStreamWaitEvent(h->lane_stream, h->ev_fork, 0), "fork wait"))) return rc;
    }
    // Experiment (MDGAT_LANE_STAGGER = layer index, unset / negative: off): with four or more slices the second
    // lane starts when the first lane's first slice has passed that layer, so that the Sinkhorn launch of one lane runs next to
    // layer / attention launches of the other instead of next to the other lane's Sinkhorn.
    static const int stagger = [] { const char* e = getenv("MDGAT_LANE_STAGGER"); return e ? atoi(e) : -1; }();
    const int mid_layer = (p.lanes == 2 && p.nslices >= 4 && stagger >= 0 && stagger < 2 * h->cfg.L) ? stagger : -1;
    int slice = 0;
    for (int c = 0; c < B && !rc; c += p.per, ++slice) {
        const int b = B - c < p.per ? B - c : p.per;
        const size_t c_ = (size_t)c;
        const int lane = p.lanes == 2 ? (slice & 1) : 0;
        if (slice == 1 && mid_layer >= 0)
            if ((rc = mdgat_check_hip(hipStreamWaitEvent(h->lane_stream, h->ev_mid, 0), "stagger wait"))) break;
        rc = forward_impl(h, b, N, M, in.from(c_, N, M),
                          matches0 + c_ * N, matches1 + c_ * M, mscores0 + c_ * N, mscores1 + c_ * M,
                          Z ? Z + c_ * (N + 1) * (M + 1) : nullptr, nullptr, static_cast<char*>(workspace) + (size_t)lane * p.lane_bytes,
                          p.lane_bytes, lane ? static_cast<void*>(h->lane_stream) : stream, 1, lane, slice == 0 ? mid_layer : -1);
    }
    if (p.lanes == 2) {
        // (joined even after a failed launch: the caller's stream must not run ahead of what the second lane was given)
        const int rj = mdgat_check_hip(hipEventRecord(h->ev_join, h->lane_stream), "join record");
        const int rw = mdgat_check_hip(hipStreamWaitEvent(s0, h->ev_join, 0), "join wait");
        if (!rc) rc = rj ? rj : rw;
    }
    if (rc) return rc;
    if ((rc = launch_alldust_fixup(B, N, M, h->cfg.extract_mode, matches0, mscores1, s0))) return rc;
    return prof_collect(h);
}

extern "C" int mdgat_forward(mdgat_handle* h, int B, int N, int M, const float* kpts0, const float* sigma0,
                             const float* fpfh0, const float* kpts1, const float* sigma1, const float* fpfh1,
                             int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* Z,
                             const mdgat_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
    if (!kpts0 || !sigma0 || !fpfh0 || !kpts1 || !sigma1 || !fpfh1) { mdgat_set_error("mdgat_forward: null input pointer"); return MDGAT_ERR_BAD_ARG; }
    const FwdIn in{kpts0, sigma0, fpfh0, kpts1, sigma1, fpfh1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return forward_batched(h, B, N, M, in, matches0, matches1, mscores0, mscores1, Z, taps, workspace, workspace_bytes, stream);
}

extern "C" int mdgat_forward_f64(mdgat_handle* h, int B, int N, int M, const double* kpts0, const double* sigma0,
                                 const double* fpfh0, const double* kpts1, const double* sigma1, const double* fpfh1,
                                 int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* Z,
                                 const mdgat_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
    if (!kpts0 || !sigma0 || !fpfh0 || !kpts1 || !sigma1 || !fpfh1) { mdgat_set_error("mdgat_forward_f64: null input pointer"); return MDGAT_ERR_BAD_ARG; }
    const FwdIn in{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, kpts0, sigma0, fpfh0, kpts1, sigma1, fpfh1};
    return forward_batched(h, B, N, M, in, matches0, matches1, mscores0, mscores1, Z, taps, workspace, workspace_bytes, stream);
}

extern "C" int mdgat_forward_frames(mdgat_handle* h, int B, int N, int M, const float* frames0, const float* frames1,
                                    int normalize_fpfh, int64_t* matches0, int64_t* matches1, float* mscores0,
                                    float* mscores1, float* Z, const mdgat_taps* taps, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!frames0 || !frames1) { mdgat_set_error("mdgat_forward_frames: null frame pointer"); return MDGAT_ERR_BAD_ARG; }
    const FwdIn in{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, frames0, frames1, normalize_fpfh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return forward_batched(h, B, N, M, in, matches0, matches1, mscores0, mscores1, Z, taps, workspace, workspace_bytes, stream);
}

extern "C" int mdgat_async_status(mdgat_handle* h, int clear, unsigned* sinkhorn_fallback, unsigned* range_violation) {
    if (!h) { mdgat_set_error("mdgat_async_status: null handle"); return MDGAT_ERR_BAD_ARG; }
    volatile unsigned* st = h->host_error;
    const unsigned fb = st[MDGAT_STATUS_SK_FALLBACK], rg = st[MDGAT_STATUS_RANGE];
    if (sinkhorn_fallback) *sinkhorn_fallback = fb;
    if (range_violation) *range_violation = rg;
    if (clear) { st[MDGAT_STATUS_SK_FALLBACK] = 0; st[MDGAT_STATUS_RANGE] = 0; }
    if (rg) {
        mdgat_set_error("activations outside the f16 operand range (|v| >= 6e4; fp64 layers of the exact mode: |v| >= 2^500) or non-finite "
                        "values reached a kernel: the outputs of the calls since the last check are invalid (this checkpoint / input does "
                        "not fit the arithmetic)");
        return MDGAT_ERR_UNSUPPORTED;
    }
    return MDGAT_OK;
}

extern "C" unsigned mdgat_last_token(mdgat_handle* h) { return h ? h->match_token : 0u; }

extern "C" int mdgat_matched_any(mdgat_handle* h, unsigned token, unsigned* matched) {
    if (!h || !matched) { mdgat_set_error("mdgat_matched_any: null argument"); return MDGAT_ERR_BAD_ARG; }
    // the extraction kernels of the forward that carried `token` write it into the token's slot when a frame-0 keypoint is matched
    // (host-mapped words; a slot per call, so calls of other threads / streams on this handle in between do not disturb the answer)
    if (!token) token = h->match_token;
    *matched = static_cast<volatile unsigned*>(h->host_error)[MDGAT_STATUS_MATCHED + (token % MDGAT_MATCH_SLOTS)] == token ? 1u : 0u;
    return MDGAT_OK;
}

extern "C" int mdgat_profile(mdgat_handle* h, int enable, double* ms_out, long long* launches_out) {
    if (!h) { mdgat_set_error("mdgat_profile: null handle"); return MDGAT_ERR_BAD_ARG; }
    for (int c = 0; c < MDGAT_PROF_CLASSES; ++c) {
        if (ms_out) ms_out[c] = h->prof_ms[c];
        if (launches_out) launches_out[c] = h->prof_launches[c];
        h->prof_ms[c] = 0.0;
        h->prof_launches[c] = 0;
    }
    h->prof_on = enable != 0;
    return MDGAT_OK;
}

// ---------------------------------------------------------------------------------- per-op entry points
extern "C" size_t mdgat_sinkhorn_workspace_bytes(int B, int N, int M) { return mdgat_sinkhorn_ws_bytes_impl(B, N, M); }

extern "C" int mdgat_sinkhorn(int B, int N, int M, const float* scores, float bin_score, int iters, float* Z, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!scores || !Z) { mdgat_set_error("mdgat_sinkhorn: null pointer"); return MDGAT_ERR_BAD_ARG; }
    // without (enough, 256-byte aligned) workspace the streaming kernel is used instead of the cluster kernel
    return launch_sinkhorn(B, N, M, scores, nullptr, bin_score, iters, Z, workspace, workspace_bytes, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int mdgat_extract(int B, int N, int M, const float* Z, int mode, float match_threshold, int64_t* matches0,
                             int64_t* matches1, float* mscores0, float* mscores1, void* stream) {
    if (!Z || !matches0 || !matches1 || !mscores0 || !mscores1) { mdgat_set_error("mdgat_extract: null pointer"); return MDGAT_ERR_BAD_ARG; }
    return launch_extract(B, N, M, Z, mode, match_threshold, matches0, matches1, mscores0, mscores1, static_cast<hipStream_t>(stream));
}

extern "C" size_t mdgat_attention_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return mdgat_qkv16_halves(B, N, M) * sizeof(_Float16);
}

extern "C" int mdgat_attention(int B, int N, int M, int cross, int topk, const float* qkv, float* msg, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return mdgat_attention_sel(B, N, M, cross, topk, qkv, msg, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int mdgat_attention_sel(int B, int N, int M, int cross, int topk, const float* qkv, float* msg, uint32_t* sel,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!qkv || !msg || !workspace) { mdgat_set_error("mdgat_attention: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (topk < 0) { mdgat_set_error("mdgat_attention: topk < 0"); return MDGAT_ERR_BAD_ARG; }
    if (workspace_bytes < mdgat_attention_workspace_bytes(B, N, M) || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
        mdgat_set_error("mdgat_attention: workspace too small or not 16-byte aligned");
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Qkv16 q16 = mdgat_qkv16_carve(static_cast<_Float16*>(workspace), B, N, M);
    if (int rc = launch_qkv_split(B, N, M, qkv, q16, s)) return rc;
    return launch_attention(B, N, M, cross, topk, q16, msg, s, 0, sel);
}

extern "C" int mdgat_attention_qk_probe(int B, int N, int M, int cross, const float* qkv, float* msg, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (!qkv || !msg || !workspace) { mdgat_set_error("mdgat_attention_qk_probe: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (workspace_bytes < mdgat_attention_workspace_bytes(B, N, M) || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
        mdgat_set_error("mdgat_attention_qk_probe: workspace too small or not 16-byte aligned");
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Qkv16 q16 = mdgat_qkv16_carve(static_cast<_Float16*>(workspace), B, N, M);
    // (qkv == workspace: the operands are already there in the library's split layout - bench.py times only the probe)
    if (static_cast<const void*>(qkv) != workspace)
        if (int rc = launch_qkv_split(B, N, M, qkv, q16, s)) return rc;
    return launch_attention_qk_probe(B, N, M, cross, q16, msg, s);
}

extern "C" int mdgat_attention_qk_probe_sets(int B, int N, int M, int cross, int nq_sets, const float* qkv, float* msg, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    if (!qkv || !msg || !workspace) { mdgat_set_error("mdgat_attention_qk_probe_sets: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (workspace_bytes < mdgat_attention_workspace_bytes(B, N, M) || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
        mdgat_set_error("mdgat_attention_qk_probe_sets: workspace too small or not 16-byte aligned");
        return MDGAT_ERR_BAD_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Qkv16 q16 = mdgat_qkv16_carve(static_cast<_Float16*>(workspace), B, N, M);
    if (static_cast<const void*>(qkv) != workspace)
        if (int rc = launch_qkv_split(B, N, M, qkv, q16, s)) return rc;
    return launch_qk_phase_probe(B, N, M, cross, nq_sets, q16, msg, s);
}

extern "C" int mdgat_pointwise(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                               int relu, const float* R, int ldr, float* C, int ldc, void* stream) {
    if (!A || !W || !C) { mdgat_set_error("mdgat_pointwise: null pointer"); return MDGAT_ERR_BAD_ARG; }
    GemmArgs g = pointwise(A, lda, K, W, bias, relu, C, ldc, M, N);
    g.ldw = ldw; g.R = R; g.ldr = ldr;
    return launch_gemm(g, static_cast<hipStream_t>(stream));
}

extern "C" size_t mdgat_knn_workspace_bytes(int B, int C, int N, int M) { return mdgat_knn_ws_bytes_impl(B, C, N, M); }

extern "C" int mdgat_knn(int B, int C, int N, int M, int k, const float* x, const float* src, int64_t* idx, int64_t* adj,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !src || !idx) { mdgat_set_error("mdgat_knn: null pointer"); return MDGAT_ERR_BAD_ARG; }
    return launch_knn(B, C, N, M, k, x, src, idx, adj, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int mdgat_pose(int B, int N, int M, const float* kpts0, const float* kpts1, const int64_t* matches0,
                          const double* T_gt, double inlier_dist, double* T, double* stats, void* stream) {
    if (!kpts0 || !kpts1 || !matches0 || !T || !stats) { mdgat_set_error("mdgat_pose: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (N <= 0 || M <= 0) { mdgat_set_error("mdgat_pose: empty frame"); return MDGAT_ERR_BAD_ARG; }
    return launch_pose(B, N, M, kpts0, kpts1, matches0, T_gt, inlier_dist, T, stats, static_cast<hipStream_t>(stream));
}

extern "C" int mdgat_gt_matches(int B, int N, int M, const float* kpts0, const float* kpts1, const double* T0, const double* T1,
                                double threshold, int mutual, int64_t* gt0, int64_t* gt1, int64_t* rep, void* stream) {
    if (!kpts0 || !kpts1 || !gt0 || !gt1 || !rep) { mdgat_set_error("mdgat_gt_matches: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (N <= 0 || M <= 0) { mdgat_set_error("mdgat_gt_matches: empty frame"); return MDGAT_ERR_BAD_ARG; }
    return launch_gt_match(B, N, M, kpts0, kpts1, T0, T1, threshold, mutual, gt0, gt1, rep, static_cast<hipStream_t>(stream));
}

extern "C" int mdgat_pointwise_f64(int M, int N, int K, const double* A, int lda, const double* W, int ldw, const double* bias,
                                   int relu, const double* R, int ldr, double* C, int ldc, void* stream) {
    if (!A || !W || !C) { mdgat_set_error("mdgat_pointwise_f64: null pointer"); return MDGAT_ERR_BAD_ARG; }
    const GemmF64Args g{A, lda, K, nullptr, 0, W, ldw, bias, R, ldr, C, ldc, M, N, K, relu, nullptr};
    return launch_gemm_f64(g, static_cast<hipStream_t>(stream));
}

extern "C" int mdgat_attention_f64(int B, int N, int M, int cross, int topk, const double* qkv, double* msg, uint32_t* sel, void* stream) {
    if (!qkv || !msg) { mdgat_set_error("mdgat_attention_f64: null pointer"); return MDGAT_ERR_BAD_ARG; }
    if (topk < 0) { mdgat_set_error("mdgat_attention_f64: topk < 0"); return MDGAT_ERR_BAD_ARG; }
    return launch_attention_f64(B, N, M, cross, topk, qkv, msg, sel, static_cast<hipStream_t>(stream));
}
