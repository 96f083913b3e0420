// What layer.hip (one wave = 16 keypoints, weights through an LDS ring) and layer_split.hip (one wave = a slice of the
// output channels, weights straight into registers) share: the split weight images and the kernel arguments.
#pragma once
#include "common.hpp"

// Image row pitch (halves): hi plane | lo plane | 32 B pad.  A ds_read_b128 is served in four groups of 16
// lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; MI355X guide, LDS section): with lane (row l15, 16-byte
// column g) a pitch of 32 B mod 256 B puts the 16 lanes of every group on 16 different 16-byte bank groups.
constexpr int ROWH256 = 528;                 // K = 256
constexpr int ROWH128 = 272;                 // K = 128

struct LayerArgs {
    float* x;               // [R][128] descriptors, updated in place by phase 2
    const float* msg;       // [R][128] attention output (head-major channels)
    const _Float16* w1s;    // [256][ROWH256] split image (rows in P/Q order)
    const float* b1;        // [256]
    const _Float16* w2s;    // [128][ROWH256]
    const float* b2;        // [128]
    const _Float16* w3s;    // [384][ROWH128] (q|k|v of the next layer; v rows in natural order) or [128][ROWH128] (final_proj)
    const _Float16 *w1f, *w2f, *w3f;   // the same three in fragment order (common.hpp: launch_frag_image)
    const float* b3;        // [384] or [128]
    _Float16* q16;          // outputs of phase 3 (mode 1)
    _Float16* k16;
    _Float16* vt16;
    float* mdesc;           // [R][128] output of phase 3 (mode 2)
    int R, N, M, Npad, PP;
    unsigned* guard;        // optional, host-mapped: set when an input value is outside the f16 operand range or not finite
};

// layer_split.hip: the same layer for launches of a few tiles (one pair, small batches)
constexpr int MDGAT_LAYER_SPLIT_TILES_DEFAULT = 64;   // measured (tools/split_threshold.sh, N = M = 512): wins up to 64 tiles (B = 8), ties at 96-128, loses beyond
int launch_layer_split(const LayerArgs& a, int do_mlp, int mode3, hipStream_t s);
