// fp64 Sinkhorn of the reference-exact mode (sinkhorn_f64.hip): declarations shared with api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// scores [B][N][M] fp64 -> Z (mdgat.py:279-308) as fp64 and / or its fp32 rounding, and / or the arg-maxes the match extraction needs
// (rbest [B][N]: per row over the columns - the inner M ones when `inner`, else including the dustbin; cbest [B][M] per column over the
// rows, the slabs of a pair merged in fp64; both decided on the fp64 values, the values handed on as fp32).  workspace: 256-byte aligned,
// sinkhorn_f64_workspace_bytes.  error_word: bit 0 raised when a workgroup gave up waiting for a partner (optional).
size_t sinkhorn_f64_workspace_bytes(int B, int N, int M);
bool sinkhorn_f64_supported(int N, int M);
int launch_sinkhorn_f64(int B, int N, int M, const double* scores, double alpha, int iters, double* Z64, float* Z32, int inner, int* rbest_idx,
                        float* rbest_val, int* cbest_idx, float* cbest_val, void* workspace, size_t workspace_bytes, unsigned* error_word,
                        hipStream_t s, const double* alpha_dev = nullptr);      // alpha_dev: the bin score on the device (replaces alpha)
size_t sinkhorn_f64_bests_bytes(int B, int N, int M);      // room for rbest / cbest (idx + val each) behind the kernel's own workspace
