// Internal entry points shared by the dense kernels' translation units and the layer driver (softmax_layer.hip).
#pragma once
#include "gnnrag_common.h"

namespace gnnrag {

// tables_vq_launch that ALSO zeroes `zero_n` floats at `zero` (may be null): the layer driver passes the layer's score
// buffer, onto which k_update_b3 accumulates its two column parts' shares - that kernel's own memset launch (~5 us
// per layer) is then skipped (update_score_fused_z(..., score_zeroed = true)).
int tables_vq_launch_z(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                       int32_t I, int32_t only_dir, float* zero, int64_t zero_n, hipStream_t stream);

int update_b3_launch_z(const float* h, const float* nbr, const float* W, const float* b, const float* w_s,
                       const float* b_s, const float* mask, float* h_out, float* score, int64_t BN, int32_t D,
                       int32_t ldw, hipStream_t stream, bool score_zeroed);

// gnnrag_update_score_fused with the promise that `score` already holds zeros
int update_score_fused_z(const float* h, const float* nbr, const float* W, const float* b, const float* w_s,
                         const float* b_s, const float* mask, float* h_out, float* score, int64_t BN, int32_t D,
                         int32_t I, int32_t math, hipStream_t stream, bool score_zeroed);

}  // namespace gnnrag
