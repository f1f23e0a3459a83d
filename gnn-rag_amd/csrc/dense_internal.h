// Internal entry points shared by the dense kernels' translation units and the layer driver (softmax_layer.hip).
#pragma once
#include "gnnrag_common.h"

namespace gnnrag {

// tables_vq_launch that ALSO zeroes `zero_n` floats at `zero` (may be null): the layer driver passes the layer's score
// buffer, onto which k_update_b3 accumulates its two column parts' shares - that kernel's own memset launch (~5 us
// per layer) is then skipped (update_score_fused_z(..., score_zeroed = true)).
int tables_vq_launch_z(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                       int32_t I, int32_t only_dir, float* zero, int64_t zero_n, hipStream_t stream);

// the same tables from k_tables_vq_lite (256 threads, 81.5 KB of LDS: fits a CU beside a workgroup of the LDS walk) - what the
// stack driver's side stream launches; bit-identical to tables_vq_launch_z
int tables_vq_lite_launch_z(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                            int32_t I, int32_t only_dir, float* zero, int64_t zero_n, hipStream_t stream);

int update_b3_launch_z(const float* h, const float* nbr, const float* W, const float* b, const float* w_s,
                       const float* b_s, const float* mask, float* h_out, float* score, int64_t BN, int32_t D,
                       int32_t ldw, hipStream_t stream, bool score_zeroed);

// Row-gated form of the self-block update: `nbr` holds valid data only in the rows whose byte in add_flag is non-zero,
// every other row reads the ZERO ROW behind the buffer (row index BN), so the unflagged rows of nbr are never
// touched (frontier layers: frontier.hip).  add_flag: [BN + at least 16 bytes of readable padding] (the kernels
// read a lane's four row gates as one dword at tile granularity: up to 15 bytes past BN when BN % 16 != 0; the frontier
// workspace pads by 64), 4-byte aligned.
// Returns GNNRAG_E_UNSUPPORTED (nothing launched) when the kernel the shape would take has no gated form.
int update_score_fused_rows(const float* h, const float* nbr, const uint8_t* add_flag, const float* W, const float* b,
                            const float* w_s, const float* b_s, const float* mask, float* h_out, float* score,
                            int64_t BN, int32_t D, int32_t I, int32_t math, hipStream_t stream, bool score_zeroed);
int update_b3_launch_f(const float* h, const float* nbr, const uint8_t* add_flag, const float* W, const float* b,
                       const float* w_s, const float* b_s, const float* mask, float* h_out, float* score, int64_t BN,
                       int32_t D, int32_t ldw, hipStream_t stream, bool score_zeroed);

// true when update_score_fused_rows has a row-gated kernel for this shape / alignment / math mode (gemm_f32.hip)
bool update_rows_supported(const float* h, const float* nbr, const float* W, const float* h_out, int64_t BN, int32_t D,
                           int32_t I, int32_t math);
bool update_b3_shape_ok(int64_t BN, int32_t D, int32_t ldw);
// frontier.hip: relation tables of small batches on the one-workgroup-per-tile, split-k kernel (exact fp32)
int tables_small_launch(const gnnrag_csr* csr, const float* T_fwd, const float* T_inv, const float* ins, const float* W,
                        float* P, int32_t D, int32_t I, hipStream_t stream);

// frontier.hip: the frontier of a (sparse) prior; also zeroes two float ranges on the way (score buffer, zero row)
int frontier_build_z(const gnnrag_csr* csr, const float* dist, void* fws, size_t fws_bytes, float* zero_a,
                     int64_t zero_na, float* zero_b, int64_t zero_nb, hipStream_t stream);
const uint8_t* frontier_row_flags(const gnnrag_csr* csr, const void* fws);

// gnnrag_update_score_fused with the promise that `score` already holds zeros
int update_score_fused_z(const float* h, const float* nbr, const float* W, const float* b, const float* w_s,
                         const float* b_s, const float* mask, float* h_out, float* score, int64_t BN, int32_t D,
                         int32_t I, int32_t math, hipStream_t stream, bool score_zeroed);

// rel_transform.hip: whether gnnrag_rel_transform takes these operands (then, and only then, it writes T and the planes)
bool rel_transform_accepts(const float* relfeat_fwd, const float* relfeat_inv, int64_t R1, int32_t D, int32_t L,
                           const gnnrag_layer_params* layers, int32_t pos_rows, const float* T_out, const void* planes_out);

}  // namespace gnnrag
