// Relation projections of ALL layers of one iteration in one launch:
//
//   T[j][d][r, :] = rel_linear{j}(rel_features_d[r, :]) (+ pos_emb{j}_d[r, :])     reasongnn.py:75-79 / :102-105
//
// for layer j = 0..L-1 and direction d (forward / inverse relation features).  None of them depends on the node
// state or on the distribution, so the L x 2 skinny products ([R1, D] x [D, D], R1 = a few hundred to a few thousand
// relation rows) that used to head every layer call - each a ~13-25 us launch of a few dozen workgroups waiting on a
// dependent k loop - become one launch of (row blocks x column tiles x 2L) small workgroups in front of the layer
// sequence.  Exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32).
//
// Workgroup = 4 waves, wave w owns rows [64 bx + 16 w, +16) x columns [16 by, +16) of T[z]: lane (fr, fg) loads the
// float4 A[row fr][16 c + 4 fg ..] and W[col fr][16 c + 4 fg ..] of k group c straight from global memory (L2
// resident: relation features and the weight are a few hundred KB), kRtUnroll groups in flight; component s of the
// two float4 feeds MFMA s (the k order inside a group is a permutation both operands share).
#include "gnnrag_common.h"

namespace gnnrag {

constexpr int kRtMaxL = 8;        // layers per launch (more layers: more launches)
constexpr int kRtUnroll = 7;      // k groups (16 columns each) in flight per wave: K = 200 -> 13 groups -> 2 rounds

struct RelTArgs {
  const float* A[2];              // rel_features, rel_features_inv  [M, K]
  const float* W[kRtMaxL];        // rel_linear{j}.weight [N, K]
  const float* b[kRtMaxL];        // rel_linear{j}.bias [N] or null
  const float* add[kRtMaxL][2];   // pos_emb{j} / pos_emb_inv{j} [add_rows, N] or null
  float* C;                       // [L][2][M][N]
  int M, K, N, add_rows;
};

__global__ __launch_bounds__(256) void k_rel_transform(RelTArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int j = blockIdx.z >> 1, d = blockIdx.z & 1;
  const int row0 = blockIdx.x * 64 + wave * 16;
  if (row0 >= g.M) return;
  const int col0 = blockIdx.y * 16;
  const int K = g.K;
  // rows / columns past the end read the last valid one (never stored)
  const int ar = min(row0 + fr, g.M - 1), wr = min(col0 + fr, g.N - 1);
  const float* __restrict__ ap = g.A[d] + (size_t)ar * K;
  const float* __restrict__ wp = g.W[j] + (size_t)wr * K;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc0 = zero4, acc1 = zero4;
  const int nch = (K + 15) >> 4;
  for (int c0 = 0; c0 < nch; c0 += kRtUnroll) {
    f32x4 a[kRtUnroll], w[kRtUnroll];
#pragma unroll
    for (int u = 0; u < kRtUnroll; ++u) {
      // k groups past K are read from a clamped (valid) address and zeroed at use: no branch around a load
      const int kk = min(16 * (c0 + u) + 4 * fg, K - 4);
      a[u] = *reinterpret_cast<const f32x4*>(ap + kk);
      w[u] = *reinterpret_cast<const f32x4*>(wp + kk);
    }
#pragma unroll
    for (int u = 0; u < kRtUnroll; ++u) {
      const bool live = 16 * (c0 + u) + 4 * fg < K;
      const f32x4 av = live ? a[u] : zero4;
      const f32x4 wv = live ? w[u] : zero4;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wv[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wv[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wv[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wv[3], acc1, 0, 0, 0);
    }
  }
  // accumulator element r of lane (fr, fg) = C[row0 + 4 fg + r][col0 + fr]
  const int col = col0 + fr;
  if (col >= g.N) return;
  const float bias = g.b[j] ? g.b[j][col] : 0.f;
  const float* __restrict__ addp = g.add[j][d];
  float* __restrict__ cp = g.C + ((size_t)(2 * j + d) * g.M) * g.N;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * fg + r;
    if (row < g.M) {
      float v = acc0[r] + acc1[r] + bias;
      if (addp && row < g.add_rows) v += addp[(size_t)row * g.N + col];
      cp[(size_t)row * g.N + col] = v;
    }
  }
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_rel_transform(const float* relfeat_fwd, const float* relfeat_inv, int64_t R1, int32_t D,
                                    int32_t L, const gnnrag_layer_params* layers, int32_t pos_rows, float* T_out,
                                    gnnrag_stream_t stream) {
  if (!relfeat_fwd || !relfeat_inv || !layers || !T_out || R1 < 0 || D <= 0 || L <= 0 || pos_rows < 0)
    return GNNRAG_E_BADARG;
  if (D & 3) return GNNRAG_E_UNSUPPORTED;            // float4 operand loads
  if (R1 >= ((int64_t)1 << 31) / D) return GNNRAG_E_UNSUPPORTED;
  if (R1 == 0) return 0;
  for (int j = 0; j < L; ++j) {
    if (!layers[j].W_rel) return GNNRAG_E_BADARG;
    if ((layers[j].pos_fwd == nullptr) != (layers[j].pos_inv == nullptr)) return GNNRAG_E_BADARG;
  }
  for (int j0 = 0; j0 < L; j0 += kRtMaxL) {
    const int n = L - j0 < kRtMaxL ? L - j0 : kRtMaxL;
    RelTArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = relfeat_fwd;
    g.A[1] = relfeat_inv;
    for (int j = 0; j < n; ++j) {
      const gnnrag_layer_params& p = layers[j0 + j];
      g.W[j] = p.W_rel;
      g.b[j] = p.b_rel;
      g.add[j][0] = pos_rows > 0 ? p.pos_fwd : nullptr;
      g.add[j][1] = pos_rows > 0 ? p.pos_inv : nullptr;
    }
    g.C = T_out + (size_t)j0 * 2 * R1 * D;
    g.M = (int)R1; g.K = D; g.N = D;
    g.add_rows = (int)(pos_rows < R1 ? pos_rows : R1);
    const dim3 grid((unsigned)((R1 + 63) / 64), (unsigned)((D + 15) / 16), (unsigned)(2 * n));
    hipLaunchKernelGGL(k_rel_transform, grid, dim3(256), 0, (hipStream_t)stream, g);
    GNNRAG_LAUNCH_CHECK();
  }
  return 0;
}
