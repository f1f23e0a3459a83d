// Weight gradients of the dense projections:  C[N1, N2] = A[M, N1]^T . B[M, N2]   (M = nodes or relation rows, large;
// N1, N2 = a few hundred) - what autograd derives for nn.Linear.weight in training (train_model.py:209-233 through
// reasongnn.py:75-79, :161-165): dW = dY^T . X.  Exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32).
//
// The reduction runs over ROWS, so both operands are read as they lie: lane (fr, fg) of a wave loads the float4
// A[m0 + fg][n1_0 + 4 fr ..] and B[m0 + fg][n2_0 + 4 fr ..] (a row's 16 lanes cover 256 contiguous bytes) - component
// x of the A piece is column slot fr of output-row tile x, component y of the B piece slot fr of output-column tile
// y (interleaved slots: slot fr of tile x = index 4 fr + x), 16 MFMAs per pair of loads, a 64 x 64 block of C per
// wave.  Workgroup = 4 waves = 4 output blocks over the same row chunk (shared operand lines hit L1); grid = row
// chunks x groups of output blocks.  Every chunk writes its partial block; k_tn_reduce adds the chunks in chunk order
// (no atomics: one fixed summation order).
#include "gnnrag_common.h"

namespace gnnrag {

constexpr int kTnUnroll = 4;       // k steps (4 rows each) in flight per wave

struct TnArgs {
  const float* A;        // [M, N1]
  const float* B;        // [M, N2]
  float* part;           // [chunks][N1][N2]
  int64_t M;
  int32_t N1, N2, nb1, nb2, chunk_rows;
};

__global__ __launch_bounds__(256) void k_gemm_tn(TnArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int blk = blockIdx.x * 4 + wave;                      // this wave's 64 x 64 output block
  if (blk >= g.nb1 * g.nb2) return;
  const int n1_0 = (blk / g.nb2) * 64, n2_0 = (blk % g.nb2) * 64;
  const int64_t m_beg = (int64_t)blockIdx.y * g.chunk_rows;
  const int64_t m_end = m_beg + g.chunk_rows < g.M ? m_beg + g.chunk_rows : g.M;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // pieces past N1 / N2 read the last valid one (never stored); rows past the chunk are zeroed at use
  const int ca = min(n1_0 + 4 * fr, g.N1 - 4), cb = min(n2_0 + 4 * fr, g.N2 - 4);
  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = zero4;
  for (int64_t m0 = m_beg; m0 < m_end; m0 += 4 * kTnUnroll) {
    f32x4 av[kTnUnroll], bv[kTnUnroll];
#pragma unroll
    for (int u = 0; u < kTnUnroll; ++u) {
      const int64_t m = m0 + 4 * u + fg;
      const int64_t mc = m < g.M ? m : g.M - 1;               // clamped address: no branch around a load
      av[u] = *reinterpret_cast<const f32x4*>(g.A + mc * g.N1 + ca);
      bv[u] = *reinterpret_cast<const f32x4*>(g.B + mc * g.N2 + cb);
    }
#pragma unroll
    for (int u = 0; u < kTnUnroll; ++u) {
      const bool live = m0 + 4 * u + fg < m_end;
      const f32x4 a = live ? av[u] : zero4;
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], bv[u][y], acc[x][y], 0, 0, 0);
    }
  }
  // acc[x][y][r] = C[n1_0 + 4 (4 fg + r) + x][n2_0 + 4 fr + y]
  float* out = g.part + (size_t)blockIdx.y * g.N1 * g.N2;
  const int c2 = n2_0 + 4 * fr;
  if (c2 >= g.N2) return;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c1 = n1_0 + 4 * (4 * fg + r) + x;
      if (c1 < g.N1) {
        const f32x4 v = {acc[x][0][r], acc[x][1][r], acc[x][2][r], acc[x][3][r]};
        *reinterpret_cast<f32x4*>(out + (size_t)c1 * g.N2 + c2) = v;
      }
    }
}

__global__ __launch_bounds__(256) void k_tn_reduce(const f32x4* __restrict__ part, f32x4* __restrict__ C, int64_t n4,
                                                   int chunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = part[i];
  for (int c = 1; c < chunks; ++c) s += part[(size_t)c * n4 + i];      // chunk order: one fixed summation order
  C[i] = s;
}

static int tn_chunks(int64_t M, int32_t N1, int32_t N2, int* chunk_rows) {
  int cus = 0;
  if (device_cu_count(&cus) != 0 || cus <= 0) cus = 256;
  const int nblk = ((N1 + 63) / 64) * ((N2 + 63) / 64);
  const int groups = (nblk + 3) / 4;
  int chunks = (4 * cus + groups - 1) / groups;              // ~4 workgroups per CU
  const int64_t max_chunks = (M + 255) / 256;                // at least 256 rows per chunk
  if (chunks > max_chunks) chunks = (int)max_chunks;
  if (chunks < 1) chunks = 1;
  int64_t rows = (M + chunks - 1) / chunks;
  rows = (rows + 4 * kTnUnroll - 1) / (4 * kTnUnroll) * (4 * kTnUnroll);
  *chunk_rows = (int)rows;
  return (int)((M + rows - 1) / rows);
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" size_t gnnrag_gemm_tn_workspace_bytes(int64_t M, int32_t N1, int32_t N2) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
  int rows = 0;
  const int chunks = tn_chunks(M, N1, N2, &rows);
  return (size_t)chunks * N1 * N2 * sizeof(float);
}

extern "C" int gnnrag_gemm_tn(const float* A, const float* B, int64_t M, int32_t N1, int32_t N2, float* C,
                              void* workspace, size_t workspace_bytes, gnnrag_stream_t stream) {
  if (!A || !B || !C || M < 0 || N1 <= 0 || N2 <= 0) return GNNRAG_E_BADARG;
  if ((N1 & 3) || (N2 & 3) || M >= ((int64_t)1 << 40)) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)workspace) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  if (M == 0) {
    GNNRAG_HIP(hipMemsetAsync(C, 0, (size_t)N1 * N2 * sizeof(float), (hipStream_t)stream));
    return 0;
  }
  TnArgs g;
  memset(&g, 0, sizeof(g));
  int rows = 0;
  const int chunks = tn_chunks(M, N1, N2, &rows);
  if (!workspace || workspace_bytes < (size_t)chunks * N1 * N2 * sizeof(float)) return GNNRAG_E_WORKSPACE;
  g.A = A; g.B = B; g.part = (float*)workspace; g.M = M; g.N1 = N1; g.N2 = N2;
  g.nb1 = (N1 + 63) / 64; g.nb2 = (N2 + 63) / 64; g.chunk_rows = rows;
  const int groups = (g.nb1 * g.nb2 + 3) / 4;
  hipLaunchKernelGGL(k_gemm_tn, dim3((unsigned)groups, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, g);
  GNNRAG_LAUNCH_CHECK();
  const int64_t n4 = (int64_t)N1 * N2 / 4;
  hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)workspace, (f32x4*)C, n4, chunks);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}
