// Seed retrieval of QueryReform: between two ReaRev iterations every
// instruction is fused with  seed_retrieve[b,:] = sum_n seed_info[b,n] * ent_emb[b,n,:]
// (reference gnn/modules/query_update.py:40, a [B,1,N] x [B,N,D] bmm that streams the whole node
// state, 102 MB at C2, to pick the one or two seed rows).  seed_info is the questions' seed indicator:
// here a wave scans the N flags of its question 64 at a time and reads only the rows whose flag is
// non-zero, in ascending node order (one fixed summation order).
#include "gnnrag_common.h"

namespace gnnrag {

__global__ __launch_bounds__(256) void k_seed_retrieve(const float* __restrict__ seed, const float* __restrict__ ent,
                                                       float* __restrict__ out, int N, int D) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const float* s = seed + (size_t)b * N;
  const float* e = ent + (size_t)b * N * D;
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + (int)threadIdx.x;
    float acc = 0.f;
    for (int base = 0; base < N; base += 64) {
      const int n = base + lane;
      const float v = n < N ? s[n] : 0.f;
      unsigned long long m = __ballot(v != 0.f);
      while (m) {
        const int j = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float sv = __shfl(v, j, 64);
        if (c < D) acc += sv * e[(size_t)(base + j) * D + c];
      }
    }
    if (c < D) out[(size_t)b * D + c] = acc;
  }
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_seed_retrieve(const float* seed_info, const float* ent_emb, float* out, int32_t B, int32_t N,
                                    int32_t D, gnnrag_stream_t stream) {
  if (!seed_info || !ent_emb || !out || B <= 0 || N <= 0 || D <= 0) return GNNRAG_E_BADARG;
  hipLaunchKernelGGL(k_seed_retrieve, dim3(B), dim3(256), 0, (hipStream_t)stream, seed_info, ent_emb, out, N, D);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}
