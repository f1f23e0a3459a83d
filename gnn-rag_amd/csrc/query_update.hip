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

// QueryReform.forward in ONE launch (query_update.py:26-44 -> Fusion, :6-16): the seed retrieval above, then
//   feats = [x, y, x - y]   gate = sigmoid(W_g feats)   out = gate * (W_r feats) + (1 - gate) * x
// for x = q_node[b] and y = seed_retrieve[b].  The torch form is ~11 launches per call and 6 calls per forward; at
// batch 1 that is 0.7 ms of the forward's 2.8 ms of host time (profiles/r05i_forward_host_time.txt).
// Workgroup (b, s): part s of the D outputs of question b (every part retrieves y itself: one or two rows); a wave
// owns an output at a time, its lanes stride over the 3 D terms (coalesced rows of W), fixed shuffle tree: one
// summation order.  ent_emb rows may be padded (row stride ldE >= D).
__global__ __launch_bounds__(256) void k_query_reform(const float* __restrict__ q, const float* __restrict__ seed,
                                                      const float* __restrict__ ent, const float* __restrict__ Wr,
                                                      const float* __restrict__ Wg, float* __restrict__ out, int N, int D,
                                                      int ldE) {
  extern __shared__ float feats[];                       // [3 D]
  const int b = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* s = seed + (size_t)b * N;
  const float* e = ent + (size_t)b * N * ldE;
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
        if (c < D) acc += sv * e[(size_t)(base + j) * ldE + c];
      }
    }
    if (c < D) {
      const float x = q[(size_t)b * D + c];
      feats[c] = x;
      feats[D + c] = acc;
      feats[2 * D + c] = x - acc;
    }
  }
  __syncthreads();
  const int K = 3 * D;
  const int j0 = (int)((long long)D * part / nparts), j1 = (int)((long long)D * (part + 1) / nparts);
  // two outputs per wave and round: four independent load streams in flight (the loop is a chain of L2 round trips)
  for (int j = j0 + wave; j < j1; j += 8) {
    const int jb = j + 4 < j1 ? j + 4 : j;               // the second output of the round (the first again at the end)
    const float* wr0 = Wr + (size_t)j * K;
    const float* wg0 = Wg + (size_t)j * K;
    const float* wr1 = Wr + (size_t)jb * K;
    const float* wg1 = Wg + (size_t)jb * K;
    float ar0 = 0.f, ag0 = 0.f, ar1 = 0.f, ag1 = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float f = feats[k];
      ar0 += wr0[k] * f;
      ag0 += wg0[k] * f;
      ar1 += wr1[k] * f;
      ag1 += wg1[k] * f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ar0 += __shfl_xor(ar0, o, 64);
      ag0 += __shfl_xor(ag0, o, 64);
      ar1 += __shfl_xor(ar1, o, 64);
      ag1 += __shfl_xor(ag1, o, 64);
    }
    if (lane == 0) {
      const float gate = 1.f / (1.f + expf(-ag0));
      out[(size_t)b * D + j] = gate * ar0 + (1.f - gate) * feats[j];
    }
    if (lane == 1 && jb != j) {
      const float gate = 1.f / (1.f + expf(-ag1));
      out[(size_t)b * D + jb] = gate * ar1 + (1.f - gate) * feats[jb];
    }
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

extern "C" int gnnrag_query_reform(const float* q_node, const float* seed_info, const float* ent_emb, int64_t ld_ent,
                                   const float* W_r, const float* W_g, float* out, int32_t B, int32_t N, int32_t D,
                                   gnnrag_stream_t stream) {
  if (!q_node || !seed_info || !ent_emb || !W_r || !W_g || !out || B <= 0 || N <= 0 || D <= 0 || ld_ent < D)
    return GNNRAG_E_BADARG;
  if ((size_t)3 * D * sizeof(float) > 48 * 1024) return GNNRAG_E_UNSUPPORTED;       // feats in LDS: D <= 4096
  // workgroups of a question share its outputs: about one workgroup per CU at batch 64, at most 8 per question
  int parts = (256 + B - 1) / B;
  parts = parts < 1 ? 1 : parts > 8 ? 8 : parts;
  hipLaunchKernelGGL(k_query_reform, dim3(B, parts), dim3(256), (size_t)3 * D * sizeof(float), (hipStream_t)stream, q_node,
                     seed_info, ent_emb, W_r, W_g, out, N, D, (int)ld_ent);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}
