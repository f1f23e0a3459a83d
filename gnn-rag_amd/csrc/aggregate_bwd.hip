// Backward of the typed-edge aggregation: what torch autograd derives for
// reason_layer / reason_layer_inv (reference gnn/modules/kg_reasoning/reasongnn.py:61-116) and for
// TypeLayer's two sparse products (gnn/modules/layer_init.py:47-57), so that Trainer_KBQA.train_epoch
// (gnn/train_model.py:209-233) can run on the HIP operator.
//
//   forward   agg[n, 2i+d, :] = sum_{f: dst_d(f)=n} w_f * dist[src_d(f)] * relu(T_d[rel_f,:] * ins[b,i,:])
//
//   g_dist[s]    = sum_d sum_{f: src_d(f)=s} w_f * sum_i < g_agg[dst_d(f), 2i+d, :], relu(T_d[rel_f] * ins[b,i]) >
//   U[d,b,r,i,:] = sum_{f in b, rel_f=r} w_f * dist[src_d(f)] * g_agg[dst_d(f), 2i+d, :]
//   g_T_d[r,:]   = sum_b sum_i U[d,b,r,i,:] * [T_d[r,:]*ins[b,i,:] > 0] * ins[b,i,:]
//   g_ins[b,i,:] = sum_d sum_r U[d,b,r,i,:] * [T_d[r,:]*ins[b,i,:] > 0] * T_d[r,:]
//
// Two kernels, both walks of the same destination-sorted structure the forward uses:
//  * k_bwd_prior: the facts with source s in direction d are row s of the OTHER direction's structure
//    (its records hold (dst_d(f), rel_f)), so g_dist is a gather - one wave per node, lanes across the
//    D columns, one cross-lane reduction per node; rows above heavy_deg go to one wave per 256-fact
//    chunk (k_bwd_prior_heavy) and are added atomically.
//  * k_bwd_tables: U is the transpose of the fused forward's relation tables: a workgroup owns
//    (question, 16-column slice, instruction), keeps U[2][relations the question uses][16] in LDS,
//    walks the question's nodes exactly like the forward LDS walk but ADDS p * g_agg[n, cols] into the
//    row of the fact's relation (ds_add_f32) instead of reading it; the epilogue applies the ReLU gate
//    and reduces to g_ins (exclusive store) and g_T (global fp32 atomics over the questions).  U never
//    reaches HBM.  Sums are in atomic order: gradients are reproducible to rounding, not bit for bit.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "gnnrag_common.h"

namespace gnnrag {

constexpr int kBwdSliceW = 16;
constexpr int kBwdThreads = 1024;

enum { BWD_REASON = 0, BWD_TYPE = 1 };

struct BwdArgs {
  const int32_t* row_ptr[2];
  const int2* edge[2];      // (source node, relation id)                     - k_bwd_prior
  const int2* edge_l[2];    // (source node, compact relation of the question) - k_bwd_tables
  const float* w[2];        // per-fact weight in sorted order or nullptr
  const int32_t* heavy[2];
  const int32_t* chunk_off[2];
  const int32_t* n_heavy;
  const int32_t* n_chunks;
  int32_t heavy_cap, max_chunks, heavy_deg;
  const int32_t* rel_off;
  const int2* rel_rows;
  const int32_t* big_cnt;
  const int32_t* big_nodes;
  int32_t big_deg;
  const float* dist;        // [BN]
  const float* ins;         // [B,I,D]
  const float* T[2];        // [R1,D]
  const float* g;           // REASON: g_agg [BN, 2I*D]; TYPE: gradient of the pre-activation [BN, D]
  float* g_dist;            // [BN]
  float* g_ins;             // [B,I,D]
  float* g_T[2];            // [R1,D] (TYPE: only [0])
  int32_t B, N, D, I, Rmax;
};

// facts [j0, j1) of structure o = 1 - d, all with the same source: this lane's share (columns
// lane, lane + 64, ...) of  sum_f w_f sum_i < g_agg[dst_f, 2i+d, :], relu(T_d[rel_f,:] * q_i) >.
// Two facts per step so their row loads overlap.
__device__ __forceinline__ float prior_grad_range(const BwdArgs& a, const float* __restrict__ q, int d, int j0,
                                                  int j1, int lane) {
  const int o = 1 - d;
  const int D = a.D, I = a.I;
  const size_t ld = (size_t)2 * I * D;
  const int2* __restrict__ edge = a.edge[o];
  const float* __restrict__ w = a.w[o];
  const float* __restrict__ T = a.T[d];
  const float* __restrict__ g = a.g + (size_t)d * D;
  float acc = 0.f;
  int j = j0;
  for (; j + 1 < j1; j += 2) {
    const int2 e0 = edge[j], e1 = edge[j + 1];
    const float w0 = w ? w[j] : 1.f, w1 = w ? w[j + 1] : 1.f;
    const float* t0 = T + (size_t)e0.y * D;
    const float* t1 = T + (size_t)e1.y * D;
    const float* g0 = g + (size_t)e0.x * ld;
    const float* g1 = g + (size_t)e1.x * ld;
    float p0 = 0.f, p1 = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float tv0 = t0[c], tv1 = t1[c];
      for (int i = 0; i < I; ++i) {
        const float qv = q[i * D + c];
        p0 += g0[(size_t)2 * i * D + c] * fmaxf(tv0 * qv, 0.f);
        p1 += g1[(size_t)2 * i * D + c] * fmaxf(tv1 * qv, 0.f);
      }
    }
    acc += w0 * p0 + w1 * p1;
  }
  if (j < j1) {
    const int2 e0 = edge[j];
    const float w0 = w ? w[j] : 1.f;
    const float* t0 = T + (size_t)e0.y * D;
    const float* g0 = g + (size_t)e0.x * ld;
    float p0 = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float tv0 = t0[c];
      for (int i = 0; i < I; ++i) p0 += g0[(size_t)2 * i * D + c] * fmaxf(tv0 * q[i * D + c], 0.f);
    }
    acc += w0 * p0;
  }
  return acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per source node; blockIdx.y = question (its instructions are staged in LDS once)
__global__ __launch_bounds__(256) void k_bwd_prior(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float q_s[];      // [I][D]
  const int b = blockIdx.y;
  for (int x = threadIdx.x; x < a.I * a.D; x += 256) q_s[x] = a.ins[(size_t)b * a.I * a.D + x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nl = blockIdx.x * 4 + wave;
  if (nl >= a.N) return;
  const int s = b * a.N + nl;
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int o = 1 - d;
    const int beg = a.row_ptr[o][s], end = a.row_ptr[o][s + 1];
    if (end - beg > a.heavy_deg) continue;           // k_bwd_prior_heavy adds these
    acc += prior_grad_range(a, q_s, d, beg, end, lane);
  }
  acc = wave_sum(acc);
  if (lane == 0) a.g_dist[s] = acc;
}

// rows above heavy_deg: one wave per 256-fact chunk, added atomically (after k_bwd_prior's store)
__global__ __launch_bounds__(256) void k_bwd_prior_heavy(const BwdArgs a) {
  const int o = blockIdx.y, d = 1 - o;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cnt = min(a.n_heavy[o], a.heavy_cap);
  const int nch = min(a.n_chunks[o], a.max_chunks);
  const int32_t* off = a.chunk_off[o];
  for (int c = blockIdx.x * 4 + wave; c < nch; c += gridDim.x * 4) {
    int lo = 0, hi = cnt;                       // largest e with off[e] <= c
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= c) lo = mid; else hi = mid;
    }
    const int s = a.heavy[o][lo];
    const int beg = a.row_ptr[o][s] + (c - off[lo]) * kHeavyDeg;
    const int end = min(beg + kHeavyDeg, a.row_ptr[o][s + 1]);
    const float* q = a.ins + (size_t)(s / a.N) * a.I * a.D;
    const float acc = wave_sum(prior_grad_range(a, q, d, beg, end, lane));
    if (lane == 0) unsafeAtomicAdd(a.g_dist + s, acc);
  }
}

// ---- relation-bucketed sums through LDS --------------------------------------------------------
// this lane group's 4 columns of row `row` of g (guarded, any D)
template <bool V4>
__device__ __forceinline__ f32x4 load_cols(const float* __restrict__ row, int c0, int D) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if constexpr (V4) {
    if (c0 < D) v = *reinterpret_cast<const f32x4*>(row + c0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < D) v[e] = row[c0 + e];
  }
  return v;
}

// facts first, first+stride, ... of row n in direction d: U[d][rel][cols] += p * gv
template <int MODE>
__device__ __forceinline__ void bucket_row(const BwdArgs& a, float* __restrict__ Ud, int d, int n, int first,
                                           int stride, int sub, const f32x4& gv) {
  const int beg = a.row_ptr[d][n], end = a.row_ptr[d][n + 1];
  const int2* __restrict__ edge = a.edge_l[d];
  const float* __restrict__ w = a.w[d];
  for (int j = beg + first; j < end; j += stride) {
    const int2 e = edge[j];
    float p = w ? w[j] : 1.f;
    if constexpr (MODE == BWD_REASON) p *= a.dist[e.x];
    if (p != 0.f) {
      float* u = Ud + (size_t)e.y * kBwdSliceW + 4 * sub;
#pragma unroll
      for (int k = 0; k < 4; ++k) unsafeAtomicAdd(u + k, p * gv[k]);
    }
  }
}

template <int MODE, bool V4>
__global__ __launch_bounds__(kBwdThreads) void k_bwd_tables(const BwdArgs a, int nslice) {
  constexpr int ND = (MODE == BWD_REASON) ? 2 : 1;     // TYPE: both directions read the same table
  extern __shared__ __attribute__((aligned(16))) float s_u[];   // U [ND][Rg][16], then red [16 waves][16]
  float* red = s_u + (size_t)ND * a.Rmax * kBwdSliceW;
  const int NI = (MODE == BWD_REASON) ? a.I : 1;
  // XCD-aware order: the workgroups of question g land on XCD g % 8 (they share its CSR rows in L2)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per = nslice * NI;
  const int g = (slot / per) * 8 + xcd;
  if (g >= a.B) return;
  const int rem = slot % per;
  const int c = rem % nslice, i = rem / nslice;
  const int col0 = c * kBwdSliceW;
  const int D = a.D, N = a.N;
  const int roff = a.rel_off[g], Rg = a.rel_off[g + 1] - roff;
  const int tid = threadIdx.x;
  for (int x = tid; x < ND * Rg * kBwdSliceW; x += kBwdThreads) s_u[x] = 0.f;
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 2, sub = lane & 3;
  const int c0 = col0 + 4 * sub;
  const size_t ld = (MODE == BWD_REASON) ? (size_t)2 * a.I * D : (size_t)D;
  // nodes with many facts (listed at plan time): a whole wave per node, lane group k takes facts k, k+16, ...
  const int nbig = a.big_cnt[g];
  for (int h = wave; h < nbig; h += 16) {
    const int n = a.big_nodes[(size_t)g * N + h];
    const float* grow = a.g + (size_t)n * ld;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const f32x4 gv = load_cols<V4>(grow + (MODE == BWD_REASON ? (size_t)(2 * i + d) * D : 0), c0, D);
      bucket_row<MODE>(a, s_u + (size_t)(ND == 2 ? d : 0) * Rg * kBwdSliceW, d, n, grp, 16, sub, gv);
    }
  }
  // everything else: a 4-lane group per node
  for (int nl = tid >> 2; nl < N; nl += kBwdThreads / 4) {
    const int n = g * N + nl;
    const int l0 = a.row_ptr[0][n + 1] - a.row_ptr[0][n], l1 = a.row_ptr[1][n + 1] - a.row_ptr[1][n];
    if (max(l0, l1) > a.big_deg || (l0 | l1) == 0) continue;
    const float* grow = a.g + (size_t)n * ld;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const f32x4 gv = load_cols<V4>(grow + (MODE == BWD_REASON ? (size_t)(2 * i + d) * D : 0), c0, D);
      bucket_row<MODE>(a, s_u + (size_t)(ND == 2 ? d : 0) * Rg * kBwdSliceW, d, n, 0, 1, sub, gv);
    }
  }
  __syncthreads();

  // epilogue: thread t owns float4 granule k = t & 3 of the slice for the rows t/4, t/4 + 256, ...
  const int k = tid & 3;
  const int ce = col0 + 4 * k;
  if constexpr (MODE == BWD_TYPE) {
    for (int idx = tid; idx < Rg * 4; idx += kBwdThreads) {
      const int r = idx >> 2;
      const int rg = a.rel_rows[roff + r].y;
      const float* u = s_u + (size_t)r * kBwdSliceW + 4 * k;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ce + e < D && u[e] != 0.f) unsafeAtomicAdd(a.g_T[0] + (size_t)rg * D + ce + e, u[e]);
    }
  } else {
    const f32x4 q = load_cols<V4>(a.ins + ((size_t)g * a.I + i) * D, ce, D);
    f32x4 gq = {0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < 2 * Rg * 4; idx += kBwdThreads) {
      const int d = idx >= Rg * 4;
      const int r = (idx - d * Rg * 4) >> 2;
      const int rg = a.rel_rows[roff + r].y;
      const float* u = s_u + ((size_t)d * Rg + r) * kBwdSliceW + 4 * k;
      const f32x4 t = load_cols<V4>(a.T[d] + (size_t)rg * D, ce, D);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (t[e] * q[e] > 0.f && u[e] != 0.f) {        // the ReLU gate of relu(T_d[r] * ins[b,i])
          gq[e] += u[e] * t[e];
          unsafeAtomicAdd(a.g_T[d] + (size_t)rg * D + ce + e, u[e] * q[e]);
        }
      }
    }
    // g_ins[g,i,slice]: sum gq over the threads that own the same granule
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) gq[e] += __shfl_xor(gq[e], o, 64);
    }
    if (lane < 4) *reinterpret_cast<f32x4*>(red + wave * 16 + 4 * lane) = gq;   // lane = k here
    __syncthreads();
    if (tid < 16) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += red[w * 16 + tid];
      if (col0 + tid < D) a.g_ins[((size_t)g * a.I + i) * D + col0 + tid] = t;
    }
  }
}

static size_t bwd_lds_bytes(int nd, int rmax) {
  return ((size_t)nd * rmax * kBwdSliceW + 16 * 16) * sizeof(float);
}

static void fill_bwd(BwdArgs& a, const gnnrag_csr* csr, int D, int I) {
  memset(&a, 0, sizeof(a));
  for (int d = 0; d < 2; ++d) {
    a.row_ptr[d] = csr->row_ptr[d];
    a.edge[d] = (const int2*)csr->edge[d];
    a.edge_l[d] = (const int2*)csr->edge_l[d];
    a.heavy[d] = csr->heavy[d];
    a.chunk_off[d] = csr->chunk_off[d];
  }
  a.n_heavy = csr->n_heavy;
  a.n_chunks = csr->n_chunks;
  a.heavy_cap = csr->heavy_cap;
  a.max_chunks = csr->max_chunks;
  a.heavy_deg = csr->heavy_deg;
  a.rel_off = csr->rel_off;
  a.rel_rows = (const int2*)csr->rel_rows;
  a.big_cnt = csr->big_cnt;
  a.big_nodes = csr->big_nodes;
  a.big_deg = csr->big_deg;
  a.B = csr->B;
  a.N = csr->N;
  a.D = D;
  a.I = I;
  a.Rmax = csr->rel_max;
}

template <int MODE>
static int launch_tables(const BwdArgs& a, const gnnrag_csr* csr, hipStream_t stream) {
  const int nd = MODE == BWD_REASON ? 2 : 1;
  const size_t lds = bwd_lds_bytes(nd, csr->rel_max);
  if (lds > 160 * 1024 - 1024) return GNNRAG_E_UNSUPPORTED;   // a question uses too many relations for one CU's LDS
  const int nslice = (a.D + kBwdSliceW - 1) / kBwdSliceW;
  const int ni = MODE == BWD_REASON ? a.I : 1;
  const int nblk = 8 * ((csr->B + 7) / 8) * nslice * ni;
  const bool v4 = a.D % 4 == 0;
  static bool attr_set = false;
  if (!attr_set) {
    GNNRAG_HIP(hipFuncSetAttribute((const void*)k_bwd_tables<MODE, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GNNRAG_HIP(hipFuncSetAttribute((const void*)k_bwd_tables<MODE, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (v4) hipLaunchKernelGGL((k_bwd_tables<MODE, true>), dim3(nblk), dim3(kBwdThreads), lds, stream, a, nslice);
  else hipLaunchKernelGGL((k_bwd_tables<MODE, false>), dim3(nblk), dim3(kBwdThreads), lds, stream, a, nslice);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_aggregate_backward(const gnnrag_csr* csr, const float* dist, const float* ins,
                                         const float* T_fwd, const float* T_inv, const float* g_agg,
                                         float* g_dist, float* g_ins, float* g_T_fwd, float* g_T_inv,
                                         int32_t D, int32_t I, gnnrag_stream_t stream_) {
  if (!csr || !dist || !ins || !T_fwd || !T_inv || !g_agg || !g_dist || !g_ins || !g_T_fwd || !g_T_inv ||
      D <= 0 || I <= 0 || csr->rel_total < 0)
    return GNNRAG_E_BADARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (bwd_lds_bytes(2, csr->rel_max) > 160 * 1024 - 1024) return GNNRAG_E_UNSUPPORTED;
  const size_t tbytes = (size_t)csr->R1 * D * sizeof(float);
  GNNRAG_HIP(hipMemsetAsync(g_T_fwd, 0, tbytes, stream));
  GNNRAG_HIP(hipMemsetAsync(g_T_inv, 0, tbytes, stream));
  BwdArgs a;
  fill_bwd(a, csr, D, I);
  a.w[0] = csr->w_gnn[0];
  a.w[1] = csr->w_gnn[1];
  a.dist = dist;
  a.ins = ins;
  a.T[0] = T_fwd;
  a.T[1] = T_inv;
  a.g = g_agg;
  a.g_dist = g_dist;
  a.g_ins = g_ins;
  a.g_T[0] = g_T_fwd;
  a.g_T[1] = g_T_inv;
  hipLaunchKernelGGL(k_bwd_prior, dim3((csr->N + 3) / 4, csr->B), dim3(256), (size_t)I * D * sizeof(float), stream, a);
  GNNRAG_LAUNCH_CHECK();
  if (csr->F > 0) {
    const int nb = csr->max_chunks < 4096 ? (csr->max_chunks + 3) / 4 : 1024;
    hipLaunchKernelGGL(k_bwd_prior_heavy, dim3(nb > 0 ? nb : 1, 2), dim3(256), 0, stream, a);
    GNNRAG_LAUNCH_CHECK();
  }
  return launch_tables<BWD_REASON>(a, csr, stream);
}

extern "C" int gnnrag_typelayer_backward(const gnnrag_csr* csr, const float* g_pre, int use_w_rel, float* g_T,
                                         int32_t D, gnnrag_stream_t stream_) {
  if (!csr || !g_pre || !g_T || D <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  if (use_w_rel && (!csr->w_rel[0] || !csr->w_rel[1])) return GNNRAG_E_BADARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (bwd_lds_bytes(1, csr->rel_max) > 160 * 1024 - 1024) return GNNRAG_E_UNSUPPORTED;
  GNNRAG_HIP(hipMemsetAsync(g_T, 0, (size_t)csr->R1 * D * sizeof(float), stream));
  BwdArgs a;
  fill_bwd(a, csr, D, 1);
  a.w[0] = use_w_rel ? csr->w_rel[0] : nullptr;
  a.w[1] = use_w_rel ? csr->w_rel[1] : nullptr;
  a.g = g_pre;
  a.g_T[0] = g_T;
  return launch_tables<BWD_TYPE>(a, csr, stream);
}
