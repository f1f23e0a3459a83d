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
// Kernels:
//  * k_bwd_prior: the facts with source s in direction d are row s of the OTHER direction's structure
//    (its records hold (dst_d(f), rel_f)), so g_dist is a gather - one wave per node, lanes across the
//    D columns, one cross-lane reduction per node; rows above heavy_deg go to one workgroup per 256-fact
//    chunk (k_bwd_prior_heavy), pieces added in chunk order by k_bwd_prior_heavy_reduce.
//  * gather form of U (default): k_bwd_rel_gather / k_bwd_type_gather over the facts ordered by (question,
//    relation) (gnnrag_relorder, csr_plan.hip), one wave per chunk of a relation row, partial sums per chunk,
//    then k_bwd_reduce_tables_chunks / k_bwd_reduce_ins_chunks in a fixed order.  No atomics.
//  * LDS form of U (fallback for D % 4 != 0): k_bwd_tables - U is the transpose of the fused forward's
//    relation tables: a workgroup owns (question, 16-column slice), keeps U[2][relations the question
//    uses][16] in LDS, walks the question's nodes like the forward LDS walk (over (p_f, relation) pairs made
//    once per call by k_bwd_pairs) but ADDS p * g_agg[n, cols] into the row of the fact's relation
//    (ds_add_f32); the epilogue applies the ReLU gate and writes g_ins and the question's own gradient rows
//    V[d][compact row]; k_bwd_reduce_tables sums V over the questions that use a relation (global atomics
//    across XCDs go to the memory side and were no faster).  LDS sums are in atomic order.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "gnnrag_common.h"

namespace gnnrag {

constexpr int kBwdSliceW = 16;
constexpr int kBwdThreads = 1024;
constexpr int kBwdTeamDeg = 128;    // rows with more facts are bucketed by the whole workgroup

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
  int32_t B, N, D, I, Rmax, Rtot;
};

// facts [j0, j1) of structure o = 1 - d, all with the same source: this lane's share of
//   sum_f w_f sum_i < g_agg[dst_f, 2i+d, :], relu(T_d[rel_f,:] * q_i) >.
// V4: the lane owns float4 columns lane, lane + 64, ... (D % 4 == 0); else scalar columns lane, lane + 64, ...
// Two facts per step so their row loads overlap.
// FUSED (backward of the fused walk, gnnrag_aggregate_fused_backward): the message of a fact is row
// rel_off[b] + (compact relation) of the question's own table P[d] (a.T[d]; `q` then carries that row offset in its
// ADDRESS-free form `roff`), nothing is multiplied by an instruction or gated, g is [BN, D] for both directions.
template <bool V4, bool FUSED = false>
__device__ __forceinline__ float prior_grad_range(const BwdArgs& a, const float* __restrict__ q, int d, int j0,
                                                  int j1, int lane, int roff = 0) {
  const int o = 1 - d;
  const int D = a.D, I = FUSED ? 1 : a.I;
  const size_t ld = FUSED ? (size_t)D : (size_t)2 * I * D;
  const int2* __restrict__ edge = FUSED ? a.edge_l[o] : a.edge[o];
  const float* __restrict__ w = a.w[o];
  const float* __restrict__ T = a.T[d] + (FUSED ? (size_t)roff * D : 0);
  const float* __restrict__ g = FUSED ? a.g : a.g + (size_t)d * D;
  constexpr int W = V4 ? 4 : 1;
  typedef float vec __attribute__((ext_vector_type(W)));
  auto dot2 = [&](const float* t0, const float* g0, const float* t1, const float* g1, float& p0, float& p1) {
    for (int c = W * lane; c < D; c += 64 * W) {
      const vec tv0 = *reinterpret_cast<const vec*>(t0 + c);
      const vec tv1 = *reinterpret_cast<const vec*>(t1 + c);
      for (int i = 0; i < I; ++i) {
        const vec gv0 = *reinterpret_cast<const vec*>(g0 + (size_t)2 * i * D + c);
        const vec gv1 = *reinterpret_cast<const vec*>(g1 + (size_t)2 * i * D + c);
        vec m0 = tv0, m1 = tv1;
        if constexpr (!FUSED) {
          const vec qv = *reinterpret_cast<const vec*>(q + i * D + c);
          m0 = __builtin_elementwise_max(tv0 * qv, (vec)0.f);
          m1 = __builtin_elementwise_max(tv1 * qv, (vec)0.f);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
          p0 += gv0[k] * m0[k];
          p1 += gv1[k] * m1[k];
        }
      }
    }
  };
  float acc = 0.f;
  int j = j0;
  for (; j + 1 < j1; j += 2) {
    const int2 e0 = edge[j], e1 = edge[j + 1];
    const float w0 = w ? w[j] : 1.f, w1 = w ? w[j + 1] : 1.f;
    float p0 = 0.f, p1 = 0.f;
    dot2(T + (size_t)e0.y * D, g + (size_t)e0.x * ld, T + (size_t)e1.y * D, g + (size_t)e1.x * ld, p0, p1);
    acc += w0 * p0 + w1 * p1;
  }
  if (j < j1) {
    const int2 e0 = edge[j];
    const float w0 = w ? w[j] : 1.f;
    float p0 = 0.f, p1 = 0.f;
    const float* t0 = T + (size_t)e0.y * D;
    const float* g0 = g + (size_t)e0.x * ld;
    dot2(t0, g0, t0, g0, p0, p1);
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
template <bool V4, bool FUSED = false>
__global__ __launch_bounds__(256) void k_bwd_prior(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float q_s[];      // [I][D]
  const int b = blockIdx.y;
  if constexpr (!FUSED) {
    for (int x = threadIdx.x; x < a.I * a.D; x += 256) q_s[x] = a.ins[(size_t)b * a.I * a.D + x];
    __syncthreads();
  }
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
    acc += prior_grad_range<V4, FUSED>(a, q_s, d, beg, end, lane, FUSED ? a.rel_off[b] : 0);
  }
  acc = wave_sum(acc);
  if (lane == 0) a.g_dist[s] = acc;
}

// rows above heavy_deg: one workgroup per 256-fact chunk (64 facts per wave) -> part[o][chunk][wave];
// k_bwd_prior_heavy_reduce adds a row's pieces in order onto k_bwd_prior's store (no atomics)
template <bool V4, bool FUSED = false>
__global__ __launch_bounds__(256) void k_bwd_prior_heavy(const BwdArgs a, float* __restrict__ part) {
  const int o = blockIdx.y, d = 1 - o;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cnt = min(a.n_heavy[o], a.heavy_cap);
  const int nch = min(a.n_chunks[o], a.max_chunks);
  const int32_t* off = a.chunk_off[o];
  for (int c = blockIdx.x; c < nch; c += gridDim.x) {
    int lo = 0, hi = cnt;                       // largest e with off[e] <= c
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= c) lo = mid; else hi = mid;
    }
    const int s = a.heavy[o][lo];
    const int cbeg = a.row_ptr[o][s] + (c - off[lo]) * kHeavyDeg;
    const int cend = min(cbeg + kHeavyDeg, a.row_ptr[o][s + 1]);
    const int beg = min(cbeg + wave * (kHeavyDeg / 4), cend);
    const int end = min(beg + kHeavyDeg / 4, cend);
    const float* q = FUSED ? nullptr : a.ins + (size_t)(s / a.N) * a.I * a.D;
    const float acc = wave_sum(prior_grad_range<V4, FUSED>(a, q, d, beg, end, lane, FUSED ? a.rel_off[s / a.N] : 0));
    if (lane == 0) part[((size_t)o * a.max_chunks + c) * 4 + wave] = acc;
  }
}

// one thread per heavy row of structure o: g_dist[s] += its chunk pieces, in chunk order.  Launched for
// o = 0 and then o = 1 (a node can be heavy in both structures).
__global__ __launch_bounds__(256) void k_bwd_prior_heavy_reduce(const BwdArgs a, const float* __restrict__ part, int o) {
  const int cnt = min(a.n_heavy[o], a.heavy_cap);
  const int nch = min(a.n_chunks[o], a.max_chunks);
  const int32_t* off = a.chunk_off[o];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256) {
    float t = 0.f;
    const int c1 = min(off[e + 1], nch);
    for (int c = off[e]; c < c1; ++c) {
      const float* p = part + ((size_t)o * a.max_chunks + c) * 4;
      t += ((p[0] + p[1]) + p[2]) + p[3];
    }
    a.g_dist[a.heavy[o][e]] += t;
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

// (p_f, compact relation) per fact and direction, p_f = w_f * dist[src_d(f)] (REASON) or v_f (TYPE):
// computed once per backward call, so the bucketing loop below has no dependent gather in it
template <int MODE>
__global__ __launch_bounds__(256) void k_bwd_pairs(const BwdArgs a, int64_t F, int2* __restrict__ pr) {
  const int d = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= F) return;
  const int2 e = a.edge_l[d][j];
  float p = a.w[d] ? a.w[d][j] : 1.f;
  if constexpr (MODE == BWD_REASON) p *= a.dist[e.x];
  pr[(size_t)d * F + j] = make_int2(__float_as_int(p), e.y);
}

// facts first, first+stride, ... of the row [beg, end): U[rel][col] += p * gv, four facts per step.
// One lane per column (16 lanes per fact): a wave's ds_add_f32 then covers 16 consecutive banks per fact
// and 4 facts - with float4-per-lane ownership the 64 lanes of an instruction fell on 8 banks.
__device__ __forceinline__ void bucket_row(const int2* __restrict__ pr, float* __restrict__ Ud, int beg, int end,
                                           int first, int stride, int sub, float gv) {
  for (int j = beg + first; j < end; j += 4 * stride) {
    int2 e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jj = j + u * stride;
      e[u] = jj < end ? pr[jj] : make_int2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float p = __int_as_float(e[u].x);
      if (p != 0.f) unsafeAtomicAdd(Ud + (size_t)e[u].y * kBwdSliceW + sub, p * gv);
    }
  }
}

template <int MODE, bool V4>
__global__ __launch_bounds__(kBwdThreads) void k_bwd_tables(const BwdArgs a, const int2* __restrict__ pr, int64_t F,
                                                            float* __restrict__ V, int nslice) {
  constexpr int ND = (MODE == BWD_REASON) ? 2 : 1;     // TYPE: both directions read the same table
  extern __shared__ __attribute__((aligned(16))) float s_u[];   // U [ND][Rg][16], then red [16 waves][16]
  float* red = s_u + (size_t)ND * a.Rmax * kBwdSliceW;
  const int NI = (MODE == BWD_REASON) ? a.I : 1;
  // XCD-aware order: the workgroups of question g land on XCD g % 8 (they share its CSR rows in L2)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int g = (slot / nslice) * 8 + xcd;
  if (g >= a.B) return;
  const int c = slot % nslice;
  const int col0 = c * kBwdSliceW;
  const int D = a.D, N = a.N;
  const int roff = a.rel_off[g], Rg = a.rel_off[g + 1] - roff;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, sub = lane & 15;          // 16 lanes per node: one lane per slice column
  const int c0 = col0 + sub;
  const bool cok = c0 < D;
  const size_t ld = (MODE == BWD_REASON) ? (size_t)2 * a.I * D : (size_t)D;
  const int nbig = a.big_cnt[g];
  // epilogue ownership: thread t holds float4 granule k = t & 3 of the slice for rows t/4, t/4 + 256, ...
  const int k = tid & 3;
  const int ce = col0 + 4 * k;
  const size_t vdir = (size_t)a.Rtot * D;              // V [ND][Rtot][D]: per-question gradient rows

  for (int i = 0; i < NI; ++i) {
    for (int x = tid; x < ND * Rg * kBwdSliceW; x += kBwdThreads) s_u[x] = 0.f;
    __syncthreads();
    // nodes with many facts (listed at plan time).  Hubs (> kBwdTeamDeg facts in a direction): the whole
    // workgroup per node, its 64 lane groups take facts k, k+64, ... (adds commute: no reduction step);
    // the others: one wave per node, lane group k takes facts k, k+4, ...
    for (int h = 0; h < nbig; ++h) {
      const int n = a.big_nodes[(size_t)g * N + h];
      const int b0 = a.row_ptr[0][n], e0 = a.row_ptr[0][n + 1], b1 = a.row_ptr[1][n], e1 = a.row_ptr[1][n + 1];
      if (max(e0 - b0, e1 - b1) <= kBwdTeamDeg) continue;          // workgroup-uniform
      const float* grow = a.g + (size_t)n * ld + c0;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float gv = cok ? grow[MODE == BWD_REASON ? (size_t)(2 * i + d) * D : 0] : 0.f;
        bucket_row(pr + (size_t)d * F, s_u + (size_t)(ND == 2 ? d : 0) * Rg * kBwdSliceW, d ? b1 : b0,
                   d ? e1 : e0, tid >> 4, kBwdThreads / 16, sub, gv);
      }
    }
    for (int h = wave; h < nbig; h += 16) {
      const int n = a.big_nodes[(size_t)g * N + h];
      const int b0 = a.row_ptr[0][n], e0 = a.row_ptr[0][n + 1], b1 = a.row_ptr[1][n], e1 = a.row_ptr[1][n + 1];
      if (max(e0 - b0, e1 - b1) > kBwdTeamDeg) continue;
      const float* grow = a.g + (size_t)n * ld + c0;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float gv = cok ? grow[MODE == BWD_REASON ? (size_t)(2 * i + d) * D : 0] : 0.f;
        bucket_row(pr + (size_t)d * F, s_u + (size_t)(ND == 2 ? d : 0) * Rg * kBwdSliceW, d ? b1 : b0,
                   d ? e1 : e0, grp, 4, sub, gv);
      }
    }
    // everything else: a 16-lane group per node
    for (int nl = tid >> 4; nl < N; nl += kBwdThreads / 16) {
      const int n = g * N + nl;
      const int b0 = a.row_ptr[0][n], e0 = a.row_ptr[0][n + 1], b1 = a.row_ptr[1][n], e1 = a.row_ptr[1][n + 1];
      const int l0 = e0 - b0, l1 = e1 - b1;
      if (max(l0, l1) > a.big_deg || (l0 | l1) == 0) continue;
      const float* grow = a.g + (size_t)n * ld + c0;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float gv = cok ? grow[MODE == BWD_REASON ? (size_t)(2 * i + d) * D : 0] : 0.f;
        bucket_row(pr + (size_t)d * F, s_u + (size_t)(ND == 2 ? d : 0) * Rg * kBwdSliceW, d ? b1 : b0,
                   d ? e1 : e0, 0, 1, sub, gv);
      }
    }
    __syncthreads();

    // epilogue of instruction i.  The rows of V this workgroup owns (question g, its slice) are touched by
    // nobody else and by the same thread for every i: plain read-modify-write, no atomics.
    if constexpr (MODE == BWD_TYPE) {
      for (int idx = tid; idx < Rg * 4; idx += kBwdThreads) {
        const int r = idx >> 2;
        const float* u = s_u + (size_t)r * kBwdSliceW + 4 * k;
        float* v = V + (size_t)(roff + r) * D + ce;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ce + e < D) v[e] = u[e];
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
        float* v = V + (size_t)d * vdir + (size_t)(roff + r) * D + ce;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float add = 0.f;
          if (t[e] * q[e] > 0.f) {                       // the ReLU gate of relu(T_d[r] * ins[b,i])
            gq[e] += u[e] * t[e];
            add = u[e] * q[e];
          }
          if (ce + e < D) v[e] = (i == 0) ? add : v[e] + add;
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
    __syncthreads();                                   // U and red are reused by the next instruction
  }
}

// g_T_d[r,:] = sum over the questions that use relation r of their row of V: one workgroup per
// (relation, direction); the rows are found by binary search in each question's sorted relation list.
// Fixed order over questions: deterministic given V.
__global__ __launch_bounds__(256) void k_bwd_reduce_tables(const BwdArgs a, const float* __restrict__ V, int nd) {
  __shared__ int rows[256];
  const int r = blockIdx.x, d = blockIdx.y;
  const int D = a.D;
  const float* Vd = V + (size_t)d * a.Rtot * D;
  float* out = a.g_T[d] + (size_t)r * D;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                // columns tid, tid + 256, ... (D <= 1024)
  for (int b0 = 0; b0 < a.B; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    int row = -1;
    if (b < a.B) {
      int lo = a.rel_off[b], hi = a.rel_off[b + 1];   // first row with relation >= r
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.rel_rows[mid].y < r) lo = mid + 1; else hi = mid;
      }
      if (lo < a.rel_off[b + 1] && a.rel_rows[lo].y == r) row = lo;
    }
    rows[threadIdx.x] = row;
    __syncthreads();
    const int nb = min(256, a.B - b0);
    for (int j = 0; j < nb; ++j) {
      const int rw = rows[j];
      if (rw < 0) continue;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = (int)threadIdx.x + 256 * m;
        if (c < D) acc[m] += Vd[(size_t)rw * D + c];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = (int)threadIdx.x + 256 * m;
    if (c < D) out[c] = acc[m];
  }
}

// ---- table / instruction gradients by gathering over (question, relation) rows -------------------
// One wave per chunk (<= 256 facts of one compact relation row, gnnrag_relorder).  All facts of a row
// share T_d[r] and the question's instructions, so the wave keeps U[d][i] (its float4 columns) in
// registers while it streams the facts: for each fact and direction one scalar prior and NI coalesced
// row segments of g_agg.  The ReLU gate depends on (r, b, i) only, so a chunk's partial sums are gated
// and written on their own:  Vc[chunk][d][:] = sum_i U[d][i]*gate*q_i   (-> g_T_d[r] over chunks),
//                            Qc[chunk][i][:] = sum_d U[d][i]*gate*t_d   (-> g_ins[b,i] over chunks).
// No atomics: every output element has one writer and the reductions run in a fixed order.
// FUSED (backward of the fused walk): g is [BN, D] for both directions and the chunk's raw sums are the result -
// Vc[chunk][d][:] = U[d] = sum_f w_f dist[src_d(f)] g[dst_d(f), :], the chunk's share of g_P[d][row]; no gate, no Qc.
template <int NI, int CPL, bool FUSED = false>
__global__ __launch_bounds__(256) void k_bwd_rel_gather(const BwdArgs a, const int2* __restrict__ ht,
                                                        const float* __restrict__ w,
                                                        const int32_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ chunk_ptr, int n_chunks,
                                                        float* __restrict__ Vc, float* __restrict__ Qc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD b % 8 gets a contiguous eighth of the chunks (= a few questions): their g_agg rows share one L2
  const int nblk = gridDim.x;
  const int blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int c = blk * 4 + wave;
  if (c >= n_chunks) return;
  int lo = 0, hi = a.Rtot;                      // row of chunk c: largest row with chunk_ptr[row] <= c
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_ptr[mid] <= c) lo = mid; else hi = mid;
  }
  const int row = lo;
  const int2 br = a.rel_rows[row];              // (question, relation id)
  const int beg = row_ptr[row] + (c - chunk_ptr[row]) * kHeavyDeg;
  const int end = min(beg + kHeavyDeg, row_ptr[row + 1]);
  const int D = a.D;
  const size_t ld = FUSED ? (size_t)D : (size_t)2 * NI * D;
  f32x4 U[2][NI][CPL];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int m = 0; m < CPL; ++m) U[d][i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool cv[CPL];
  int col[CPL];
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    col[m] = 4 * (lane + 64 * m);
    cv[m] = col[m] < D;
  }
  for (int j0 = beg; j0 < end; j0 += 4) {
    int2 e[4];
    float wf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, end - 1);
      e[u] = ht[j];
      wf[u] = (j0 + u < end) ? (w ? w[j] : 1.f) : 0.f;
    }
    float p[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      p[u][0] = wf[u] * a.dist[e[u].x];         // forward: prior of the head, gradient row of the tail
      p[u][1] = wf[u] * a.dist[e[u].y];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float pv = p[u][d];
        if (pv == 0.f) continue;                // wave-uniform
        const float* grow = a.g + (size_t)(d == 0 ? e[u].y : e[u].x) * ld + (FUSED ? 0 : (size_t)d * D);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int m = 0; m < CPL; ++m)
            if (cv[m]) U[d][i][m] += pv * *reinterpret_cast<const f32x4*>(grow + (size_t)2 * i * D + col[m]);
      }
    }
  }
  float* vout = Vc + (size_t)c * 2 * D;
  if constexpr (FUSED) {
#pragma unroll
    for (int m = 0; m < CPL; ++m)
      if (cv[m]) {
#pragma unroll
        for (int d = 0; d < 2; ++d) *reinterpret_cast<f32x4*>(vout + (size_t)d * D + col[m]) = U[d][0][m];
      }
    return;
  }
  float* qout = Qc + (size_t)c * NI * D;
  const float* qin = a.ins + (size_t)br.x * NI * D;
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    if (!cv[m]) continue;
    f32x4 t[2], q[NI];
#pragma unroll
    for (int d = 0; d < 2; ++d) t[d] = *reinterpret_cast<const f32x4*>(a.T[d] + (size_t)br.y * D + col[m]);
#pragma unroll
    for (int i = 0; i < NI; ++i) q[i] = *reinterpret_cast<const f32x4*>(qin + (size_t)i * D + col[m]);
    f32x4 v[2], gq[NI];
#pragma unroll
    for (int d = 0; d < 2; ++d) v[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) gq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t[d][k] * q[i][k] > 0.f) {        // the ReLU gate of relu(T_d[r] * ins[b,i])
            v[d][k] += U[d][i][m][k] * q[i][k];
            gq[i][k] += U[d][i][m][k] * t[d][k];
          }
#pragma unroll
    for (int d = 0; d < 2; ++d) *reinterpret_cast<f32x4*>(vout + (size_t)d * D + col[m]) = v[d];
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(qout + (size_t)i * D + col[m]) = gq[i];
  }
}

// TypeLayer: Vc[chunk][0][:] = sum over the chunk's facts of v_f (g_pre[tail_f,:] + g_pre[head_f,:])
template <int CPL>
__global__ __launch_bounds__(256) void k_bwd_type_gather(const BwdArgs a, const int2* __restrict__ ht,
                                                         const int32_t* __restrict__ perm,
                                                         const float* __restrict__ w_fact,
                                                         const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ chunk_ptr, int n_chunks,
                                                         float* __restrict__ Vc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = gridDim.x;
  const int blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int c = blk * 4 + wave;
  if (c >= n_chunks) return;
  int lo = 0, hi = a.Rtot;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_ptr[mid] <= c) lo = mid; else hi = mid;
  }
  const int beg = row_ptr[lo] + (c - chunk_ptr[lo]) * kHeavyDeg;
  const int end = min(beg + kHeavyDeg, row_ptr[lo + 1]);
  const int D = a.D;
  f32x4 U[CPL];
#pragma unroll
  for (int m = 0; m < CPL; ++m) U[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int j0 = beg; j0 < end; j0 += 4) {
    int2 e[4];
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, end - 1);
      e[u] = ht[j];
      v[u] = (j0 + u < end) ? (w_fact ? w_fact[perm[j]] : 1.f) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (v[u] == 0.f) continue;
#pragma unroll
      for (int m = 0; m < CPL; ++m) {
        const int col = 4 * (lane + 64 * m);
        if (col < D)
          U[m] += v[u] * (*reinterpret_cast<const f32x4*>(a.g + (size_t)e[u].y * D + col) +
                          *reinterpret_cast<const f32x4*>(a.g + (size_t)e[u].x * D + col));
      }
    }
  }
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    const int col = 4 * (lane + 64 * m);
    if (col < D) *reinterpret_cast<f32x4*>(Vc + (size_t)c * 2 * D + col) = U[m];
  }
}

// g_T_d[r,:] = sum over the questions that use r (found by binary search in each question's sorted relation
// list) and the chunks of their row of Vc.  One workgroup per (relation, direction); thread group k of 4 takes
// the k-th, (k+4)-th, ... question, the four partial sums are combined in group order (fixed order).
__global__ __launch_bounds__(1024) void k_bwd_reduce_tables_chunks(const BwdArgs a, const float* __restrict__ Vc,
                                                                   const int32_t* __restrict__ chunk_ptr) {
  __shared__ int cbs[256], ces[256];
  __shared__ float part[4][256];
  const int r = blockIdx.x, d = blockIdx.y;
  const int D = a.D;
  const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
  for (int col0 = 0; col0 < D; col0 += 256) {
    const int col = col0 + t;
    float acc = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += 256) {
      __syncthreads();
      if (threadIdx.x < 256) {                  // thread b: chunk range of row (b, r), empty if b does not use r
        int cb = 0, ce = 0;
        const int b = b0 + (int)threadIdx.x;
        if (b < a.B) {
          int lo = a.rel_off[b], hi = a.rel_off[b + 1];
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.rel_rows[mid].y < r) lo = mid + 1; else hi = mid;
          }
          if (lo < a.rel_off[b + 1] && a.rel_rows[lo].y == r) {
            cb = chunk_ptr[lo];
            ce = chunk_ptr[lo + 1];
          }
        }
        cbs[threadIdx.x] = cb;
        ces[threadIdx.x] = ce;
      }
      __syncthreads();
      if (col < D)
        for (int j = grp; j < 256; j += 4)
          for (int c = cbs[j]; c < ces[j]; ++c) acc += Vc[((size_t)c * 2 + d) * D + col];
    }
    part[grp][t] = acc;
    __syncthreads();
    if (grp == 0 && col < D) a.g_T[d][(size_t)r * D + col] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
  }
}

// g_ins[b,i,:] = sum over the chunks of question b of Qc: one workgroup per (question, instruction), four
// thread groups over the chunks, combined in group order
__global__ __launch_bounds__(1024) void k_bwd_reduce_ins_chunks(const BwdArgs a, const float* __restrict__ Qc,
                                                                const int32_t* __restrict__ chunk_ptr) {
  __shared__ float part[4][256];
  const int b = blockIdx.x, i = blockIdx.y;
  const int D = a.D, NI = a.I;
  const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int c0 = chunk_ptr[a.rel_off[b]], c1 = chunk_ptr[a.rel_off[b + 1]];
  for (int col0 = 0; col0 < D; col0 += 256) {
    const int col = col0 + t;
    float acc0 = 0.f, acc1 = 0.f;
    if (col < D) {
      int c = c0 + grp;
      for (; c + 4 < c1; c += 8) {
        acc0 += Qc[((size_t)c * NI + i) * D + col];
        acc1 += Qc[((size_t)(c + 4) * NI + i) * D + col];
      }
      if (c < c1) acc0 += Qc[((size_t)c * NI + i) * D + col];
    }
    part[grp][t] = acc0 + acc1;
    __syncthreads();
    if (grp == 0 && col < D)
      a.g_ins[((size_t)b * NI + i) * D + col] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
    __syncthreads();
  }
}

// fused walk: g_P[d][row][:] = sum over the row's chunks of Vc, in chunk order (one workgroup per compact row)
__global__ __launch_bounds__(256) void k_bwd_reduce_rows_chunks(const float* __restrict__ Vc,
                                                                const int32_t* __restrict__ chunk_ptr, int Rtot, int D,
                                                                float* __restrict__ g_P) {
  const int row = blockIdx.x;
  const int c0 = chunk_ptr[row], c1 = chunk_ptr[row + 1];
  for (int x = threadIdx.x; x < 2 * D; x += 256) {
    const int d = x >= D, col = x - d * D;
    float acc = 0.f;
    for (int c = c0; c < c1; ++c) acc += Vc[((size_t)c * 2 + d) * D + col];
    g_P[((size_t)d * Rtot + row) * D + col] = acc;
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
  a.Rtot = csr->rel_total;
}

static size_t bwd_pairs_bytes(const gnnrag_csr* csr) {
  return align_up((size_t)2 * (size_t)(csr->F > 0 ? csr->F : 1) * sizeof(int2), 256);
}
// (p, relation) pairs of both directions + V [2][rel_total][D]
static size_t bwd_ws_bytes(const gnnrag_csr* csr, int D) {
  return bwd_pairs_bytes(csr) +
         align_up((size_t)2 * (size_t)(csr->rel_total > 0 ? csr->rel_total : 1) * D * sizeof(float), 256);
}
// gather path: Vc [n_chunks][2][D] + Qc [n_chunks][I][D]
static size_t bwd_gather_ws_bytes(const gnnrag_relorder* ro, int D, int I) {
  return align_up((size_t)(ro->n_chunks > 0 ? ro->n_chunks : 1) * (size_t)(2 + I) * D * sizeof(float), 256);
}
static bool gather_ok(const gnnrag_relorder* ro, int D, int I) {
  return ro && D % 4 == 0 && D <= 1024 && I >= 1 && I <= 4;
}

template <int NI>
static int launch_gather(const BwdArgs& a, const gnnrag_relorder* ro, float* Vc, float* Qc, hipStream_t stream) {
  const int nblk = 8 * ((ro->n_chunks + 31) / 32);          // 4 chunks per workgroup, multiple of 8 workgroups
  const int cpl = (a.D / 4 + 63) / 64;
#define GNNRAG_GATHER(C)                                                                                       \
  hipLaunchKernelGGL((k_bwd_rel_gather<NI, C>), dim3(nblk), dim3(256), 0, stream, a, (const int2*)ro->ht,    \
                     (const float*)ro->w, (const int32_t*)ro->row_ptr, (const int32_t*)ro->chunk_ptr,        \
                     ro->n_chunks, Vc, Qc)
  if (cpl == 1) GNNRAG_GATHER(1);
  else if (cpl == 2) GNNRAG_GATHER(2);
  else GNNRAG_GATHER(4);
#undef GNNRAG_GATHER
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

template <int MODE>
static int launch_tables(const BwdArgs& a, const gnnrag_csr* csr, void* ws, size_t ws_bytes, hipStream_t stream) {
  const int nd = MODE == BWD_REASON ? 2 : 1;
  if (!ws || ws_bytes < bwd_ws_bytes(csr, a.D)) return GNNRAG_E_WORKSPACE;
  if (a.D > 1024) return GNNRAG_E_UNSUPPORTED;
  int2* pr = (int2*)ws;
  float* V = (float*)((char*)ws + bwd_pairs_bytes(csr));
  const int64_t F = csr->F;
  if (F > 0) {
    hipLaunchKernelGGL(k_bwd_pairs<MODE>, dim3((unsigned)((F + 255) / 256), 2), dim3(256), 0, stream, a, F, pr);
    GNNRAG_LAUNCH_CHECK();
  }
  const size_t lds = bwd_lds_bytes(nd, csr->rel_max);
  if (lds > 160 * 1024 - 1024) return GNNRAG_E_UNSUPPORTED;   // a question uses too many relations for one CU's LDS
  const int nslice = (a.D + kBwdSliceW - 1) / kBwdSliceW;
  const int nblk = 8 * ((csr->B + 7) / 8) * nslice;
  const bool v4 = a.D % 4 == 0;
  static DeviceMask cap_v4, cap_v1;   // per instantiation, per device
  {
    int rc = raise_lds_cap(k_bwd_tables<MODE, true>, cap_v4);
    if (rc) return rc;
    rc = raise_lds_cap(k_bwd_tables<MODE, false>, cap_v1);
    if (rc) return rc;
  }
  if (v4) hipLaunchKernelGGL((k_bwd_tables<MODE, true>), dim3(nblk), dim3(kBwdThreads), lds, stream, a,
                             (const int2*)pr, F, V, nslice);
  else hipLaunchKernelGGL((k_bwd_tables<MODE, false>), dim3(nblk), dim3(kBwdThreads), lds, stream, a,
                          (const int2*)pr, F, V, nslice);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd_reduce_tables, dim3(csr->R1, nd), dim3(256), 0, stream, a, (const float*)V, nd);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_aggregate_backward(const gnnrag_csr* csr, const gnnrag_relorder* relorder,
                                         const float* dist, const float* ins,
                                         const float* T_fwd, const float* T_inv, const float* g_agg,
                                         float* g_dist, float* g_ins, float* g_T_fwd, float* g_T_inv,
                                         int32_t D, int32_t I, void* workspace, size_t workspace_bytes,
                                         gnnrag_stream_t stream_) {
  if (!csr || !dist || !ins || !T_fwd || !T_inv || !g_agg || !g_dist || !g_ins || !g_T_fwd || !g_T_inv ||
      D <= 0 || I <= 0 || csr->rel_total < 0)
    return GNNRAG_E_BADARG;
  hipStream_t stream = (hipStream_t)stream_;
  const bool gather = gather_ok(relorder, D, I);
  if (relorder && (relorder->F != csr->F || relorder->rel_total != csr->rel_total)) return GNNRAG_E_BADARG;
  if (!gather && bwd_lds_bytes(2, csr->rel_max) > 160 * 1024 - 1024) return GNNRAG_E_UNSUPPORTED;
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
  const bool v4 = D % 4 == 0 && ((uintptr_t)g_agg & 15) == 0 && ((uintptr_t)T_fwd & 15) == 0 &&
                  ((uintptr_t)T_inv & 15) == 0 && ((uintptr_t)ins & 15) == 0;
  const dim3 pgrid((csr->N + 3) / 4, csr->B);
  const size_t plds = (size_t)I * D * sizeof(float);
  if (v4) hipLaunchKernelGGL(k_bwd_prior<true>, pgrid, dim3(256), plds, stream, a);
  else hipLaunchKernelGGL(k_bwd_prior<false>, pgrid, dim3(256), plds, stream, a);
  GNNRAG_LAUNCH_CHECK();
  if (csr->F > 0) {
    // the heavy rows' pieces borrow the head of the workspace: they are consumed before the table-gradient
    // kernels (same stream) write there
    const size_t hbytes = (size_t)2 * csr->max_chunks * 4 * sizeof(float);
    if (!workspace || workspace_bytes < hbytes) return GNNRAG_E_WORKSPACE;
    float* part = (float*)workspace;
    const int nb = csr->max_chunks < 4096 ? csr->max_chunks : 4096;
    if (v4) hipLaunchKernelGGL(k_bwd_prior_heavy<true>, dim3(nb > 0 ? nb : 1, 2), dim3(256), 0, stream, a, part);
    else hipLaunchKernelGGL(k_bwd_prior_heavy<false>, dim3(nb > 0 ? nb : 1, 2), dim3(256), 0, stream, a, part);
    GNNRAG_LAUNCH_CHECK();
    for (int o = 0; o < 2; ++o) {
      hipLaunchKernelGGL(k_bwd_prior_heavy_reduce, dim3((csr->heavy_cap + 255) / 256 < 64 ? (csr->heavy_cap + 255) / 256 : 64),
                         dim3(256), 0, stream, a, (const float*)part, o);
      GNNRAG_LAUNCH_CHECK();
    }
  }
  if (!gather) return launch_tables<BWD_REASON>(a, csr, workspace, workspace_bytes, stream);
  if (!workspace || workspace_bytes < bwd_gather_ws_bytes(relorder, D, I)) return GNNRAG_E_WORKSPACE;
  float* Vc = (float*)workspace;
  float* Qc = Vc + (size_t)(relorder->n_chunks > 0 ? relorder->n_chunks : 1) * 2 * D;
  if (relorder->n_chunks > 0) {
    int rc = 0;
    switch (I) {
      case 1: rc = launch_gather<1>(a, relorder, Vc, Qc, stream); break;
      case 2: rc = launch_gather<2>(a, relorder, Vc, Qc, stream); break;
      case 3: rc = launch_gather<3>(a, relorder, Vc, Qc, stream); break;
      default: rc = launch_gather<4>(a, relorder, Vc, Qc, stream); break;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_bwd_reduce_tables_chunks, dim3(csr->R1, 2), dim3(1024), 0, stream, a, (const float*)Vc,
                     (const int32_t*)relorder->chunk_ptr);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd_reduce_ins_chunks, dim3(csr->B, I), dim3(1024), 0, stream, a, (const float*)Qc,
                     (const int32_t*)relorder->chunk_ptr);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_aggregate_fused_backward(const gnnrag_csr* csr, const gnnrag_relorder* relorder, const float* dist,
                                               const float* P, const float* g_nbr, float* g_dist, float* g_P, int32_t D,
                                               void* workspace, size_t workspace_bytes, gnnrag_stream_t stream_) {
  if (!csr || !relorder || !dist || !P || !g_nbr || !g_dist || !g_P || D <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  if (relorder->F != csr->F || relorder->rel_total != csr->rel_total) return GNNRAG_E_BADARG;
  if (!gather_ok(relorder, D, 1) || ((((uintptr_t)P | (uintptr_t)g_nbr | (uintptr_t)g_P) & 15) != 0)) return GNNRAG_E_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  BwdArgs a;
  fill_bwd(a, csr, D, 1);
  a.w[0] = csr->w_gnn[0];
  a.w[1] = csr->w_gnn[1];
  a.dist = dist;
  a.T[0] = P;
  a.T[1] = P + (size_t)csr->rel_total * D;
  a.g = g_nbr;
  a.g_dist = g_dist;
  const dim3 pgrid((csr->N + 3) / 4, csr->B);
  hipLaunchKernelGGL((k_bwd_prior<true, true>), pgrid, dim3(256), 0, stream, a);
  GNNRAG_LAUNCH_CHECK();
  const size_t need = bwd_gather_ws_bytes(relorder, D, 1);
  const size_t hbytes = (size_t)2 * csr->max_chunks * 4 * sizeof(float);
  if (!workspace || workspace_bytes < (need > hbytes ? need : hbytes)) return GNNRAG_E_WORKSPACE;
  if (csr->F > 0) {
    float* part = (float*)workspace;
    const int nb = csr->max_chunks < 4096 ? csr->max_chunks : 4096;
    hipLaunchKernelGGL((k_bwd_prior_heavy<true, true>), dim3(nb > 0 ? nb : 1, 2), dim3(256), 0, stream, a, part);
    GNNRAG_LAUNCH_CHECK();
    for (int o = 0; o < 2; ++o) {
      hipLaunchKernelGGL(k_bwd_prior_heavy_reduce, dim3((csr->heavy_cap + 255) / 256 < 64 ? (csr->heavy_cap + 255) / 256 : 64),
                         dim3(256), 0, stream, a, (const float*)part, o);
      GNNRAG_LAUNCH_CHECK();
    }
  }
  float* Vc = (float*)workspace;
  if (relorder->n_chunks > 0) {
    const int nblk = 8 * ((relorder->n_chunks + 31) / 32);
    const int cpl = (D / 4 + 63) / 64;
#define GNNRAG_GATHER_F(C)                                                                                      \
  hipLaunchKernelGGL((k_bwd_rel_gather<1, C, true>), dim3(nblk), dim3(256), 0, stream, a, (const int2*)relorder->ht, \
                     (const float*)relorder->w, (const int32_t*)relorder->row_ptr,                             \
                     (const int32_t*)relorder->chunk_ptr, relorder->n_chunks, Vc, (float*)nullptr)
    if (cpl == 1) GNNRAG_GATHER_F(1);
    else if (cpl == 2) GNNRAG_GATHER_F(2);
    else GNNRAG_GATHER_F(4);
#undef GNNRAG_GATHER_F
    GNNRAG_LAUNCH_CHECK();
  }
  if (csr->rel_total > 0) {
    hipLaunchKernelGGL(k_bwd_reduce_rows_chunks, dim3(csr->rel_total), dim3(256), 0, stream, (const float*)Vc,
                       (const int32_t*)relorder->chunk_ptr, csr->rel_total, D, g_P);
    GNNRAG_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" size_t gnnrag_backward_workspace_bytes(const gnnrag_csr* csr, const gnnrag_relorder* relorder, int32_t D,
                                                  int32_t I) {
  if (!csr || D <= 0 || I <= 0) return 0;
  const size_t lds_path = bwd_ws_bytes(csr, D);
  const size_t gather_path = gather_ok(relorder, D, I) ? bwd_gather_ws_bytes(relorder, D, I) : 0;
  const size_t heavy = align_up((size_t)2 * csr->max_chunks * 4 * sizeof(float), 256);
  size_t need = lds_path > gather_path ? lds_path : gather_path;
  return need > heavy ? need : heavy;
}

extern "C" int gnnrag_typelayer_backward(const gnnrag_csr* csr, const gnnrag_relorder* relorder, const float* g_pre,
                                         const float* w_rel_per_fact, int use_w_rel, float* g_T,
                                         int32_t D, void* workspace, size_t workspace_bytes,
                                         gnnrag_stream_t stream_) {
  if (!csr || !g_pre || !g_T || D <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (gather_ok(relorder, D, 1) && (!use_w_rel || w_rel_per_fact)) {
    if (relorder->F != csr->F || relorder->rel_total != csr->rel_total) return GNNRAG_E_BADARG;
    if (!workspace || workspace_bytes < bwd_gather_ws_bytes(relorder, D, 1)) return GNNRAG_E_WORKSPACE;
    BwdArgs a;
    fill_bwd(a, csr, D, 1);
    a.g = g_pre;
    a.g_T[0] = g_T;
    float* Vc = (float*)workspace;
    if (relorder->n_chunks > 0) {
      const int nblk = 8 * ((relorder->n_chunks + 31) / 32);
      const int cpl = (D / 4 + 63) / 64;
      const float* wf = use_w_rel ? w_rel_per_fact : nullptr;
#define GNNRAG_TGATHER(C)                                                                                       \
  hipLaunchKernelGGL((k_bwd_type_gather<C>), dim3(nblk), dim3(256), 0, stream, a, (const int2*)relorder->ht,     \
                     (const int32_t*)relorder->perm, wf, (const int32_t*)relorder->row_ptr,                     \
                     (const int32_t*)relorder->chunk_ptr, relorder->n_chunks, Vc)
      if (cpl == 1) GNNRAG_TGATHER(1);
      else if (cpl == 2) GNNRAG_TGATHER(2);
      else GNNRAG_TGATHER(4);
#undef GNNRAG_TGATHER
      GNNRAG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_bwd_reduce_tables_chunks, dim3(csr->R1, 1), dim3(1024), 0, stream, a, (const float*)Vc,
                       (const int32_t*)relorder->chunk_ptr);
    GNNRAG_LAUNCH_CHECK();
    return 0;
  }
  if (use_w_rel && (!csr->w_rel[0] || !csr->w_rel[1])) return GNNRAG_E_BADARG;
  if (bwd_lds_bytes(1, csr->rel_max) > 160 * 1024 - 1024) return GNNRAG_E_UNSUPPORTED;
  BwdArgs a;
  fill_bwd(a, csr, D, 1);
  a.w[0] = use_w_rel ? csr->w_rel[0] : nullptr;
  a.w[1] = use_w_rel ? csr->w_rel[1] : nullptr;
  a.g = g_pre;
  a.g_T[0] = g_T;
  return launch_tables<BWD_TYPE>(a, csr, workspace, workspace_bytes, stream);
}
