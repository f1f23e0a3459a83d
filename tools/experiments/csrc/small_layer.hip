// Small hidden sizes (the released checkpoints' entity_dim 50, run zero-padded as 56; anything <= 64) on small batches
// (BASELINE configs 0 and 2: one WebQSP question / a dev batch of 32): at these sizes every kernel of the fused path is a
// few microseconds of work behind 5 - 19 us of launch, staging and dependent round trips (k_gemm_f32 tables 15.8 us,
// k_walk_slice 18.7, k_gemm_wres update 12.8, softmax 6, per layer; profiles/r06k timeline), and an iteration is 13 of them.
// Two kernels replace them (round 6):
//
//   k_tables_small    relation tables of ALL layers of an iteration in one launch (they depend on the instructions, the
//                     relation projections and e2e_linear only: reasongnn.py:71-79,98-105 pushed through e2e_linear's
//                     column blocks) - one wave per 16-row tile, exact fp32 on
//                     v_mfma_f32_16x16x4_f32, the (layer, direction)'s weight blocks staged once per workgroup in LDS;
//   k_layer_small     ONE launch per layer: the previous layer's softmax (recomputed per workgroup from the 8 KB of
//                     scores of its question in exactly k_masked_softmax's order, so the distribution is bit-identical
//                     to that kernel's and no launch is spent on it), the typed-edge walk of the workgroup's nodes (a
//                     16-lane group per node, float4 per lane, table rows gathered from L2, facts without a prior
//                     skipped - a seed prior costs next to nothing, so layer 0 needs no frontier launches), the
//                     self-block update + bias + relu, and the masked score (reasongnn.py:80-84,106-111,161-168).
//
// Every sum has one fixed order (a node's facts in stored order; k ascending; fixed trees for rows longer than
// kSlBigFacts), so results are bit-reproducible and independent of the batch composition; they are NOT bit-identical to
// the large-shape kernels' (other association), only fp32-close - the same parity gates apply.
#include "gnnrag_common.h"
#include "dense_internal.h"

namespace gnnrag {

constexpr int kSlThreads = 1024;          // = k_masked_softmax's block size: its reduction order is reproduced exactly
constexpr int kSlGroups = kSlThreads / 16;
constexpr int kSlNodesPerGroup = 4;       // nodes a 16-lane group updates together (one W row read feeds 4 nodes)
constexpr int kSlBigFacts = 192;          // rows with more facts are walked by the whole workgroup
constexpr int kSlBigCap = 256;            // such rows per workgroup kept in LDS (more: the medium list takes them)
constexpr int kSlMedCap = 2048;           // rows of 17 .. kSlBigFacts records per workgroup (a workgroup owns <= N <= 2048 nodes: never full)
constexpr float kSlVeryNeg = -100000000000.0f;
#ifndef GNNRAG_SMALL_PATH_MAX_SLOTS
#define GNNRAG_SMALL_PATH_MAX_SLOTS 131072     // B * N up to which the two-kernel path is taken (64 questions of 2048 slots)
#endif

struct SmallLayerArgs {
  // structure (merged rows)
  const int32_t* rp0;
  const int32_t* rp1;
  const int2* edge_m;           // [2F] (source node, compact relation; direction 1 offset by Rg + 1)
  const int32_t* m_from;        // [2F] d * F + sorted position (per-fact weights)
  const float* w0;              // per-fact weights of the two directions or null
  const float* w1;
  const int32_t* rel_off;       // [B + 1]
  int64_t F;
  int32_t B, N, D, rel_total, parts;
  // prior: prev_score != null -> dist = softmax(prev_score) per question (and part 0 writes it to dist_out);
  //        else dist_in is read as it is
  const float* prev_score;
  const float* dist_in;
  float* dist_out;              // [B, N] or null
  const float* P;               // [2, rel_total, D] this layer's tables
  const float* h;               // [BN, D]
  const float* W;               // e2e_linear.weight [D, ldw]: columns 0..D-1 = the self block
  const float* bias;            // [D]
  const float* w_s;             // [D]
  const float* b_s;             // [1]
  const float* mask;            // [BN]
  float* h_out;                 // [BN, D]
  float* score_out;             // [BN]
  int32_t ldw;
  int32_t dbg_skip;             // (profiling only, GNNRAG_SL_SKIP: 1 update loop, 2 medium rows, 4 big rows, 8 first-batch gathers)
};

// ---- all layers' relation tables ------------------------------------------------------------------------------------
struct SmallTablesArgs {
  const float* T;               // [L][2][R1][D] relation projections
  const float* ins;             // [B, I, D]
  const float* W[8];            // per layer: e2e_linear.weight [D, ldw]
  float* P[8];                  // per layer: [2, rel_total, D]
  const int2* rel_rows;         // [rel_total] (question, relation id)
  int32_t L, D, I, R1, rel_total, ldw, chunks;
};

// workgroup = (layer, direction, chunk of row tiles); LDS: the I weight blocks of (layer, direction) as [col][k], k padded
__global__ __launch_bounds__(kSlThreads) void k_tables_small(SmallTablesArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  const int D = a.D, I = a.I, K = I * D;
  const int KP = K + 4;                                     // row stride (floats): 16-byte aligned, breaks the bank pattern
  const int ld = blockIdx.y;                                // layer * 2 + direction
  const int layer = ld >> 1, d = ld & 1;
  const float* W = a.W[layer];
  // stage: s_w[col][i * D + kk] = W[col][(1 + 2 i + d) D + kk]
  for (int x = threadIdx.x; x < 64 * (K >> 2); x += kSlThreads) {
    const int col = x / (K >> 2), k4 = x - col * (K >> 2);
    const int k = 4 * k4, i = k / D, kk = k - i * D;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < D) v = *reinterpret_cast<const f32x4*>(W + (size_t)col * a.ldw + (size_t)(1 + 2 * i + d) * D + kk);
    *reinterpret_cast<f32x4*>(s_w + (size_t)col * KP + k) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntile = (a.rel_total + 15) >> 4;
  const float* T = a.T + ((size_t)layer * 2 + d) * a.R1 * D;
  float* P = a.P[layer] + (size_t)d * a.rel_total * D;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int nkg = (K + 15) >> 4;
  for (int tile = blockIdx.x * 16 + wave; tile < ntile; tile += a.chunks * 16) {
    const int m = min(tile * 16 + fr, a.rel_total - 1);
    const int2 br = a.rel_rows[m];
    const float* trow = T + (size_t)br.y * D;
    const float* qrow = a.ins + (size_t)br.x * I * D;
    f32x4 acc[4] = {zero4, zero4, zero4, zero4};
    // all k groups' A operands first (independent loads: one round trip instead of one per group), then the MFMA stream
    // with the B fragments from LDS
    constexpr int kMaxKg = 32;                               // I * D <= 512
    for (int kg0 = 0; kg0 < nkg; kg0 += 8) {
      f32x4 av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = (kg0 + u) * 16 + fg * 4;
        av[u] = zero4;
        if (kg0 + u < nkg && k < K) {
          const int i = k / D, kk = k - i * D;
          const f32x4 t = *reinterpret_cast<const f32x4*>(trow + kk);
          const f32x4 q = *reinterpret_cast<const f32x4*>(qrow + (size_t)i * D + kk);
          av[u] = __builtin_elementwise_max(t * q, zero4);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = (kg0 + u) * 16 + fg * 4;
        if (kg0 + u < nkg) {                                 // wave-uniform
          f32x4 bv[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            bv[nt] = k < K ? *reinterpret_cast<const f32x4*>(s_w + (size_t)(nt * 16 + fr) * KP + k) : zero4;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], bv[nt][e], acc[nt], 0, 0, 0);
        }
      }
    }
    (void)kMaxKg;
    // C layout: lane (fr, fg) holds rows 4 fg + r of column nt * 16 + fr
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = nt * 16 + fr;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = tile * 16 + fg * 4 + rr;
        if (mm < a.rel_total && col < D) P[(size_t)mm * D + col] = acc[nt][rr];
      }
    }
  }
}

// ---- one layer ------------------------------------------------------------------------------------------------------
// sum over the 16 lanes of a group (xor tree inside the DPP row)
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// acc += sum over the merged records [beg, end) of  w * dist[src] * P[row]   (this lane's float4 of the row)
// 16 records are fetched at once (one per lane of the group), their priors computed, then walked in stored order with up
// to four table rows in flight; records without a prior are not fetched.
__device__ __forceinline__ void sl_walk(f32x4& acc, const SmallLayerArgs& a, const float* __restrict__ s_dist, int n0, int Rg,
                                        const float* __restrict__ Pq0, const float* __restrict__ Pq1, int beg, int end, int l16,
                                        bool colok, int stride, int first) {
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int base = beg + first * 16; base < end; base += 16 * stride) {
    const int j = base + l16;
    float p = 0.f;
    int row = 0;
    if (j < end) {
      typedef int i32x2 __attribute__((ext_vector_type(2)));
      const i32x2 e = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(a.edge_m) + j);
      const unsigned s = (unsigned)(e.x - n0);
      p = s < (unsigned)a.N ? s_dist[s] : 0.f;
      if (a.w0) {
        const int f = a.m_from[j];
        p *= f < a.F ? a.w0[f] : a.w1[f - a.F];
      }
      row = e.y;
    }
    const int cnt = min(16, end - base);
    for (int u0 = 0; u0 < cnt; u0 += 4) {
      float pj[4];
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int src = (threadIdx.x & ~15) + min(u0 + u, 15);
        pj[u] = __shfl(p, src, 64);
        const int rj = __shfl(row, src, 64);
        t[u] = zero4;
        if (u0 + u < cnt && pj[u] != 0.f && colok) {
          const float* prow = rj > Rg ? Pq1 + (size_t)(rj - Rg - 1) * a.D : Pq0 + (size_t)rj * a.D;
          t[u] = *reinterpret_cast<const f32x4*>(prow + 4 * l16);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u0 + u < cnt) acc += pj[u] * t[u];      // (a record with p == 0 adds 0 * 0)
    }
  }
}

__global__ __launch_bounds__(kSlThreads) void k_layer_small(SmallLayerArgs a) {
  // (no static LDS: the dynamic-LDS cap of 160 KB is raised for this kernel and counts every byte)
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  float* red = s_mem;                                       // [16] wave partials of the softmax
  float& bcast = s_mem[16];
  int& s_nbig = *reinterpret_cast<int*>(s_mem + 17);
  int& s_nmed = *reinterpret_cast<int*>(s_mem + 18);
  unsigned short* s_big = reinterpret_cast<unsigned short*>(s_mem + 32);      // [kSlBigCap] rows walked by the workgroup
  unsigned short* s_med = s_big + kSlBigCap;                // [kSlMedCap] rows walked by one group with the general loop (N <= 2048)
  const int D = a.D, N = a.N;
  const int DP = 64;                                        // LDS row width of the transposed self block
  float* s_dist = s_mem + 32 + (kSlBigCap + kSlMedCap) / 2; // [N rounded up to 4]
  float* s_wt = s_dist + ((N + 3) & ~3);                    // [D][64]: s_wt[k][c] = W[c][k]
  float* s_h = s_wt + (size_t)D * DP;                       // [kSlGroups][kSlNodesPerGroup][64] node rows of the update
  float* s_nbr = s_h + (size_t)kSlGroups * kSlNodesPerGroup * 64;      // same shape: the nodes' neighbour sums
  float* s_part = s_h + (size_t)kSlNodesPerGroup * 64;      // big rows: [kSlGroups][64] partial sums, behind group 0's s_h rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = tid & 15, grp = tid >> 4, gbase = lane & ~15;
  const int g = blockIdx.x / a.parts, part = blockIdx.x - g * a.parts;
  const int64_t n0 = (int64_t)g * N;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool colok = 4 * l16 < D;
  float* hrow = s_h + (size_t)grp * kSlNodesPerGroup * 64;
  float* nrow = s_nbr + (size_t)grp * kSlNodesPerGroup * 64;
  const int per = a.parts * kSlGroups;
  if (tid == 0) { s_nbig = 0; s_nmed = 0; }

  // ---- everything that depends on nothing is requested first: the scores, the self block, the first pass's row bounds ----
  float v[2];
  if (a.prev_score) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * 1024;
      v[k] = i < N ? a.prev_score[n0 + i] : -INFINITY;
    }
  }
  // row bounds of the group's nodes of a pass: lane 4 u + {0, 1, 2, 3} reads rp0[n], rp1[n], rp0[n + 1], rp1[n + 1] of node u
  auto load_bounds = [&](int first) -> int {
    const int u = l16 >> 2, w = l16 & 3;
    const int n = first + per * u;
    int x = 0;
    if (n < N) x = ((w & 1) ? a.rp1 : a.rp0)[n0 + n + (w >> 1)];
    return x;
  };
  const int first0 = part + a.parts * grp;
  int bounds = load_bounds(first0);
  // the self block: a thread's float4 W[c][4 k4 .. 4 k4 + 3] goes to s_wt[4 k4 + e][c] (consecutive lanes = consecutive
  // columns: conflict-free LDS writes)
  const int nw4 = D * (D >> 2);
  for (int x0 = 0; x0 < nw4; x0 += kSlThreads) {
    const int x = x0 + tid;
    if (x < nw4) {
      const int k4 = x / D, c = x - k4 * D;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(a.W + (size_t)c * a.ldw + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) s_wt[(size_t)(4 * k4 + e) * DP + c] = wv[e];
    }
  }
  for (int x = tid; x < D * (DP - D); x += kSlThreads) {     // columns D .. 63 of every row: zero
    const int k = x / (DP - D), c = D + x - k * (DP - D);
    s_wt[(size_t)k * DP + c] = 0.f;
  }

  // ---- the prior: softmax of the previous layer's scores, in k_masked_softmax<2>'s exact order (softmax_layer.hip) ----
  if (a.prev_score) {
    float m = fmaxf(fmaxf(-INFINITY, v[0]), v[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (tid == 0) {
      float mm = red[0];
      for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
      bcast = mm;
    }
    __syncthreads();
    m = bcast;
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      v[k] = expf(v[k] - m);
      sum += v[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += red[w];
      bcast = t;
    }
    __syncthreads();
    const float total = bcast;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * 1024;
      if (i < N) {
        const float dd = v[k] / total;
        s_dist[i] = dd;
        if (part == 0 && a.dist_out) a.dist_out[n0 + i] = dd;
      }
    }
  } else {
    for (int i = tid; i < N; i += kSlThreads) s_dist[i] = a.dist_in[n0 + i];
  }
  __syncthreads();

  const int Rg = a.rel_off[g + 1] - a.rel_off[g];
  const float* Pq0 = a.P + (size_t)a.rel_off[g] * D;
  const float* Pq1 = a.P + ((size_t)a.rel_total + a.rel_off[g]) * D;
  const f32x4 bias4 = colok ? *reinterpret_cast<const f32x4*>(a.bias + 4 * l16) : zero4;
  const f32x4 ws4 = colok ? *reinterpret_cast<const f32x4*>(a.w_s + 4 * l16) : zero4;
  const float b_s = a.b_s[0];

  // the update of the group's nodes first + per * u (u with bit u of okmask set; their neighbour sums are in the group's
  // rows of s_nbr, their node rows in its rows of s_h): h' = relu(h W^T + b + nbr), score
  auto update = [&](int first, int per_, unsigned okmask) {
    f32x4 acc[kSlNodesPerGroup];
#pragma unroll
    for (int u = 0; u < kSlNodesPerGroup; ++u) acc[u] = *reinterpret_cast<const f32x4*>(nrow + u * 64 + 4 * l16) + bias4;
    // (the group's own lanes wrote the rows it reads: one wave, program order - no barrier needed)
    for (int k = 0; k < ((a.dbg_skip & 1) ? 0 : D); ++k) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_wt + (size_t)k * DP + 4 * l16);
#pragma unroll
      for (int u = 0; u < kSlNodesPerGroup; ++u) acc[u] += hrow[u * 64 + k] * wv;
    }
#pragma unroll
    for (int u = 0; u < kSlNodesPerGroup; ++u) {
      if ((okmask >> u) & 1u) {                              // group-uniform
        const f32x4 o = __builtin_elementwise_max(acc[u], zero4);
        const int64_t n = n0 + first + per_ * u;
        if (colok) *reinterpret_cast<f32x4*>(a.h_out + n * D + 4 * l16) = o;
        const f32x4 pr = o * ws4;
        const float sc = group16_sum((pr[0] + pr[1]) + (pr[2] + pr[3]));
        if (l16 == 0) a.score_out[n] = (sc + b_s) + (1.f - a.mask[n]) * kSlVeryNeg;
      }
    }
  };
  auto load_h = [&](int n, int u, bool ok) {
    f32x4 hv = zero4;
    if (ok && colok) hv = *reinterpret_cast<const f32x4*>(a.h + (n0 + n) * D + 4 * l16);
    *reinterpret_cast<f32x4*>(hrow + u * 64 + 4 * l16) = hv;
  };

  // ---- this workgroup's nodes: node = part + parts * (grp + kSlGroups * j), four j per pass.  A pass: the four nodes'
  // row bounds (one load, requested a pass ahead), their first 16 records each (four loads in flight), the priors, the
  // table rows four at a time; a node with more than 16 records goes to the lists below ----
  for (int first = first0; first < N; first += per * kSlNodesPerGroup) {      // (no barrier inside: groups leave on their own)
    int beg[kSlNodesPerGroup], cnt[kSlNodesPerGroup];
#pragma unroll
    for (int u = 0; u < kSlNodesPerGroup; ++u) {
      const int r00 = __shfl(bounds, gbase + 4 * u, 64), r10 = __shfl(bounds, gbase + 4 * u + 1, 64);
      const int r01 = __shfl(bounds, gbase + 4 * u + 2, 64), r11 = __shfl(bounds, gbase + 4 * u + 3, 64);
      beg[u] = r00 + r10;
      cnt[u] = (first + per * u < N) ? (r01 + r11) - beg[u] : -1;
    }
    bounds = load_bounds(first + per * kSlNodesPerGroup);    // the next pass's
    float p[kSlNodesPerGroup];
    int row[kSlNodesPerGroup];
    unsigned okmask = 0;
#pragma unroll
    for (int u = 0; u < kSlNodesPerGroup; ++u) {
      p[u] = 0.f;
      row[u] = 0;
      const bool light = cnt[u] >= 0 && cnt[u] <= 16;
      if (light) okmask |= 1u << u;
      if (light && l16 < cnt[u]) {
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        const int j = beg[u] + l16;
        const i32x2 e = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(a.edge_m) + j);
        const unsigned sidx = (unsigned)(e.x - (int)n0);
        p[u] = sidx < (unsigned)N ? s_dist[sidx] : 0.f;
        if (a.w0) {
          const int f = a.m_from[j];
          p[u] *= f < a.F ? a.w0[f] : a.w1[f - a.F];
        }
        row[u] = e.y;
      }
      load_h(first + per * u, u, light);
      if (cnt[u] > 16 && l16 == 0) {                         // longer rows: the lists
        if (cnt[u] > kSlBigFacts) {
          const int slot = atomicAdd(&s_nbig, 1);
          if (slot < kSlBigCap) s_big[slot] = (unsigned short)(first + per * u);
          else {                                             // (list full: the medium list takes it, one group walks it)
            const int s2 = atomicAdd(&s_nmed, 1);
            if (s2 < kSlMedCap) s_med[s2] = (unsigned short)(first + per * u);
          }
        } else {
          const int s2 = atomicAdd(&s_nmed, 1);
          if (s2 < kSlMedCap) s_med[s2] = (unsigned short)(first + per * u);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kSlNodesPerGroup; ++u) {
      f32x4 acc = zero4;
      if ((okmask >> u) & 1u) {
        for (int t0 = 0; t0 < ((a.dbg_skip & 8) ? 0 : cnt[u]); t0 += 4) {             // group-uniform trip count
          float pj[4];
          f32x4 t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int src = gbase + min(t0 + e, 15);
            pj[e] = __shfl(p[u], src, 64);
            const int rj = __shfl(row[u], src, 64);
            t[e] = zero4;
            if (t0 + e < cnt[u] && pj[e] != 0.f && colok) {
              const float* prow = rj > Rg ? Pq1 + (size_t)(rj - Rg - 1) * D : Pq0 + (size_t)rj * D;
              t[e] = *reinterpret_cast<const f32x4*>(prow + 4 * l16);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t0 + e < cnt[u]) acc += pj[e] * t[e];
        }
      }
      *reinterpret_cast<f32x4*>(nrow + u * 64 + 4 * l16) = acc;
    }
    update(first, per, okmask);
  }
  __syncthreads();
  // ---- rows of 17 .. kSlBigFacts records: one group each, the general loop ----
  const int nmed = (a.dbg_skip & 2) ? 0 : min(s_nmed, kSlMedCap);
  for (int i = grp; i < nmed; i += kSlGroups) {
    const int n = s_med[i];
    const int beg = a.rp0[n0 + n] + a.rp1[n0 + n], end = a.rp0[n0 + n + 1] + a.rp1[n0 + n + 1];
    f32x4 acc = zero4;
    sl_walk(acc, a, s_dist, (int)n0, Rg, Pq0, Pq1, beg, end, l16, colok, 1, 0);
    *reinterpret_cast<f32x4*>(nrow + 4 * l16) = acc;
    load_h(n, 0, true);
#pragma unroll
    for (int u = 1; u < kSlNodesPerGroup; ++u) load_h(n, u, false);
    update(n, 0, 1u);
  }
  __syncthreads();
  // ---- rows longer than kSlBigFacts: the whole workgroup walks one (group k takes records 16 k .. 16 k + 15 of every
  // 16 * kSlGroups), the groups' partial sums are added in group order ----
  const int nbig = (a.dbg_skip & 4) ? 0 : min(s_nbig, kSlBigCap);
  for (int b = 0; b < nbig; ++b) {
    // (the list order depends on which group found a row first; a row's own sum does not)
    const int n = s_big[b];
    const int beg = a.rp0[n0 + n] + a.rp1[n0 + n], end = a.rp0[n0 + n + 1] + a.rp1[n0 + n + 1];
    f32x4 acc = zero4;
    sl_walk(acc, a, s_dist, (int)n0, Rg, Pq0, Pq1, beg, end, l16, colok, kSlGroups, grp);
    *reinterpret_cast<f32x4*>(s_part + (size_t)grp * 64 + 4 * l16) = acc;
    __syncthreads();
    if (grp == 0) {
      f32x4 sum = zero4;
      for (int k = 0; k < kSlGroups; ++k) sum += *reinterpret_cast<const f32x4*>(s_part + (size_t)k * 64 + 4 * l16);
      *reinterpret_cast<f32x4*>(nrow + 4 * l16) = sum;      // (group 0's first s_nbr row; s_part lives behind its s_h rows)
      load_h(n, 0, true);
#pragma unroll
      for (int u = 1; u < kSlNodesPerGroup; ++u) load_h(n, u, false);
      update(n, 0, 1u);
    }
    __syncthreads();
  }
}

// shapes of the small path: hidden size a multiple of 4 up to 64, a question's scores in one 1024-thread pass, merged rows,
// both directions, tables that k_tables_small can address (<= 8 layers), a batch small enough that launch latency - not
// throughput - is what the large-shape kernels spend their time on
bool small_layer_shape_ok(const gnnrag_csr* csr, int32_t L, int32_t D, int32_t I) {
  const char* env = getenv("GNNRAG_SMALL_PATH");
  if (env && env[0] == '0') return false;
  if (!csr || D % 4 || D > 64 || D < 4 || I < 1 || I * D > 512 || L < 1 || L > 8) return false;
  if (csr->N > 2048 || csr->N < 1 || !csr->edge_m || !csr->m_from || csr->rel_total <= 0 || csr->F <= 0) return false;
  return (int64_t)csr->B * csr->N <= (int64_t)GNNRAG_SMALL_PATH_MAX_SLOTS;
}

size_t small_tables_bytes(const gnnrag_csr* csr, int32_t D) {
  return align_up((size_t)2 * (csr->rel_total > 0 ? csr->rel_total : 1) * D * sizeof(float), 256);
}

int small_tables_launch(const gnnrag_csr* csr, int32_t L, const float* T_all, const float* ins, const float* const* W,
                        float* const* P, int32_t D, int32_t I, hipStream_t stream) {
  if ((((uintptr_t)T_all | (uintptr_t)ins) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  SmallTablesArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T_all; a.ins = ins;
  for (int j = 0; j < L; ++j) {
    if ((((uintptr_t)W[j] | (uintptr_t)P[j]) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
    a.W[j] = W[j];
    a.P[j] = P[j];
  }
  a.rel_rows = (const int2*)csr->rel_rows;
  a.L = L; a.D = D; a.I = I; a.R1 = csr->R1; a.rel_total = csr->rel_total; a.ldw = (2 * I + 1) * D;
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  const int ntile = (csr->rel_total + 15) / 16;
  int chunks = (ntile + 15) / 16;                           // one tile per wave ...
  const int cap = (2 * cus) / (2 * L) > 0 ? (2 * cus) / (2 * L) : 1;      // ... up to two workgroups per CU over the launch
  if (chunks > cap) chunks = cap;
  a.chunks = chunks;
  const size_t lds = (size_t)64 * (I * D + 4) * sizeof(float);
  static DeviceMask cap_raised;
  {
    const int rc = raise_lds_cap(k_tables_small, cap_raised);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_tables_small, dim3(chunks, 2 * L), dim3(kSlThreads), lds, stream, a);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

int small_layer_launch(const gnnrag_csr* csr, const float* prev_score, const float* dist_in, float* dist_out, const float* P,
                       const float* h, const float* W, const float* bias, const float* w_s, const float* b_s, const float* mask,
                       float* h_out, float* score_out, int32_t D, int32_t I, hipStream_t stream) {
  if ((((uintptr_t)P | (uintptr_t)h | (uintptr_t)h_out | (uintptr_t)bias | (uintptr_t)w_s) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  SmallLayerArgs a;
  memset(&a, 0, sizeof(a));
  a.rp0 = csr->row_ptr[0]; a.rp1 = csr->row_ptr[1];
  a.edge_m = (const int2*)csr->edge_m; a.m_from = csr->m_from;
  a.w0 = csr->w_gnn[0]; a.w1 = csr->w_gnn[1];
  if (!a.w0 || !a.w1) a.w0 = a.w1 = nullptr;
  a.rel_off = csr->rel_off;
  a.F = csr->F; a.B = csr->B; a.N = csr->N; a.D = D; a.rel_total = csr->rel_total;
  a.prev_score = prev_score; a.dist_in = dist_in; a.dist_out = dist_out;
  a.P = P; a.h = h; a.W = W; a.bias = bias; a.w_s = w_s; a.b_s = b_s; a.mask = mask;
  a.h_out = h_out; a.score_out = score_out; a.ldw = (2 * I + 1) * D;
  if (const char* e = getenv("GNNRAG_SL_SKIP")) a.dbg_skip = atoi(e);
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  // node parts per question: about two workgroups per CU over the batch, at least 16 nodes per group and pass
  int parts = (2 * cus + csr->B - 1) / csr->B;
  const int maxp = (csr->N + kSlGroups * kSlNodesPerGroup - 1) / (kSlGroups * kSlNodesPerGroup);
  if (parts > maxp) parts = maxp;
  if (parts < 1) parts = 1;
  a.parts = parts;
  const size_t lds = ((size_t)32 + (kSlBigCap + kSlMedCap) / 2 + (size_t)((csr->N + 3) & ~3) + (size_t)D * 64 +
                      (size_t)2 * kSlGroups * kSlNodesPerGroup * 64) * sizeof(float);
  static DeviceMask cap_raised;
  {
    const int rc = raise_lds_cap(k_layer_small, cap_raised);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_layer_small, dim3(csr->B * parts), dim3(kSlThreads), lds, stream, a);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag

// ---- how it was wired in for the measurement (not in the library) ------------------------------------------------------
// dense_internal.h:   bool small_layer_shape_ok(const gnnrag_csr*, int32_t L, int32_t D, int32_t I);
//                     size_t small_tables_bytes(const gnnrag_csr*, int32_t D);
//                     int small_tables_launch(...);  int small_layer_launch(...);          (signatures above)
// build.py:           "small_layer.hip" in SOURCES
// gnnrag_stack_workspace_bytes:  + (small_layer_shape_ok(csr, L, D, I) ? L * small_tables_bytes(csr, D) : 0)
// gnnrag_reason_layer (behind rel_projections, when the path is not forced unfused / one-directional and the shape is ok):
//     small_tables_launch(csr, 1, T_fwd, ins, {W_e2e}, {P}, D, I, stream);
//     small_layer_launch(csr, nullptr, dist, nullptr, P, h, W_e2e, b_e2e, w_score, b_score, mask, h_out, score_out, D, I, stream);
//     gnnrag_masked_softmax(score_out, dist_out, B, N, stream);
// gnnrag_reason_stack (behind the up-front projections): small_tables_launch for all L layers into L table buffers behind
//     the stack workspace, then per layer j  small_layer_launch(csr, j ? score_out[j - 1] : nullptr, j ? nullptr : dist0,
//     j ? dist_out[j - 1] : nullptr, P[j], h[j - 1], ...), then gnnrag_masked_softmax of the last layer's scores.
