// Relation-typed neighbour aggregation over the destination-sorted structure.
//
//   REASON:  agg[n, 2i+d, :] = sum_{f : dst_d(f) = n}  w_f * dist[src_d(f)] * relu( T_d[rel_f, :] * ins[n/N, i, :] )
//   FUSED :  out[n, :]       = sum_d sum_{f : dst_d(f) = n}  w_f * dist[src_d(f)] * P[d, n/N, rel_f, :]
//   TYPE  :  h0[n, :]        = relu( sum_d sum_{f : dst_d(f) = n} v_f * T[rel_f, :] )
//
// REASON = ReasonGNNLayer.reason_layer / reason_layer_inv of the reference
// (gnn/modules/kg_reasoning/reasongnn.py:61-89 / :91-116), with rel_linear hoisted from the
// gathered per-fact rows [F,D] to the relation table [R1,D] (row-wise op, same arithmetic per row)
// and the two torch.sparse.mm products (prior gather :80/:106, scatter-add :84/:111) replaced by a
// CSR walk.  FUSED is the same walk after e2e_linear has been pushed through the (linear) sum:
// P[d,b,r,:] = sum_i W_e2e[:, block(i,d)] relu(T_d[r,:] * ins[b,i,:]) is a per-question relation
// table (gemm_f32.hip), so the walk emits the [D] contribution to e2e_linear directly instead of
// the [2I*D] concat operand.  TYPE = TypeLayer.forward (gnn/modules/layer_init.py:25-62).
// No [F,D] temporary ever exists and there are no floating-point atomics: every destination node
// is summed in its stored order (ascending fact id; hub rows of large vocabularies by relation, then
// fact id) by one owner, in registers.
//
// Mapping to CDNA4 (wave64):
//  * a group of LPN lanes (16/32/64, from D) owns one node; lane `sub` holds the VEC-wide column
//    chunks sub, sub+LPN, ... of the D-vector (float4 chunks when D % 4 == 0);
//  * row pointers, the first 64 (src, rel) records of BOTH directions and their priors dist[src]
//    are requested before any of them is consumed (one 8-byte coalesced load per lane, streamed
//    with the non-temporal hint so the CSR does not evict the relation tables from L2);
//  * facts whose prior is exactly 0 are skipped (they contribute exact zeros): on the first layer
//    of every iteration dist is the seed distribution and only the seeds' facts are live;
//  * live facts are consumed U at a time: U table rows (D*4 contiguous bytes each, L2 resident)
//    are requested back to back, then combined in fact order - the kernel is bound by the CU's
//    vector-memory path (64 B / clk: measured by ablation, round 3), not by arithmetic or misses;
//  * one wave writes the complete output row of its node (3200 B at D=200, I=2) -> full 128-byte
//    lines; this write stream is the kernel's compulsory HBM traffic;
//  * destination nodes with more than kHeavyDeg facts (Freebase hubs) are cut into 256-fact
//    chunks, one wave per chunk (k_heavy_partial; a run of equal relations = one table row), and
//    reduced in chunk order (k_heavy_reduce): deterministic, no atomics, scales to hubs with 10^5
//    facts.  The FUSED walk takes them as a dense hub-by-relation product instead when the hub rows
//    are in relation order (k_hub_weights / k_hub_dense / k_hub_finish below).
#include <cstdlib>
#include <type_traits>

#include "gnnrag_common.h"

#ifndef GNNRAG_REASON_SLICE
#define GNNRAG_REASON_SLICE 0    // unfused aggregation through LDS table slices when the tables fit.  Measured (C2):
                                 // dense prior 258 vs 281 us, seed prior 202 vs 138 us - its 64-byte output pieces
                                 // (4 per node and slice) lose to the gather walk's full 3200-byte rows, so it is off
#endif

#ifndef GNNRAG_SLICE_WIDE
#define GNNRAG_SLICE_WIDE 1         // LDS walk: 32-column slices for questions whose tables allow two of them per CU
#endif
#ifndef GNNRAG_SLICE_WIDE_LDS_KB
#define GNNRAG_SLICE_WIDE_LDS_KB 79   // ... as long as two wide workgroups fit one CU's 160 KB
#endif
#ifndef GNNRAG_SLICE_HALFSTEP
#define GNNRAG_SLICE_HALFSTEP 1     // LDS walk: facts 4..7 of a step are skipped when no node of the set has them (-2.5 us)
#endif
#ifndef GNNRAG_SLICE_SPLIT_TAIL
#define GNNRAG_SLICE_SPLIT_TAIL 1   // LDS walk: work items of the last, partial round of workgroup slots are cut in two
#endif

namespace gnnrag {

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<4> { typedef f32x4 type; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vload(const float* p) {
  return *reinterpret_cast<const typename VecT<VEC>::type*>(p);
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, typename VecT<VEC>::type v) {
  *reinterpret_cast<typename VecT<VEC>::type*>(p) = v;
}
template <int VEC>
__device__ __forceinline__ void vstore_nt(float* p, typename VecT<VEC>::type v) {
  __builtin_nontemporal_store(v, reinterpret_cast<typename VecT<VEC>::type*>(p));
}
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero() {
  typename VecT<VEC>::type z = {};
  return z;
}
__device__ __forceinline__ float vsplat1(float x, float) { return x; }
__device__ __forceinline__ f32x2 vsplat1(float x, f32x2) { return (f32x2){x, x}; }
__device__ __forceinline__ f32x4 vsplat1(float x, f32x4) { return (f32x4){x, x, x, x}; }
__device__ __forceinline__ float vfirst(float x) { return x; }
__device__ __forceinline__ float vfirst(f32x2 x) { return x[0]; }
__device__ __forceinline__ float vfirst(f32x4 x) { return x[0]; }
__device__ __forceinline__ float vrelu(float x) { return fmaxf(x, 0.f); }
__device__ __forceinline__ f32x2 vrelu(f32x2 x) { return __builtin_elementwise_max(x, (f32x2){0.f, 0.f}); }
__device__ __forceinline__ f32x4 vrelu(f32x4 x) {
  return __builtin_elementwise_max(x, (f32x4){0.f, 0.f, 0.f, 0.f});
}

enum { MODE_REASON = 0, MODE_TYPE = 1, MODE_FUSED = 2 };

struct WalkArgs {
  const int32_t* row_ptr[2];
  const int2* edge[2];
  const float* w[2];          // per-fact weight in sorted order or nullptr
  const float* T[2];          // REASON/TYPE: [R1,D]; FUSED: P[d] = [rel_total][D], question b's rows at rel_off[b]
  const int32_t* rel_off;     // FUSED: [B+1] first compact relation row of each question
  const float* dist;          // [BN] (REASON, FUSED)
  const float* ins;           // [B,I,D] (REASON)
  float* out;                 // REASON: [BN,2I*D]; TYPE/FUSED: [BN,D]
  const int32_t* heavy[2];
  const int32_t* chunk_off[2];
  const int32_t* n_heavy;
  const int32_t* n_chunks;
  float* partial;             // [2][max_chunks][NI*D] heavy-chunk partial sums
  int32_t max_chunks, heavy_cap, heavy_deg;
  int32_t BN, N, D, I, i0, R1, B;   // R1: table rows (FUSED: largest per-question count, sizes the LDS slices)
  int32_t bpg;                // FUSED: workgroups per question for the XCD-aware mapping (0 = off)
  // dense hub form (FUSED gather walk; see k_hub_dense): per-question hub offsets of the structure, the hub-by-relation
  // weights (null = form off), their capacity in floats, one (relation, partial weight) record per 256-fact chunk
  const int32_t* hub_q_off[2];
  const int32_t* hub_wbase[2];
  float* hub_w;
  long long hub_w_cap;
  int2* hub_bnd;
  int32_t hub_ks;             // relation ranges a question's dense product is cut into (one workgroup each)
  int32_t dir;                // k_heavy_reduce in read-modify-write modes: direction of this launch
  int32_t heavy_only;         // host side: the light rows were already walked by another kernel
  const int32_t* big_cnt;     // LDS walk: per-question count / list of nodes with > big_deg facts in a direction
  const int32_t* big_nodes;
  int32_t big_deg;
  int32_t skip_dir;           // LDS walk: 1 + direction to leave out (one-direction layers, NSM), 0 = walk both
  int32_t merged;             // LDS walk (FUSED): a node's facts of both directions are ONE run of the pair stream
                              // (gnnrag_csr::edge_m); direction 1's pairs carry table rows offset by Rg + 1
  const int2* edge_m;
  const int32_t* m_from;
};

template <int MODE, int NI> struct AccN { static constexpr int n = (MODE == MODE_REASON) ? NI : 1; };

// acc[i][m] += pj * f(t): the per-fact message in the three modes
template <int MODE, int VEC, int CPL, int NI>
__device__ __forceinline__ void fma_row(float pj, const typename VecT<VEC>::type (&t)[CPL], const bool (&cv)[CPL],
                                        const typename VecT<VEC>::type (&q)[NI][CPL],
                                        typename VecT<VEC>::type (&acc)[NI][CPL]) {
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    if (cv[m]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if constexpr (MODE == MODE_REASON) acc[i][m] += pj * vrelu(t[m] * q[i][m]);
        else acc[i][m] += pj * t[m];
      }
    }
  }
}

// One batch of <= LPN facts held one per lane as (p, r): consume the live ones in fact order.
template <int MODE, int VEC, int LPN, int CPL, int NI>
__device__ __forceinline__ void consume_batch(float p, int r, int cnt, const float* __restrict__ T, int D,
                                              const int (&col)[CPL], const bool (&cv)[CPL],
                                              const typename VecT<VEC>::type (&q)[NI][CPL],
                                              typename VecT<VEC>::type (&acc)[NI][CPL]) {
  typedef typename VecT<VEC>::type V;
  constexpr int U = (CPL == 1) ? 8 : (CPL == 2) ? 4 : 2;   // table rows in flight per wave
  if constexpr (LPN == 64) {
    // one node per wave: (p, rel) of a fact are wave-uniform -> scalar control flow
    // The kernel is bound by the CU's vector-memory path (64 B / clk: an 800-byte table row is ~13 clocks of it, hit or
    // miss - timing ablations: all rows from L2-resident lines saves 10 %, no row loads 50 %), so a slot of the last,
    // partial round must not load anything: its load is skipped by a scalar branch, not fed a duplicate row.
    unsigned long long live = __ballot(p != 0.f);
    while (live) {
      float pj[U];
      int rj[U];
      bool lv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        lv[u] = live != 0;
        if (live) {
          const int j = __builtin_ctzll(live);
          live &= live - 1;
          pj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), j));
          rj[u] = __builtin_amdgcn_readlane(r, j);
        } else {
          pj[u] = 0.f;          // padding: +0 * 0 leaves the sum unchanged
          rj[u] = 0;
        }
      }
      V t[U][CPL];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
          t[u][m] = vzero<VEC>();
          if (lv[u]) {
            if (cv[m]) t[u][m] = vload<VEC>(T + (size_t)rj[u] * D + col[m]);
          }
        }
#pragma unroll
      for (int u = 0; u < U; ++u) fma_row<MODE, VEC, CPL, NI>(pj[u], t[u], cv, q, acc);
    }
  } else {
    // several nodes per wave (D <= 128): each LPN-lane group broadcasts its own facts
    for (int j0 = 0; j0 < cnt; j0 += U) {
      float pj[U];
      int rj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u;
        const bool ok = j < cnt;
        pj[u] = __shfl(p, ok ? j : 0, LPN);
        rj[u] = __shfl(r, ok ? j : 0, LPN);
        if (!ok) pj[u] = 0.f;
      }
      V t[U][CPL];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int m = 0; m < CPL; ++m)
          t[u][m] = (cv[m] && pj[u] != 0.f) ? vload<VEC>(T + (size_t)rj[u] * D + col[m]) : vzero<VEC>();
#pragma unroll
      for (int u = 0; u < U; ++u) fma_row<MODE, VEC, CPL, NI>(pj[u], t[u], cv, q, acc);
    }
  }
}

// (p, r) of the fact at sorted position beg + off (off < len), else (0, 0)
template <int MODE>
__device__ __forceinline__ void load_fact(const int2* __restrict__ edge, const float* __restrict__ w,
                                          const float* __restrict__ dist, int beg, int off, int len, float& p,
                                          int& r) {
  p = 0.f;
  r = 0;
  if (off < len) {
    const int idx = beg + off;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 e = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(edge) + idx);
    r = e.y;
    if constexpr (MODE == MODE_TYPE) {
      p = w ? __builtin_nontemporal_load(w + idx) : 1.f;
    } else {
      p = dist[e.x];
      if (w) p *= __builtin_nontemporal_load(w + idx);
    }
  }
}

// dense hub form: on when the caller handed a weight buffer that holds both directions' hub-by-relation blocks (the
// sizes live on the device: every kernel of either form is launched and the form not in use returns at once)
// (the buffer holds the weight blocks and, behind them, hub_ks partial rows per hub: see hub_part)
__device__ __forceinline__ long long hub_w_total(const WalkArgs& a) {
  return (long long)a.hub_wbase[0][a.B] + (long long)a.hub_wbase[1][a.B];
}
__device__ __forceinline__ bool hub_dense_on(const WalkArgs& a) {
  if (a.hub_w == nullptr) return false;
  const long long hubs = (long long)min(a.n_heavy[0], a.heavy_cap) + min(a.n_heavy[1], a.heavy_cap);
  return hub_w_total(a) + hubs * a.hub_ks * a.D <= a.hub_w_cap;
}
// partial output row of hub entry e (direction d) from relation range ks
__device__ __forceinline__ float* hub_part(const WalkArgs& a, int ks, int d, int e) {
  const long long hubs = (long long)min(a.n_heavy[0], a.heavy_cap) + min(a.n_heavy[1], a.heavy_cap);
  const long long eg = (d ? min(a.n_heavy[0], a.heavy_cap) : 0) + e;
  return a.hub_w + hub_w_total(a) + ((long long)ks * hubs + eg) * a.D;
}

// Hub rows are stored in relation order (csr_plan.hip, hub_sort_scratch): a run of adjacent facts of one relation is
// one table row times the SUM of the run's priors.  The run's last lane gets the sum (segmented scan over the wave, a
// fixed tree: deterministic), the other lanes of the run 0, which consume_batch skips - one gather per run instead of
// one per fact.  Correct for any order (an unsorted row just has shorter runs).
__device__ __forceinline__ void merge_runs(float& p, int r, int lane) {
  const int rp = __shfl_up(r, 1, 64), rn = __shfl_down(r, 1, 64);
  int f = (lane == 0) || (rp != r);
  if (__ballot(!f) == 0) return;               // no two neighbours share a relation
  float x = p;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float y = __shfl_up(x, o, 64);
    const int g = __shfl_up(f, o, 64);
    if (lane >= o) {
      if (!f) x += y;
      f |= g;
    }
  }
  p = ((lane == 63) || (rn != r)) ? x : 0.f;
}

template <int LPN>
__device__ __forceinline__ int wave_max_over_groups(int v) {
#pragma unroll
  for (int o = LPN; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

template <int MODE, int VEC, int CPL, int NI>
__device__ __forceinline__ void load_q(const WalkArgs& a, int b, const int (&col)[CPL], const bool (&cv)[CPL],
                                       typename VecT<VEC>::type (&q)[NI][CPL]) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int m = 0; m < CPL; ++m) {
      if constexpr (MODE == MODE_REASON)
        q[i][m] = cv[m] ? vload<VEC>(a.ins + ((size_t)b * a.I + a.i0 + i) * a.D + col[m]) : vzero<VEC>();
      else
        q[i][m] = vzero<VEC>();
    }
}

__device__ __forceinline__ const float* table_of(const WalkArgs& a, int mode, int d, int b) {
  return (mode == MODE_FUSED) ? a.T[d] + (size_t)a.rel_off[b] * a.D : a.T[d];
}

// ---- light rows: one LPN-lane group per destination node, both directions ------------------
// A node's work is a chain of dependent memory round trips (row bounds -> fact records -> priors -> table rows -> store)
// and an average row has a dozen facts: at full occupancy the kernel is bound by that chain, not by bandwidth.  Every
// lane group therefore walks NPW nodes: the row bounds of all of them are requested together, then their first fact
// batches and priors, and only then are the table rows gathered node after node - one chain per NPW nodes.
#ifndef GNNRAG_HEAVY_GRID
#define GNNRAG_HEAVY_GRID 512     // workgroups (4 waves, a 256-fact chunk per wave and turn) of the hub-row kernel per direction
#endif
#ifndef GNNRAG_LIGHT_NPW
#define GNNRAG_LIGHT_NPW 2        // C5, dense prior, light + hub rows (us): 1: 961, 2: 943, 4: 1005, 8: 1043
#endif
template <int MODE, int VEC, int LPN, int CPL, int NI>
__global__ __launch_bounds__(256) void k_walk_light(const WalkArgs a) {
  typedef typename VecT<VEC>::type V;
  constexpr int NA = AccN<MODE, NI>::n;
  constexpr int NPW = GNNRAG_LIGHT_NPW;
  constexpr int GPB = 256 / LPN;                // lane groups of a workgroup
  int blk = blockIdx.x;
  if (a.bpg > 0) {
    // XCD-aware order (workgroup b runs on XCD b % 8): all workgroups of question g go to XCD
    // g % 8, so a question's relation table P[:, g] stays in ONE 4 MB L2 while it is walked.
    const int xcd = blk & 7, slot = blk >> 3;
    const int g = (slot / a.bpg) * 8 + xcd;
    if (g >= a.B) return;
    blk = g * a.bpg + slot % a.bpg;
  }
  const int sub = threadIdx.x & (LPN - 1);
  const int grp = threadIdx.x / LPN;
  const int D = a.D;
  int col[CPL];
  bool cv[CPL];
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    col[m] = (sub + m * LPN) * VEC;
    cv[m] = col[m] < D;
  }
  // structure first: all nodes' row bounds, then all first fact batches and priors, are in flight together before
  // anything is consumed
  int nn[NPW], beg[NPW][2], len[NPW][2];
  bool live[NPW], heavy[NPW][2];
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    int n = (blk * NPW + t) * GPB + grp;
    live[t] = n < a.BN;
    if (!live[t]) n = a.BN - 1;  // keep every lane in the shuffles
    nn[t] = n;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      beg[t][d] = a.row_ptr[d][n];
      len[t][d] = a.row_ptr[d][n + 1] - beg[t][d];
    }
  }
  // first fact batch of every node and direction without branches: all records are requested, then all priors
  float p0[NPW][2];
  int r0[NPW][2];
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 e0[NPW][2];
#pragma unroll
  for (int t = 0; t < NPW; ++t)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      heavy[t][d] = len[t][d] > a.heavy_deg;
      if (!live[t] || heavy[t][d]) len[t][d] = 0;
      const int idx = sub < len[t][d] ? beg[t][d] + sub : 0;      // record 0 exists in every structure (F >= 1 slots)
      e0[t][d] = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(a.edge[d]) + idx);
    }
#pragma unroll
  for (int t = 0; t < NPW; ++t)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const bool fv = sub < len[t][d];
      const int idx = fv ? beg[t][d] + sub : 0;
      float p;
      if constexpr (MODE == MODE_TYPE) {
        p = a.w[d] ? __builtin_nontemporal_load(a.w[d] + idx) : 1.f;
      } else {
        p = a.dist[fv ? e0[t][d].x : 0];
        if (a.w[d]) p *= __builtin_nontemporal_load(a.w[d] + idx);
      }
      p0[t][d] = fv ? p : 0.f;
      r0[t][d] = fv ? e0[t][d].y : 0;
    }
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    const int n = nn[t];
    const int b = n / a.N;
    V q[NA][CPL];
    load_q<MODE, VEC, CPL, NA>(a, b, col, cv, q);
    V acc[NA][CPL];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      if (MODE == MODE_REASON || d == 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int m = 0; m < CPL; ++m) acc[i][m] = vzero<VEC>();
      }
      const float* T = table_of(a, MODE, d, b);
      const int maxlen = wave_max_over_groups<LPN>(len[t][d]);
      for (int base = 0; base < maxlen; base += LPN) {
        float p = p0[t][d];
        int r = r0[t][d];
        if (base > 0) load_fact<MODE>(a.edge[d], a.w[d], a.dist, beg[t][d], base + sub, len[t][d], p, r);
        consume_batch<MODE, VEC, LPN, CPL, NA>(p, r, min(LPN, maxlen - base), T, D, col, cv, q, acc);
      }
      if constexpr (MODE == MODE_REASON) {
        if (live[t] && !heavy[t][d]) {
          float* orow = a.out + (size_t)n * (2 * a.I) * D;
#pragma unroll
          for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int m = 0; m < CPL; ++m)
              if (cv[m]) vstore_nt<VEC>(orow + (size_t)(2 * (a.i0 + i) + d) * D + col[m], acc[i][m]);
        }
      }
    }
    if constexpr (MODE != MODE_REASON) {
      // both directions summed.  If a direction is heavy its chunks are added by k_heavy_reduce
      // afterwards (which also applies TYPE's final ReLU); store the light part raw in that case.
      if (live[t]) {
        const bool fin = !(heavy[t][0] || heavy[t][1]);
#pragma unroll
        for (int m = 0; m < CPL; ++m)
          if (cv[m]) {
            V v = acc[0][m];
            if (MODE == MODE_TYPE && fin) v = vrelu(v);
            vstore<VEC>(a.out + (size_t)n * D + col[m], v);   // (non-temporal stores: no difference, 756 vs 757 us at C5)
          }
      }
    }
  }
}

// ---- heavy rows, pass 1: one wave per 256-fact chunk -> partial sums --------------------------
// ---- light rows, FOUR facts per step (FUSED gather walk, 128 < D <= 256, D % 4 == 0; round 4) ------------
// k_walk_light handles one fact per step: (p, rel) go to scalar registers, the 64 lanes cover the D columns.  At
// BASELINE config 5 that is ~12 vector and ~15 scalar instructions per fact of which two multiply - with the table
// loads and the stores ablated, the dense-prior launch still costs 160 us more than the seed-prior one: instruction
// issue, not memory, is what the kernel is bound by beside its gathers.  Here a wave handles four facts per step: its
// four 16-lane groups take one fact each, a lane covers the columns 64 c + 4 l (c = 0..NCH-1, l = lane % 16) of its
// group's fact with NCH 16-byte loads from  question table + (rel * 4 D + its column bytes)  - one 24-bit multiply per
// step, no scalar work per fact - and keeps NCH float4 partial sums; per node (both directions) the four groups' partial
// sums are reduced so that group g ends with column piece g (gfx950 v_permlane32_swap / v_permlane16_swap: a swap and an
// add per register) and stores it.  A fact's 800-byte row is still read as whole 256-byte pieces.
// Sum order: facts g, g + 4, g + 8, ... of a row in fact order per group, then (g0 + g2) + (g1 + g3): fixed.
// A zero-prior fact in a step with a live one multiplies its table row by 0 (k_walk_light skips it): the reference's own
// 0 x value; steps whose four priors are all zero are skipped.
// Measured beside it (config 5, dense-prior launch, us): k_walk_light 729; this kernel with a run per direction 714,
// merged runs 681, one step in flight instead of two 662 (50 registers: 8 waves per SIMD); a packed form with 5 facts
// per four loads (slot k = 64 i + lane -> fact k / 50, piece k % 50, partial sums through LDS) 657 against 655: the
// number of load instructions is not what bounds the gather, the 6.5 GB that pass through L1 are (DESIGN A.7).
#ifndef GNNRAG_LIGHT_QUAD
#define GNNRAG_LIGHT_QUAD 1       // 0: k_walk_light for every shape
#endif
#ifndef GNNRAG_QUAD_MERGED
#define GNNRAG_QUAD_MERGED 1      // both directions of a node as one run of the merged record stream
#endif
#ifndef GNNRAG_QUAD_STEPS
#define GNNRAG_QUAD_STEPS 1       // steps (4 facts each) whose table-row loads are in flight together
#endif
// MG: the node's facts of both directions are one run of the merged record stream (gnnrag_csr::edge_m: direction 1's
// relation index offset by the question's relation count + 1): one record batch and one prior gather per node
// instead of two, and no half-empty step per direction (a node has ~10 forward and ~2 light inverse facts at config 5).
template <int NCH, bool MG>
__global__ __launch_bounds__(256) void k_walk_light_q(const WalkArgs a) {
  constexpr int NPW = GNNRAG_LIGHT_NPW;
  constexpr int S = GNNRAG_QUAD_STEPS;
  constexpr int ND = MG ? 1 : 2;        // record runs per node
  int blk = blockIdx.x;
  if (a.bpg > 0) {                      // questions pinned to XCDs: see k_walk_light
    const int xcd = blk & 7, slot = blk >> 3;
    const int g = (slot / a.bpg) * 8 + xcd;
    if (g >= a.B) return;
    blk = g * a.bpg + slot % a.bpg;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane >> 4, l = lane & 15;
  const int D = a.D;
  const unsigned D4 = (unsigned)D * 4u;
  unsigned cb[NCH];                     // byte offset of this lane's piece c inside a table row (clamped: loads need no mask)
#pragma unroll
  for (int c = 0; c < NCH; ++c) cb[c] = 4u * (unsigned)min(64 * c + 4 * l, D - 4);
  // structure first, as in k_walk_light: row bounds of all NPW nodes, then their first record batches, then the priors
  int nn[NPW], beg[NPW][ND], len[NPW][ND];
  bool live[NPW];
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    int n = (blk * NPW + t) * 4 + wave;
    live[t] = n < a.BN;
    if (!live[t]) n = a.BN - 1;
    nn[t] = n;
    int rb[2], rl[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      rb[d] = a.row_ptr[d][n];
      rl[d] = a.row_ptr[d][n + 1] - rb[d];
    }
    if constexpr (MG) {
      // merged run = direction 0's facts, then direction 1's; a hub row's part is left to the hub kernels
      const int lo = rl[0] > a.heavy_deg ? rl[0] : 0;
      const int hi = rl[0] + (rl[1] > a.heavy_deg ? 0 : rl[1]);
      beg[t][0] = rb[0] + rb[1] + lo;
      len[t][0] = live[t] ? hi - lo : 0;
    } else {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        beg[t][d] = rb[d];
        len[t][d] = (!live[t] || rl[d] > a.heavy_deg) ? 0 : rl[d];
      }
    }
  }
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 e0[NPW][ND];
  float p0[NPW][ND];
#pragma unroll
  for (int t = 0; t < NPW; ++t)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const int idx = lane < len[t][d] ? beg[t][d] + lane : 0;         // record 0 exists in every structure
      e0[t][d] = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(MG ? a.edge_m : a.edge[d]) + idx);
    }
#pragma unroll
  for (int t = 0; t < NPW; ++t)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const bool fv = lane < len[t][d];
      float p = a.dist[fv ? e0[t][d].x : 0];
      if (!MG && a.w[d]) p *= __builtin_nontemporal_load(a.w[d] + (fv ? beg[t][d] + lane : 0));
      p0[t][d] = fv ? p : 0.f;
    }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned dir1 = MG ? (unsigned)((const char*)a.T[1] - (const char*)a.T[0]) : 0u;   // (checked on the host: fits)
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    const int n = nn[t];
    const int b = n / a.N;
    const int Rq = MG ? a.rel_off[b + 1] - a.rel_off[b] : 0;
    f32x4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = zero4;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const char* T = reinterpret_cast<const char*>(table_of(a, MODE_FUSED, d, b));
      const int rowlen = __builtin_amdgcn_readfirstlane(len[t][d]);
      for (int base = 0; base < rowlen; base += 64) {
        float p = p0[t][d];
        int r = e0[t][d].y;
        if (base > 0) {
          if constexpr (MG) load_fact<MODE_FUSED>(a.edge_m, nullptr, a.dist, beg[t][d], base + lane, rowlen, p, r);
          else load_fact<MODE_FUSED>(a.edge[d], a.w[d], a.dist, beg[t][d], base + lane, rowlen, p, r);
        }
        const int cnt = min(64, rowlen - base);
        // byte offset of the fact's table row from the question's direction-0 table (MG) / this direction's table
        unsigned ro = MG && r > Rq ? __umul24((unsigned)(r - Rq - 1), D4) + dir1 : __umul24((unsigned)r, D4);
        // a slot past the row's end has weight 0 and reads the row of the batch's first fact (a relation this node has)
        if (lane >= cnt) ro = __builtin_amdgcn_readfirstlane(ro);
        const int nsteps = (cnt + 3) >> 2;
        // steps of four facts that all have a zero prior (a seed prior: nearly all) are not walked
        const unsigned long long nz = __ballot(p != 0.f);
        int s = 0;
        for (; s + S <= nsteps; s += S) {              // S steps' table rows in flight together
          if (((nz >> (4 * s)) & ((1ull << (4 * S)) - 1ull)) == 0) continue;
          float pj[S];
          f32x4 tv[S][NCH];
#pragma unroll
          for (int u = 0; u < S; ++u) {
            const int src = 4 * (s + u) + grp;         // <= 63
            pj[u] = __shfl(p, src, 64);
            const unsigned rs = (unsigned)__shfl((int)ro, src, 64);
#pragma unroll
            for (int c = 0; c < NCH; ++c) tv[u][c] = *reinterpret_cast<const f32x4*>(T + (rs + cb[c]));
          }
#pragma unroll
          for (int u = 0; u < S; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] += pj[u] * tv[u][c];
        }
        for (; s < nsteps; ++s) {                      // the batch's last steps, one at a time
          if (((nz >> (4 * s)) & 15ull) == 0) continue;
          const int src = 4 * s + grp;
          const float pj = __shfl(p, src, 64);
          const unsigned rs = (unsigned)__shfl((int)ro, src, 64);
          f32x4 tv[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) tv[c] = *reinterpret_cast<const f32x4*>(T + (rs + cb[c]));
#pragma unroll
          for (int c = 0; c < NCH; ++c) acc[c] += pj * tv[c];
        }
      }
    }
    // the four groups' partial sums -> group g holds piece g
    f32x4 res;
    if constexpr (NCH == 4) {
      f32x4 k[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[h][e]), __float_as_uint(acc[h + 2][e]), false, false);
          k[h][e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);   // lanes 0-31: piece h of groups (0|1) + (2|3); 32-63: piece h + 2
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(k[0][e]), __float_as_uint(k[1][e]), false, false);
        res[e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[c][e];
          x += __shfl_xor(x, 32, 64);
          x += __shfl_xor(x, 16, 64);
          acc[c][e] = x;
        }
      res = acc[0];
#pragma unroll
      for (int c = 1; c < NCH; ++c)
        if (grp == c) res = acc[c];
    }
    const int col = 64 * grp + 4 * l;
    if (live[t] && grp < NCH && col < D) *reinterpret_cast<f32x4*>(a.out + (size_t)n * D + col) = res;
  }
}

template <int MODE, int VEC, int CPL, int NI>
__global__ __launch_bounds__(256) void k_heavy_partial(const WalkArgs a) {
  if constexpr (MODE == MODE_FUSED) {
    if (hub_dense_on(a)) return;               // the hub rows are a dense product in this call (k_hub_dense)
  }
  typedef typename VecT<VEC>::type V;
  constexpr int NA = AccN<MODE, NI>::n;
  const int d = blockIdx.y;
  const int wave = threadIdx.x >> 6, sub = threadIdx.x & 63;
  const int D = a.D;
  const int cnt = min(a.n_heavy[d], a.heavy_cap);
  const int nch = min(a.n_chunks[d], a.max_chunks);
  int col[CPL];
  bool cv[CPL];
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    col[m] = (sub + m * 64) * VEC;
    cv[m] = col[m] < D;
  }
  const int32_t* off = a.chunk_off[d];
  for (int c = blockIdx.x * 4 + wave; c < nch; c += gridDim.x * 4) {
    int lo = 0, hi = cnt;                       // largest e with off[e] <= c
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= c) lo = mid; else hi = mid;
    }
    const int n = a.heavy[d][lo];
    const int lc = c - off[lo];
    const int rbeg = a.row_ptr[d][n];
    const int beg = rbeg + lc * kHeavyDeg;
    const int len = min(kHeavyDeg, a.row_ptr[d][n + 1] - beg);
    const int b = n / a.N;
    V q[NA][CPL];
    load_q<MODE, VEC, CPL, NA>(a, b, col, cv, q);
    V acc[NA][CPL];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int m = 0; m < CPL; ++m) acc[i][m] = vzero<VEC>();
    const float* T = table_of(a, MODE, d, b);
    for (int base = 0; base < len; base += 64) {
      float p;
      int r;
      load_fact<MODE>(a.edge[d], a.w[d], a.dist, beg, base + sub, len, p, r);
      merge_runs(p, r, sub);
      consume_batch<MODE, VEC, 64, CPL, NA>(p, r, min(64, len - base), T, D, col, cv, q, acc);
    }
    float* prow = a.partial + ((size_t)d * a.max_chunks + c) * (NA * D);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int m = 0; m < CPL; ++m)
        if (cv[m]) vstore<VEC>(prow + (size_t)i * D + col[m], acc[i][m]);
  }
}

// ---- heavy rows, pass 2: one workgroup per heavy row sums its chunks in chunk order ------------
// (a thread per output column: the 343 chunks of a BASELINE-config-5 hub are one chain of 22 groups of 16 loads)
// REASON writes its own (i, d) output slots (grid.y = direction).  TYPE/FUSED add into the row the
// light kernel left (one launch per direction, a.dir, so two lists never touch a row concurrently).
template <int MODE>
__global__ __launch_bounds__(256) void k_heavy_reduce(const WalkArgs a, int na) {
  if constexpr (MODE == MODE_FUSED) {
    if (hub_dense_on(a)) return;
  }
  const int d = (MODE == MODE_REASON) ? (int)blockIdx.y : a.dir;
  const int D = a.D;
  const int cnt = min(a.n_heavy[d], a.heavy_cap);
  const int32_t* off = a.chunk_off[d];
  for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
    const int n = a.heavy[d][e];
    const int c0 = off[e], c1 = min(off[e + 1], a.max_chunks);
    bool relu = false;
    if (MODE == MODE_TYPE) {
      // the last pass touching node n applies the ReLU: direction 1 if n is heavy there, else 0
      const int l1 = a.row_ptr[1][n + 1] - a.row_ptr[1][n];
      relu = (d == 1) || !(l1 > a.heavy_deg);
    }
    for (int x = threadIdx.x; x < na * D; x += 256) {
      float s = 0.f;
      const float* pp = a.partial + ((size_t)d * a.max_chunks) * (na * D) + x;
      int c = c0;
      for (; c + 16 <= c1; c += 16) {                // 16 independent loads in flight, summed in chunk order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = pp[(size_t)(c + u) * (na * D)];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
      for (; c < c1; ++c) s += pp[(size_t)c * (na * D)];
      if (MODE == MODE_REASON) {
        const int i = x / D, cc = x - i * D;
        a.out[(size_t)n * (2 * a.I) * D + (size_t)(2 * (a.i0 + i) + d) * D + cc] = s;
      } else {
        float v = a.out[(size_t)n * D + x] + s;
        if (relu) v = fmaxf(v, 0.f);
        a.out[(size_t)n * D + x] = v;
      }
    }
  }
}


// ---- hub rows as a small dense product (FUSED gather walk) ------------------------------------------------------------
// A Freebase hub has far more facts than the question has relations (BASELINE config 5: 37 hubs per question hold 183 000
// of its 220 000 inverse facts; 6001 relations in use), and the gather walk is bound by the CU's vector-memory path: a
// table row per fact - or per run of equal relations, k_heavy_partial - is what costs.  With the hub rows in relation
// order the hub part of a question is  out[hubs, :] = Wgt[hubs, relations] . P[relations, :],  Wgt[h, r] = sum of the
// priors of hub h's facts of relation r:
//   k_hub_zero     zeroes the weight blocks (sizes on the device);
//   k_hub_weights  one wave per 256-fact chunk: segmented sums over runs of equal relations.  A run that STARTS in the
//                  chunk is stored to its own Wgt slot (rows are sorted, so nobody else writes it); the part of a run
//                  that started in an earlier chunk goes to the chunk's boundary record (relation, partial sum);
//   k_hub_dense    Wgt . P on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32): workgroup = (16 hubs, 64
//                  columns) of a question, its four waves split the relations and add their tiles in wave order; P is
//                  streamed once per 16 hubs in coalesced pieces instead of gathered row by row;
//   k_hub_finish   one workgroup per hub: row += dense result + the row's boundary records in chunk order.
// Every sum has a fixed order: deterministic.  Needs the hub rows in relation order (structures of this library).
#ifndef GNNRAG_HUB_DENSE
#define GNNRAG_HUB_DENSE 1
#endif
#ifndef GNNRAG_HUB_W_GRID
#define GNNRAG_HUB_W_GRID 2048   // workgroups (4 waves, a 256-fact chunk per wave and turn) of k_hub_weights per direction
#endif
#ifndef GNNRAG_HUB_U
#define GNNRAG_HUB_U 0           // k_hub_dense: groups of 16 relations per register stage (0: 2, and 4 for a single hub tile)
#endif
#ifndef GNNRAG_HUB_KS_MAX
#define GNNRAG_HUB_KS_MAX 8      // relation ranges per question (one k_hub_dense workgroup each) at batches of <= 32
                                 // questions; C5 aggregation, us: 4: 781, 8: 725, 16: 756, 32: 784
#endif

__global__ __launch_bounds__(256) void k_hub_zero(const WalkArgs a) {
  if (!hub_dense_on(a)) return;
  const long long n4 = ((long long)a.hub_wbase[0][a.B] + a.hub_wbase[1][a.B]) >> 2;
  f32x4* w = reinterpret_cast<f32x4*>(a.hub_w);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) w[i] = z;
}

constexpr int kHubOffLds = 4096;   // hubs of a direction whose chunk offsets k_hub_weights keeps in LDS (16 KB)
// (Measured and not kept, round 4: this work inside the light-row launch - as its leading workgroups 655 us for the
// light kernel against 579 + 51 apart, as every 8th group of 8 workgroups 829 us: DESIGN A.7.)
__global__ __launch_bounds__(256) void k_hub_weights(const WalkArgs a) {
  if (!hub_dense_on(a)) return;
  __shared__ int32_t s_off[kHubOffLds];
  const int d = blockIdx.y, bx = blockIdx.x, gx = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cnt = min(a.n_heavy[d], a.heavy_cap);
  const int nch = min(a.n_chunks[d], a.max_chunks);
  const int32_t* off = a.chunk_off[d];
  float* wdir = a.hub_w + (d ? a.hub_wbase[0][a.B] : 0);
  // a chunk finds its hub by bisection over the hubs' chunk offsets: ~log2(hubs) DEPENDENT loads in front of everything
  // else the wave does - from LDS when the offsets fit (one coalesced copy per workgroup), from L2 otherwise
  const bool staged = cnt <= kHubOffLds;
  if (staged && bx * 4 < nch) {
    for (int i = threadIdx.x; i < cnt; i += 256) s_off[i] = off[i];
  }
  __syncthreads();
  for (int c = bx * 4 + wave; c < nch; c += gx * 4) {
    int lo = 0, hi = cnt;                       // largest e with off[e] <= c
    if (staged) {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= c) lo = mid; else hi = mid;
      }
    } else {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid; else hi = mid;
      }
    }
    const int n = a.heavy[d][lo];
    const int lc = c - (staged ? s_off[lo] : off[lo]);
    const int beg = a.row_ptr[d][n] + lc * kHeavyDeg;
    const int len = min(kHeavyDeg, a.row_ptr[d][n + 1] - beg);
    const int q = n / a.N;
    const int nrel4 = ((a.rel_off[q + 1] - a.rel_off[q]) + 3) & ~3;
    float* wrow = wdir + a.hub_wbase[d][q] + (size_t)(lo - a.hub_q_off[d][q]) * nrel4;
    int open_rel = lc > 0 ? a.edge[d][beg - 1].y : -1;   // relation of the run that is open where the batch starts
    bool first_run = lc > 0;                              // still inside a run that started before this chunk?
    float carry = 0.f;                                    // the open run's sum so far inside this chunk
    bool pend = false, pend_first = false;                // a run reached the end of the last batch: not stored yet
    int bnd_rel = -1;
    float bnd_sum = 0.f;
    // the chunk's four batches of records, then their priors, are requested before the first is used
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    i32x2 ev[4];
    float pv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = k * 64 + lane;
      ev[k] = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(a.edge[d]) + (o < len ? beg + o : beg));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = k * 64 + lane;
      float pp = a.dist[ev[k].x];
      if (a.w[d]) pp *= __builtin_nontemporal_load(a.w[d] + (o < len ? beg + o : beg));
      pv[k] = o < len ? pp : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int base = k * 64;
      if (base >= len) break;
      float p = pv[k];
      int r = ev[k].y;
      const int nv = min(64, len - base);
      if (lane >= nv) r = -2;                             // padding: a run of its own, never stored
      int rp = __shfl_up(r, 1, 64);
      if (lane == 0) rp = open_rel;
      int f = (r != rp) ? 1 : 0;                          // first fact of a run
      float x = p;
      if (lane == 0 && !f) x += carry;
      const unsigned long long hb = __ballot(f && lane < nv);
      if (pend && (hb & 1ull)) {                          // the run carried over ended with the batch before
        if (pend_first) {
          bnd_rel = open_rel;
          bnd_sum = carry;
        } else if (lane == 0) {
          wrow[open_rel] = carry;
        }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {                  // segmented inclusive scan (fixed tree)
        const float y = __shfl_up(x, o, 64);
        const int g = __shfl_up(f, o, 64);
        if (lane >= o) {
          if (!f) x += y;
          f |= g;
        }
      }
      const int rn = __shfl_down(r, 1, 64);
      const bool last_valid = lane == nv - 1;
      const bool tail = lane < nv && (last_valid || rn != r);
      const bool more = base + 64 < len;                  // this wave has another batch of the chunk
      const bool in_first = first_run && (hb & ((2ull << lane) - 1ull)) == 0;
      if (tail && !(last_valid && more) && !in_first) wrow[r] = x;
      if (first_run) {
        const int t = hb ? __builtin_ctzll(hb) - 1 : nv - 1;      // last lane of the first run in this batch
        if (t >= 0 && (hb || !more)) {                            // ... and the run (or the chunk) ends here
          bnd_rel = __builtin_amdgcn_readlane(r, t);
          bnd_sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), t));
        }
      }
      pend = more;
      pend_first = first_run && hb == 0;                  // (first_run is still the value this batch started with)
      if (hb) first_run = false;
      carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), nv - 1));
      open_rel = __builtin_amdgcn_readlane(r, nv - 1);
    }
    if (lane == 0) a.hub_bnd[(size_t)d * a.max_chunks + c] = make_int2(bnd_rel, __float_as_int(bnd_sum));
  }
}

constexpr int kHubWaves = 8;   // waves of a k_hub_dense workgroup: 32 columns each (7 of them at D = 200)

// MT tiles of 16 hubs x the wave's 32 columns over the k groups [kg_beg, kg_end) of one question: every table row piece
// is loaded once for all MT tiles
template <int MT>
__device__ __forceinline__ void hub_dense_tiles(const WalkArgs& a, int ks, int d, int e0, int nh, int h0,
                                                const float* __restrict__ Wq, const float* __restrict__ Pq, int nrel,
                                                int nrel4, int kg_beg, int kg_end, float* __restrict__ part0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int D = a.D;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // Column tile nt holds the columns c0 + 2 j + nt (j = 0..15): lane (fr, fg) needs P[k][c0 + 2 fr + nt], nt = 0, 1 -
  // ONE 8-byte load per relation row, 128 contiguous bytes across the 16 lanes of a row; the workgroup's waves read a
  // row's whole 4 D bytes at the same moment.
  const int cb = wave * 32 + 2 * fr;               // this lane's two columns (D % 4 == 0: both in or both out)
  if (wave * 32 >= D) return;                      // (no barrier in this function)
  const bool cok = cb < D;
  const int cbl = cok ? cb : D - 2;                // loads are unconditional (clamped): no branch, no wait between them
  const float* wrow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) wrow[i] = Wq + (size_t)min(h0 + i * 16 + fr, nh - 1) * nrel4;
  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i][0] = acc[i][1] = zero4;
  // A lane (fr, fg): weights of hub fr for the relations k .. k + 3, k = 16 kg + 4 fg (one 16-byte load).  MFMA e of a
  // group contracts the relations {16 kg + 4 fg + e}: any split of k is fine as long as A and B agree.
  // The kernel is latency bound (the MFMAs of a range are ~15 us of its ~80): two register stages, the loads of the
  // next U relation groups are requested before the MFMAs of the current ones.  The stages are two distinct objects
  // and the loop is written out twice - a rotated stage would be a register copy of a load in flight, which waits for it.
  constexpr int U = GNNRAG_HUB_U ? (MT == 1 ? 2 * GNNRAG_HUB_U : GNNRAG_HUB_U) : (MT == 1 ? 4 : 2);
  struct Stage {
    f32x4 av[U][MT];
    f32x2 bv[U][4];
  };
  auto request = [&](Stage& st, int kg0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // past the range / the question's relations the WEIGHTS count as zero (select in `multiply`, zeroed padding); the
      // table rows are then read from the last row instead of being zeroed: 0 x finite
      const int k = (kg0 + u) * 16 + fg * 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) st.av[u][i] = *reinterpret_cast<const f32x4*>(wrow[i] + min(k, nrel4 - 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
        st.bv[u][e] = *reinterpret_cast<const f32x2*>(Pq + (size_t)min(k + e, nrel - 1) * D + cbl);
    }
  };
  auto multiply = [&](const Stage& st, int kg0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (kg0 + u) * 16 + fg * 4;
      const bool on = kg0 + u < kg_end && k < nrel4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float w = on ? st.av[u][i][e] : 0.f;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, st.bv[u][e][nt], acc[i][nt], 0, 0, 0);
        }
    }
  };
  Stage s0, s1;
  request(s0, kg_beg);
  for (int kg0 = kg_beg; kg0 < kg_end; kg0 += 2 * U) {
    request(s1, kg0 + U);                  // (past the range: clamped addresses, weights not used)
    __builtin_amdgcn_sched_barrier(0);     // requests stay ahead of the MFMAs
    multiply(s0, kg0);
    __builtin_amdgcn_sched_barrier(0);
    request(s0, kg0 + 2 * U);
    __builtin_amdgcn_sched_barrier(0);
    multiply(s1, kg0 + U);
    __builtin_amdgcn_sched_barrier(0);
  }
  // C layout: lane (fr, fg) holds hubs 4 fg + r of column slot fr, i.e. column c0 + 2 fr + nt of tile nt
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int m = h0 + i * 16 + fg * 4 + rr;
      if (m < nh && cok) {
        const f32x2 o = {acc[i][0][rr], acc[i][1][rr]};
        *reinterpret_cast<f32x2*>(part0 + (size_t)(e0 + m) * D + cb) = o;
      }
    }
}

// One workgroup per (relation range, question, direction): its waves cover the D columns, so the range's table rows are
// read once, as whole rows, for all hubs of the question (48 at a time), and the weight columns of the range once per
// wave (L1).  The hub_ks partial rows of a hub are added in range order by k_hub_finish.  (Two earlier forms are in
// workgroups per (16 hubs, 64 columns) re-read the table per hub tile, workgroups per 32-column group read
// 128-byte pieces of 800-byte rows - 141 and 94 us against the ~15 us of the MFMAs at BASELINE config 5.)
__global__ __launch_bounds__(64 * kHubWaves) void k_hub_dense(const WalkArgs a) {
  if (!hub_dense_on(a)) return;
  const int item = blockIdx.x;
  const int ks = item % a.hub_ks, q = (item / a.hub_ks) % a.B, d = item / (a.hub_ks * a.B);
  const int e0 = a.hub_q_off[d][q], nh = a.hub_q_off[d][q + 1] - e0;
  if (nh <= 0) return;
  const int nrel = a.rel_off[q + 1] - a.rel_off[q], nrel4 = (nrel + 3) & ~3;
  const float* Wq = a.hub_w + (d ? a.hub_wbase[0][a.B] : 0) + a.hub_wbase[d][q];
  const float* Pq = a.T[d] + (size_t)a.rel_off[q] * a.D;
  const int nkg = (nrel4 + 15) >> 4;
  const int per = (nkg + a.hub_ks - 1) / a.hub_ks;
  const int kg_beg = min(ks * per, nkg), kg_end = min(kg_beg + per, nkg);     // (an empty range still writes its zero rows)
  float* part0 = hub_part(a, ks, d, 0);      // partial row of this range and direction for hub entry 0 (sizes read once)
  for (int h0 = 0; h0 < nh; h0 += 48) {
    const int left = nh - h0;
    if (left > 32) hub_dense_tiles<3>(a, ks, d, e0, nh, h0, Wq, Pq, nrel, nrel4, kg_beg, kg_end, part0);
    else if (left > 16) hub_dense_tiles<2>(a, ks, d, e0, nh, h0, Wq, Pq, nrel, nrel4, kg_beg, kg_end, part0);
    else hub_dense_tiles<1>(a, ks, d, e0, nh, h0, Wq, Pq, nrel, nrel4, kg_beg, kg_end, part0);
  }
}

// one workgroup per hub; its four 256-thread groups take a quarter of the hub's chunks each and are added in group order
constexpr int kHubBndLds = 2048;   // boundary records of a hub staged in LDS (a hub of up to 524 288 facts)
__global__ __launch_bounds__(1024) void k_hub_finish(const WalkArgs a) {
  if (!hub_dense_on(a)) return;
  __shared__ float s_q[3][256];
  __shared__ int2 s_bnd[kHubBndLds];
  const int d = a.dir, D = a.D;
  const int grp = threadIdx.x >> 8, tx = threadIdx.x & 255;
  const int cnt = min(a.n_heavy[d], a.heavy_cap);
  const int32_t* off = a.chunk_off[d];
  const int2* bnd = a.hub_bnd + (size_t)d * a.max_chunks;
  for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
    const int n = a.heavy[d][e];
    const int q = n / a.N;
    const int c0 = off[e], c1 = min(off[e + 1], a.max_chunks);
    const int per = (c1 - c0 + 3) >> 2;
    const int cb = min(c0 + grp * per, c1), ce = min(cb + per, c1);
    const float* Pq = a.T[d] + (size_t)a.rel_off[q] * D;
    // the hub's boundary records: one coalesced load into LDS, a record without a relation becomes (row 0, weight 0) so
    // that the table-row loads below need no branch (a guarded load makes the compiler wait behind every one of them)
    const bool staged = c1 - c0 <= kHubBndLds;
    if (staged) {
      for (int i = threadIdx.x; i < c1 - c0; i += 1024) {
        int2 b = bnd[c0 + i];
        if (b.x < 0) b = make_int2(0, 0);
        s_bnd[i] = b;
      }
    }
    __syncthreads();
    for (int x0 = 0; x0 < D; x0 += 256) {
      const int x = x0 + tx;
      float s = 0.f, dense = 0.f;
      if (grp == 0 && x < D) {                       // the dense product's partial rows, in range order
        for (int ks = 0; ks < a.hub_ks; ++ks) dense += hub_part(a, ks, d, e)[x];
      }
      if (x < D) {
        int c = cb;
        if (staged) {
          for (; c + 16 <= ce; c += 16) {            // 16 table rows in flight, added in chunk order
            int2 b[16];
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) b[u] = s_bnd[c - c0 + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = Pq[(size_t)b[u].x * D + x];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = fmaf(__int_as_float(b[u].y), t[u], s);
          }
          for (; c < ce; ++c) {
            const int2 b = s_bnd[c - c0];
            s = fmaf(__int_as_float(b.y), Pq[(size_t)b.x * D + x], s);
          }
        } else {
          for (; c < ce; ++c) {
            const int2 b = bnd[c];
            if (b.x >= 0) s = fmaf(__int_as_float(b.y), Pq[(size_t)b.x * D + x], s);
          }
        }
      }
      if (grp > 0) s_q[grp - 1][tx] = s;
      __syncthreads();
      if (grp == 0 && x < D)
        a.out[(size_t)n * D + x] += dense + (((s + s_q[0][tx]) + s_q[1][tx]) + s_q[2][tx]);
      __syncthreads();
    }
  }
}

// ---- fused walk through LDS: relation-table column slices ------------------------------------
// Gathering a D*4-byte table row per fact from L2 (1.2 GB per layer call at C2) is what bounds the
// walk above.  When the per-question tables are small enough, they are staged in LDS instead:
// a workgroup owns (question g, 16-column slice c): it copies P[0:2, g, :, 16c:16c+16] (2*R1*64 B,
// 77 KB at R1 = 602) into LDS once, then walks ALL nodes of question g gathering 64-byte row
// slices from LDS.  The priors are precomputed once per call by k_fact_prior as (p, rel) pairs
// in sorted order, so the 13 slice workgroups of a question re-read a compact coalesced stream
// instead of redoing the dist[src] gather.  Four lanes own one node (float4 each), 16 nodes per
// wave; a lane group reads 4 consecutive (p, rel) pairs with one 32-byte coalesced access and
// shares them with width-4 shuffles.  Node sets are handed out by an LDS ticket so a set with a
// hub does not hold up its wave's other sets.  All slice workgroups of a question run on one
// XCD (workgroup b -> XCD b % 8), so their partial-line writes to out[] merge in that L2.
#ifndef GNNRAG_SLICE_MERGED
#define GNNRAG_SLICE_MERGED 1      // fused LDS walk over merged rows (both directions of a node in one loop)
#endif
#ifndef GNNRAG_SLICE_BL_GROUP
#define GNNRAG_SLICE_BL_GROUP 1
#endif
constexpr int kSliceW = 16;                 // floats per slice (4 lanes x float4)
#ifndef GNNRAG_SLICE_THREADS
#define GNNRAG_SLICE_THREADS 1024     // threads per LDS-walk workgroup (two workgroups per CU either way: the table slices fill its LDS)
#endif
#ifndef GNNRAG_SLICE_WPE
#define GNNRAG_SLICE_WPE 8            // waves per SIMD the register budget is cut for (1024 threads x 2 workgroups = 8: 64 VGPRs)
#endif
constexpr int kSliceThreads = GNNRAG_SLICE_THREADS;
constexpr int kSliceWaves = kSliceThreads / 64;

__global__ __launch_bounds__(256) void k_fact_prior(const int2* __restrict__ e0, const int2* __restrict__ e1,
                                                    const float* __restrict__ w0, const float* __restrict__ w1,
                                                    const float* __restrict__ dist, int64_t F,
                                                    int2* __restrict__ pr) {
  const int d = blockIdx.y;
  const int2* edge = d ? e1 : e0;
  const float* w = d ? w1 : w0;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= F) return;
  float p;
  int r;
  load_fact<MODE_FUSED>(edge, w, dist, 0, (int)i, (int)F, p, r);
  pr[(size_t)d * F + i] = make_int2(__float_as_int(p), r);
}

// the pairs of the MERGED stream (gnnrag_csr::edge_m: a node's facts of direction 0, then of direction 1, contiguous;
// direction 1's relation index already points behind direction 0's table slice): one coalesced pass over 2F records
__global__ __launch_bounds__(256) void k_fact_prior_merged(const int2* __restrict__ em, const int32_t* __restrict__ from,
                                                           const float* __restrict__ w0, const float* __restrict__ w1,
                                                           const float* __restrict__ dist, int64_t F,
                                                           int2* __restrict__ pr) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= 2 * F) return;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const i32x2 e = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(em) + j);
  float p = dist[e.x];
  if (w0) {
    const int f = from[j];
    p *= f < F ? w0[f] : w1[f - F];
  }
  pr[j] = make_int2(__float_as_int(p), e.y);
}

// Node classes of the LDS walk (by the larger of the two directions' fact counts):
//   light  (<= big_deg = 32) : a 4-lane group per node, 16 nodes per wave step;
//   medium (<= kSliceTeamDeg): one whole wave per node (64 facts per step, 8 steps in flight);
//   huge                     : the whole workgroup per node.
// A lane group walks its row with ONE outstanding 8-fact step, i.e. a row of n facts costs n/8
// dependent L2 round trips - fine for 32 facts, ruinous for 500.  Hence the wave-wide class.  The big
// (medium + huge) nodes of every question are listed once at plan time (csr->big_nodes), so a
// workgroup knows them up front: huge nodes first (whole workgroup), then ONE ticket stream hands
// out the medium nodes (longest work first) followed by the light sets - no barrier in between.
constexpr int kSliceTeamDeg = 4096;
constexpr int kSliceBigCap = 72;    // big nodes of one question kept in LDS; a question with more is walked
                                    // entirely by lane groups (slow, correct)

// one node's rows in both directions + its first 8 (p, rel) pairs per direction
struct SetRows {
  int beg[2], len[2];
  int2 first[2][2];
  int2 second[2];     // merged rows: the pairs 8..15 of the node's run, requested with the first ones (a set ahead)
  int n;
  bool valid, big;
};

// MG (compile time): merged rows - direction 1's members are constants, the compiler drops them
template <bool MG>
__device__ __forceinline__ void set_load_rows(SetRows& s, const WalkArgs& a, int g, int set, int nsets, int grp) {
  const int nl = set * 16 + grp;
  s.valid = set < nsets && nl < a.N;
  s.n = g * a.N + (s.valid ? nl : 0);
  s.big = false;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    s.beg[d] = 0;
    s.len[d] = 0;
    if (s.valid) {
      s.beg[d] = a.row_ptr[d][s.n];
      s.len[d] = a.skip_dir == d + 1 ? 0 : a.row_ptr[d][s.n + 1] - s.beg[d];
      s.big |= s.len[d] > a.big_deg;
    }
  }
  if (MG) {                 // both directions as one run: [rp0[n] + rp1[n], rp0[n+1] + rp1[n+1])
    s.beg[0] += s.beg[1];
    s.len[0] += s.len[1];
    s.beg[1] = 0;
    s.len[1] = 0;
  }
}

template <bool MG>
__device__ __forceinline__ void set_load_first(SetRows& s, const int2* const (&prd)[2], int sub, int zr) {
#pragma unroll
  for (int d = 0; d < (MG ? 1 : 2); ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      s.first[d][h] = (s.valid && !s.big && 4 * h + sub < s.len[d])
                          ? prd[d][s.beg[d] + 4 * h + sub]
                          : make_int2(0, zr);          // no fact: prior 0, the table's zero row
  if (MG) {
    // a merged run holds ~12 records on average: nearly every 16-node set has a node with more than 8, and the second
    // step's pairs, requested only when the first step starts, arrived after ~60 instructions of cover - a stall per set
#pragma unroll
    for (int h = 0; h < 2; ++h)
      s.second[h] = (s.valid && !s.big && 8 + 4 * h + sub < s.len[0]) ? prd[0][s.beg[0] + 8 + 4 * h + sub] : make_int2(0, zr);
#pragma unroll
    for (int h = 0; h < 2; ++h) s.first[1][h] = make_int2(0, zr);
  }
}

// value of lane k of this lane's quad (DPP quad_perm broadcast: VALU speed, no LDS crossbar)
template <int K>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, true);
}

// Per-lane accumulators of the LDS walk.  REASON: NI float4 (one per instruction) per direction, relu(t * q_i)
// applied per fact, 16-column slices.  FUSED: both directions summed; NI = number of 16-column groups of the
// slice a lane owns (NI = 2: 32-column slices - half as many workgroups re-walk the question's facts; used
// when a question's relation tables are small enough for two 32-column slices per CU).
template <int MODE, int NI> struct SliceAcc {
  static constexpr int n = NI;
  static constexpr int width = (MODE == MODE_REASON) ? kSliceW : kSliceW * NI;    // floats per table row in LDS
  // column offset of accumulator i inside the slice
  static constexpr int coff(int i) { return (MODE == MODE_REASON) ? 0 : kSliceW * i; }
  f32x4 v[n];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
};

// acc += sum over the 4 facts a lane group (one quad) holds one per lane in `pairs`.
// Branch-free form (default): the four facts' (p, row) are broadcast, the four LDS rows are requested back to back and
// multiplied after ONE wait - the per-fact `if (p != 0)` of the first form cost 4 of its 11 instructions per fact and
// serialised four LDS round trips per step.  Slots without a fact carry (p = 0, row = the table's extra ZERO row), so
// they add exactly 0; a real fact with p = 0 adds 0 * t like the reference's fact_val * fact_prior (reasongnn.py:82).
// One wave-uniform test skips a step none of whose 64 facts has a prior (seed priors: most steps).
template <int MODE, int NI>
__device__ __forceinline__ void slice_fma4(SliceAcc<MODE, NI>& acc, int2 pairs, const float* __restrict__ Td,
                                           const f32x4 (&q)[SliceAcc<MODE, NI>::n]) {
  if (__ballot(pairs.x != 0) == 0) return;               // (p >= 0: bits == 0 <=> p == 0)
  constexpr int NT = (MODE == MODE_REASON) ? 1 : NI;     // table float4s per fact
  // G facts' rows are in flight together (two 1024-thread workgroups per CU leave 64 VGPRs: the broadcasts are made
  // per group, not up front)
  auto group = [&](auto k0c) {
    constexpr int k0 = decltype(k0c)::value;
    constexpr int G = GNNRAG_SLICE_BL_GROUP;
    float pk[G];
    int rk[G];
    f32x4 t[G][NT];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      pk[k] = __int_as_float(k0 + k == 0 ? quad_bcast<0>(pairs.x) : k0 + k == 1 ? quad_bcast<1>(pairs.x)
                             : k0 + k == 2 ? quad_bcast<2>(pairs.x) : quad_bcast<3>(pairs.x));
      rk[k] = k0 + k == 0 ? quad_bcast<0>(pairs.y) : k0 + k == 1 ? quad_bcast<1>(pairs.y)
              : k0 + k == 2 ? quad_bcast<2>(pairs.y) : quad_bcast<3>(pairs.y);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        t[k][i] = *reinterpret_cast<const f32x4*>(Td + (size_t)rk[k] * SliceAcc<MODE, NI>::width + kSliceW * i);
    }
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if constexpr (MODE == MODE_REASON) acc.v[i] += pk[k] * vrelu(t[k][0] * q[i]);
        else acc.v[i] += pk[k] * t[k][i];
      }
  };
  group(std::integral_constant<int, 0>{});
  if constexpr (GNNRAG_SLICE_BL_GROUP < 4) group(std::integral_constant<int, GNNRAG_SLICE_BL_GROUP>{});
  if constexpr (GNNRAG_SLICE_BL_GROUP == 1) {
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
  }
}

// one direction of one row, walked by a whole wave: 64 facts per step (lane group k owns facts
// 4k..4k+3 of a step), steps first, first+stride, ...; up to 8 (wide slices: 4) steps requested before consuming
template <int MODE, int NI>
__device__ __forceinline__ void slice_walk_wave(SliceAcc<MODE, NI>& acc, const int2* __restrict__ prd, int beg,
                                                int len, int first, int stride, int lane,
                                                const float* __restrict__ Td,
                                                const f32x4 (&q)[SliceAcc<MODE, NI>::n], int zr) {
  const int nsteps = (len + 63) >> 6;
  constexpr int INF = (SliceAcc<MODE, NI>::n > 1 && MODE == MODE_FUSED) ? 4 : 8;   // steps in flight (register budget)
  for (int st = first; st < nsteps; st += stride * INF) {
    int2 pairs[INF];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      const int off = (st + stride * u) * 64 + lane;
      pairs[u] = (off < len) ? prd[beg + off] : make_int2(0, zr);
    }
#pragma unroll
    for (int u = 0; u < INF; ++u) slice_fma4<MODE, NI>(acc, pairs[u], Td, q);
  }
}

template <int MODE, int NI>
__device__ __forceinline__ void slice_wave_reduce(SliceAcc<MODE, NI>& acc) {
#pragma unroll
  for (int i = 0; i < SliceAcc<MODE, NI>::n; ++i)
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.v[i][e] += __shfl_xor(acc.v[i][e], o, 64);
    }
}

// where the float4 of (node n, accumulator i, direction d) goes
template <int MODE>
__device__ __forceinline__ float* slice_out(const WalkArgs& a, int n, int i, int d, int col) {
  if constexpr (MODE == MODE_REASON)
    return a.out + (size_t)n * (2 * a.I) * a.D + (size_t)(2 * (a.i0 + i) + d) * a.D + col;
  else
    return a.out + (size_t)n * a.D + col;
}

// Two 1024-thread workgroups per CU need 8 waves per SIMD, i.e. <= 64 VGPRs: ask for it where the
// accumulators allow (one float4 per lane in FUSED mode).
template <int MODE, int NI, bool MG>
__global__ __launch_bounds__(kSliceThreads, (MODE == MODE_FUSED ? GNNRAG_SLICE_WPE : 4)) void k_walk_slice(const WalkArgs a, const int2* __restrict__ pr,
                                                              int64_t F, int nslice, int nfull, int pl) {
  typedef SliceAcc<MODE, NI> Acc;
  constexpr int NA = Acc::n;
  constexpr int ND = (MODE == MODE_REASON) ? 2 : 1;     // output slots per node: per direction / summed
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  float* Ts = s_mem;                                   // [2][Rg + 1][16] (room for [2][R1 + 1][16]); row Rg is zero
  constexpr int SW = Acc::width;                       // floats per staged table row
  int* ctl = reinterpret_cast<int*>(s_mem + (size_t)2 * (a.R1 + 1) * SW);   // [0] the ticket
  int* blist = ctl + 16;                               // [kSliceBigCap][5]: node, beg0, len0, beg1, len1
  float* red = reinterpret_cast<float*>(blist + 5 * kSliceBigCap);   // [16 waves][NA][16 floats]
  // XCD-aware order: the nslice workgroups of question g all land on XCD g % 8.  An XCD's work items
  // (question, slice) are dispatched in order; the first nfull fill whole rounds of its workgroup slots, the
  // rest (the last, partial round) are cut in two halves of the question's nodes so that the tail of the
  // launch is made of half-length workgroups.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int item = slot, part = 0, nparts = 1;
  if (pl > 0) {               // few questions: EVERY item is cut into 2^pl parts of the question's nodes (B = 1: 13
    item = slot >> pl;        // slices x 4 parts fill the XCD's 64 workgroup slots instead of 26 of them)
    part = slot & ((1 << pl) - 1);
    nparts = 1 << pl;
  } else if (slot >= nfull) {
    const int hslot = slot - nfull;
    item = nfull + (hslot >> 1);
    part = hslot & 1;
    nparts = 2;
  }
  const int g = (item / nslice) * 8 + xcd;
  if (g >= a.B) return;
  const int c = item % nslice;
  const int col0 = c * SW;
  const int D = a.D, N = a.N;
  // rows of this question's tables: FUSED tables hold only the relations the question uses
  int Rg = a.R1, roff = 0;
  if constexpr (MODE == MODE_FUSED) {
    roff = a.rel_off[g];
    Rg = a.rel_off[g + 1] - roff;
  }
  const int tid = threadIdx.x;
  if (tid < 16) ctl[tid] = 0;
  // the question's big nodes (listed at plan time) with their row bounds -> LDS
  const int nbig = a.big_cnt[g];
  const int nlist = nbig <= kSliceBigCap ? nbig : 0;   // 0: lane groups walk everything
  if (tid < nlist) {
    const int n = a.big_nodes[(size_t)g * N + tid];
    int* e = blist + 5 * tid;
    e[0] = n;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int beg = a.row_ptr[d][n];
      e[1 + 2 * d] = beg;
      e[2 + 2 * d] = a.skip_dir == d + 1 ? 0 : a.row_ptr[d][n + 1] - beg;
    }
    if (a.merged) {
      e[1] += e[3];
      e[2] += e[4];
      e[3] = 0;
      e[4] = 0;
    }
  }
  // stage the two table slices (float4 granules; rows are D*4 bytes apart).  FUSED: the question's own
  // tables P[d, g]; REASON: the shared tables T_d
  constexpr int GR = SW / 4;                           // float4 granules per staged row
  // five granules per thread requested before the first is written (one chunk covers 602 relations x 16 columns: the
  // loop used to be load -> LDS write -> load ..., five dependent round trips to L2 / HBM in front of the walk)
  {
    constexpr int UN = 5;
    const int total = 2 * Rg * GR;
    for (int base = 0; base < total; base += kSliceThreads * UN) {
      f32x4 v[UN];
      int dst[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * kSliceThreads + tid;
        const int idc = idx < total ? idx : total - 1;
        const int d = idc >= Rg * GR;
        const int rem = idc - d * (Rg * GR);
        const int r = rem / GR, k = rem % GR;
        const bool live = idx < total && a.skip_dir != d + 1;      // (a direction that is not walked is never read)
        dst[u] = live ? (int)(((size_t)d * (Rg + 1) + r) * SW + 4 * k) : -1;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* tab = a.T[d] + (size_t)roff * D;
        if (live && col0 + 4 * k < D) v[u] = *reinterpret_cast<const f32x4*>(tab + (size_t)r * D + col0 + 4 * k);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (dst[u] >= 0) *reinterpret_cast<f32x4*>(Ts + dst[u]) = v[u];
    }
  }
  if (tid < 2 * GR) {                                  // the zero row of each direction (slots without a fact point at it)
    const int d = tid / GR, k = tid % GR;
    *reinterpret_cast<f32x4*>(Ts + ((size_t)d * (Rg + 1) + Rg) * SW + 4 * k) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 2, sub = lane & 3;
  const int nsets = (N + 15) / 16;
  bool col_ok[NA];                                     // accumulator i covers columns col0 + coff(i) + 4*sub .. +3
#pragma unroll
  for (int i = 0; i < NA; ++i) col_ok[i] = col0 + Acc::coff(i) + 4 * sub < D;
  const int2* const prd[2] = {pr, pr + F};
  const float* Td[2] = {Ts + 4 * sub, Ts + (size_t)(Rg + 1) * SW + 4 * sub};
  f32x4 q[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    q[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (MODE == MODE_REASON && col_ok[i])
      q[i] = *reinterpret_cast<const f32x4*>(a.ins + ((size_t)g * a.I + a.i0 + i) * D + col0 + 4 * sub);
  }

  // ---- huge nodes first: the whole workgroup per node; wave w takes steps w, w+16, ...; wave sums
  // are combined by a fixed xor tree inside the wave and in wave order through LDS
  {
    for (int h = 0; h < nlist; ++h) {
      const int* e = blist + 5 * h;
      if (e[2] <= kSliceTeamDeg && e[4] <= kSliceTeamDeg) continue;     // workgroup-uniform
      if ((h & (nparts - 1)) != part) continue;                         // the other half's node
      Acc acc;
      acc.zero();
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        slice_walk_wave<MODE, NI>(acc, prd[d], e[1 + 2 * d], e[2 + 2 * d], wave, kSliceWaves, lane, Td[d], q, Rg);
        if (ND == 2 || d == 1) {
          slice_wave_reduce<MODE, NI>(acc);
          if (grp == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(red + (wave * NA + i) * 16 + 4 * sub) = acc.v[i];
          }
          __syncthreads();
          if (tid < 4 * NA) {
            const int i = tid >> 2, sb = tid & 3;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < kSliceWaves; ++w) t += *reinterpret_cast<const f32x4*>(red + (w * NA + i) * 16 + 4 * sb);
            const int cc = col0 + Acc::coff(i) + 4 * sb;
            if (cc < D) *reinterpret_cast<f32x4*>(slice_out<MODE>(a, e[0], i, d, cc)) = t;
          }
          __syncthreads();
          acc.zero();
        }
      }
    }
  }

  // ---- one ticket stream: tickets [0, nlist) are the big nodes (a whole wave walks a medium one;
  // huge ones are done), tickets >= nlist are the light sets (4 lanes per node, 16 nodes per set).
  // For sets the dependent chain ticket -> row pointers -> first pairs -> table slices is software
  // pipelined three deep: while set i is walked, the first pairs of set i+1 and the row pointers of
  // set i+2 are in flight.  A row is walked 8 facts per step (two 32-byte coalesced accesses per
  // group), the next step requested before the current one is consumed.
  auto next_set = [&]() __attribute__((always_inline)) {
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&ctl[0], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nlist) return (t - nlist) * nparts + part;               // this half's sets: part, part + nparts, ...
      const int* e = blist + 5 * t;
      if (e[2] > kSliceTeamDeg || e[4] > kSliceTeamDeg) continue;       // huge: already walked
      if ((t & (nparts - 1)) != part) continue;                         // the other half's node
      Acc acc;
      acc.zero();
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        slice_walk_wave<MODE, NI>(acc, prd[d], e[1 + 2 * d], e[2 + 2 * d], 0, 1, lane, Td[d], q, Rg);
        if (ND == 2 || d == 1) {
          slice_wave_reduce<MODE, NI>(acc);
          if (grp == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
              if (col_ok[i])
                *reinterpret_cast<f32x4*>(slice_out<MODE>(a, e[0], i, d, col0 + Acc::coff(i) + 4 * sub)) = acc.v[i];
          }
          acc.zero();
        }
      }
    }
  };
  // the walk of one light set
  auto walk_set = [&](SetRows& c) __attribute__((always_inline)) {
    if (c.valid && c.big && nlist == 0) {     // no list for this question: the owner group walks it
      c.big = false;
      set_load_first<MG>(c, prd, sub, Rg);
    }
    if (c.valid && !c.big) {
      Acc acc;
      acc.zero();
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        if (MG && d == 1) continue;             // merged rows: one run per node
        const int beg = c.beg[d], len = c.len[d];
        int2 c0 = c.first[d][0], c1 = c.first[d][1];
        int j = 0;
        if (MG) {
          // first step peeled: its successor's pairs are already here (set_load_first requested them a set ago)
          if (len > 0) {
            slice_fma4<MODE, NI>(acc, c0, Td[d], q);
            if (!GNNRAG_SLICE_HALFSTEP || __ballot(4 < len)) slice_fma4<MODE, NI>(acc, c1, Td[d], q);
          }
          c0 = c.second[0];
          c1 = c.second[1];
          j = 8;
        }
        for (; j < len; j += 8) {
          int2 n0 = make_int2(0, Rg), n1 = make_int2(0, Rg);
          if (j + 8 + sub < len) n0 = prd[d][beg + j + 8 + sub];
          if (j + 12 + sub < len) n1 = prd[d][beg + j + 12 + sub];
          slice_fma4<MODE, NI>(acc, c0, Td[d], q);
          // second half of the step only if some node of the set still has facts there (most rows of the
          // inverse direction hold one or two facts)
          if (!GNNRAG_SLICE_HALFSTEP || __ballot(j + 4 < len)) slice_fma4<MODE, NI>(acc, c1, Td[d], q);
          c0 = n0;
          c1 = n1;
        }
        if (ND == 2 || d == 1 || MG) {
#pragma unroll
          for (int i = 0; i < NA; ++i)
            if (col_ok[i])
              *reinterpret_cast<f32x4*>(slice_out<MODE>(a, c.n, i, d, col0 + Acc::coff(i) + 4 * sub)) = acc.v[i];
          acc.zero();
        }
      }
    }
  };
  SetRows s0, s1, s2;
  int t0 = next_set();
  set_load_rows<MG>(s0, a, g, t0, nsets, grp);
  int t1 = t0 < nsets ? next_set() : nsets;
  set_load_rows<MG>(s1, a, g, t1, nsets, grp);
  set_load_first<MG>(s0, prd, sub, Rg);
  while (t0 < nsets) {
    const int t2 = t1 < nsets ? next_set() : nsets;
    set_load_rows<MG>(s2, a, g, t2, nsets, grp);
    set_load_first<MG>(s1, prd, sub, Rg);
    walk_set(s0);
    s0 = s1; t0 = t1;
    s1 = s2; t1 = t2;
  }
}


// GNNRAG_HUB_DENSE=0 in the environment keeps the chunked hub kernels (A/B, tests of both forms)
static bool hub_dense_enabled() {
  static const bool on = [] {
    const char* e = getenv("GNNRAG_HUB_DENSE");
    return GNNRAG_HUB_DENSE && !(e && e[0] == '0');
  }();
  return on;
}

// ---- dispatch -------------------------------------------------------------------------------
struct Shape { int vec, lpn, cpl; };

static bool pick_shape(int D, Shape* s) {
  const int vec = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
  const int chunks = D / vec;
  int lpn, cpl;
  if (chunks <= 16) { lpn = 16; cpl = 1; }
  else if (chunks <= 32) { lpn = 32; cpl = 1; }
  else if (chunks <= 64) { lpn = 64; cpl = 1; }
  else if (chunks <= 128) { lpn = 64; cpl = 2; }
  else if (chunks <= 256) { lpn = 64; cpl = 4; }
  else return false;
  s->vec = vec; s->lpn = lpn; s->cpl = cpl;
  return true;
}

template <int MODE, int VEC, int LPN, int CPL, int NI>
static int launch_one(WalkArgs a, hipStream_t stream) {
  const int nodes_per_block = (256 / LPN) * GNNRAG_LIGHT_NPW;
  int nblk = (a.BN + nodes_per_block - 1) / nodes_per_block;
  a.bpg = 0;
  if (MODE == MODE_FUSED && a.N % nodes_per_block == 0) {
    a.bpg = a.N / nodes_per_block;
    nblk = 8 * ((a.B + 7) / 8) * a.bpg;
  }
  if (!a.heavy_only) {
    bool quad = false;
    if constexpr (MODE == MODE_FUSED && VEC == 4 && LPN == 64 && CPL == 1) {
      // four facts per step (k_walk_light_q); rel * 4 D must fit the 24-bit multiply and 32-bit offsets
      quad = GNNRAG_LIGHT_QUAD && a.D > 128 && a.R1 < (1 << 20);
      if (quad) {
        // both directions as one merged run: no per-fact weights, both tables within 32-bit byte offsets of the first
        const long long span = ((const char*)a.T[1] - (const char*)a.T[0]) + (long long)(a.R1 + 1) * a.D * 4;
        const bool mg = GNNRAG_QUAD_MERGED && a.merged && !a.w[0] && !a.w[1] && a.T[1] > a.T[0] && span < (1ll << 32);
        if (a.D > 192) {
          if (mg) hipLaunchKernelGGL((k_walk_light_q<4, true>), dim3(nblk), dim3(256), 0, stream, a);
          else hipLaunchKernelGGL((k_walk_light_q<4, false>), dim3(nblk), dim3(256), 0, stream, a);
        } else {
          if (mg) hipLaunchKernelGGL((k_walk_light_q<3, true>), dim3(nblk), dim3(256), 0, stream, a);
          else hipLaunchKernelGGL((k_walk_light_q<3, false>), dim3(nblk), dim3(256), 0, stream, a);
        }
      }
    }
    if (!quad) hipLaunchKernelGGL((k_walk_light<MODE, VEC, LPN, CPL, NI>), dim3(nblk), dim3(256), 0, stream, a);
    GNNRAG_LAUNCH_CHECK();
  }
  // heavy rows (count lives on the device: fixed grids, grid-stride loops, no host sync)
  hipLaunchKernelGGL((k_heavy_partial<MODE, VEC, CPL, NI>), dim3(GNNRAG_HEAVY_GRID, 2), dim3(256), 0, stream, a);
  GNNRAG_LAUNCH_CHECK();
  const int na = AccN<MODE, NI>::n;
  if constexpr (MODE == MODE_FUSED) {
    if (a.hub_w) {      // dense hub form (returns at once on the device when the weight blocks do not fit)
      hipLaunchKernelGGL(k_hub_zero, dim3(1024), dim3(256), 0, stream, a);
      GNNRAG_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_hub_weights, dim3(GNNRAG_HUB_W_GRID, 2), dim3(256), 0, stream, a);
      GNNRAG_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_hub_dense, dim3(a.hub_ks * a.B * 2), dim3(64 * kHubWaves), 0, stream, a);
      GNNRAG_LAUNCH_CHECK();
    }
  }
  if (MODE == MODE_REASON) {
    hipLaunchKernelGGL((k_heavy_reduce<MODE>), dim3(512, 2), dim3(256), 0, stream, a, na);
    GNNRAG_LAUNCH_CHECK();
  } else {
    for (int d = 0; d < 2; ++d) {
      a.dir = d;
      hipLaunchKernelGGL((k_heavy_reduce<MODE>), dim3(512, 1), dim3(256), 0, stream, a, na);
      GNNRAG_LAUNCH_CHECK();
      if constexpr (MODE == MODE_FUSED) {
        if (a.hub_w) {
          hipLaunchKernelGGL(k_hub_finish, dim3(512), dim3(1024), 0, stream, a);
          GNNRAG_LAUNCH_CHECK();
        }
      }
    }
  }
  return 0;
}

template <int MODE, int VEC, int LPN, int CPL>
static int launch_ni(const WalkArgs& a, int ni, hipStream_t stream) {
  if constexpr (MODE != MODE_REASON) {
    return launch_one<MODE, VEC, LPN, CPL, 1>(a, stream);
  } else {
    switch (ni) {
      case 1: return launch_one<MODE, VEC, LPN, CPL, 1>(a, stream);
      case 2: return launch_one<MODE, VEC, LPN, CPL, 2>(a, stream);
      case 3: return launch_one<MODE, VEC, LPN, CPL, 3>(a, stream);
    }
    return GNNRAG_E_UNSUPPORTED;
  }
}

template <int MODE, int VEC>
static int launch_shape(const WalkArgs& a, const Shape& s, int ni, hipStream_t stream) {
  if (s.lpn == 16) return launch_ni<MODE, VEC, 16, 1>(a, ni, stream);
  if (s.lpn == 32) return launch_ni<MODE, VEC, 32, 1>(a, ni, stream);
  if (s.cpl == 1) return launch_ni<MODE, VEC, 64, 1>(a, ni, stream);
  if (s.cpl == 2) return launch_ni<MODE, VEC, 64, 2>(a, ni, stream);
  return launch_ni<MODE, VEC, 64, 4>(a, ni, stream);
}

template <int MODE>
static int launch_walk(const WalkArgs& a, int ni, hipStream_t stream) {
  Shape s;
  if (!pick_shape(a.D, &s)) return GNNRAG_E_UNSUPPORTED;
  if (s.vec == 4) return launch_shape<MODE, 4>(a, s, ni, stream);
  if (s.vec == 2) return launch_shape<MODE, 2>(a, s, ni, stream);
  return launch_shape<MODE, 1>(a, s, ni, stream);
}

static size_t partial_bytes(const gnnrag_csr* csr, int D, int na) {
  return align_up((size_t)2 * (size_t)csr->max_chunks * (size_t)na * (size_t)D * sizeof(float), 256);
}
static size_t prior_bytes(const gnnrag_csr* csr) {
  return align_up((size_t)2 * (size_t)(csr->F > 0 ? csr->F : 1) * sizeof(int2), 256);
}
static size_t slice_lds_bytes(int R1, int na = 3, int width = kSliceW) {
  return (size_t)2 * (R1 + 1) * width * sizeof(float) + (16 + 5 * kSliceBigCap) * sizeof(int) +
         (size_t)na * 16 * 16 * sizeof(float);
}
// the LDS variant needs the two table slices of a question in one CU's LDS (160 KB); rows = table rows
// of the largest question (fused: relations it uses; unfused: the whole vocabulary)
static bool slice_walk_fits(int rows, int D) {
  return D % 4 == 0 && slice_lds_bytes(rows) <= 160 * 1024 - 1024;
}

static int fill_common(WalkArgs& a, const gnnrag_csr* csr, int D, void* ws, size_t ws_bytes, int na) {
  for (int d = 0; d < 2; ++d) {
    a.row_ptr[d] = csr->row_ptr[d];
    a.edge[d] = (const int2*)csr->edge[d];
    a.heavy[d] = csr->heavy[d];
    a.chunk_off[d] = csr->chunk_off[d];
  }
  a.n_heavy = csr->n_heavy;
  a.n_chunks = csr->n_chunks;
  a.max_chunks = csr->max_chunks;
  a.heavy_cap = csr->heavy_cap;
  a.heavy_deg = csr->heavy_deg;
  a.big_cnt = csr->big_cnt;
  a.big_nodes = csr->big_nodes;
  a.big_deg = csr->big_deg;
  a.BN = csr->B * csr->N;
  a.B = csr->B;
  a.N = csr->N;
  a.R1 = csr->R1;
  a.D = D;
  if (!ws || ws_bytes < partial_bytes(csr, D, na)) return GNNRAG_E_WORKSPACE;
  a.partial = (float*)ws;
  return 0;
}

// k_fact_prior + k_walk_slice<MODE, NI>: the LDS walk of one aggregation call (or one pass of <= 3
// instructions of it).  na = accumulators whose partial-sum scratch precedes the prior pairs.
template <int MODE, int NI>
static int launch_slice(const WalkArgs& a, const gnnrag_csr* csr, void* workspace, size_t workspace_bytes,
                        int na_ws, hipStream_t stream, bool pairs_ready = false) {
  const int D = a.D;
  if (workspace_bytes < partial_bytes(csr, D, na_ws) + prior_bytes(csr)) return GNNRAG_E_WORKSPACE;
  int2* pr = (int2*)((char*)workspace + partial_bytes(csr, D, na_ws));
  const int64_t F = csr->F;
  if (F > 0 && a.i0 == 0 && !(pairs_ready && a.merged)) {     // the (p, rel) pairs do not depend on the instruction pass
    if (a.merged)
      hipLaunchKernelGGL(k_fact_prior_merged, dim3((unsigned)((2 * F + 255) / 256)), dim3(256), 0, stream, a.edge_m,
                         a.m_from, a.w[0], a.w[1], a.dist, F, pr);
    else
      hipLaunchKernelGGL(k_fact_prior, dim3((unsigned)((F + 255) / 256), 2), dim3(256), 0, stream, a.edge[0],
                         a.edge[1], a.w[0], a.w[1], a.dist, F, pr);
    GNNRAG_LAUNCH_CHECK();
  }
  constexpr int SW = SliceAcc<MODE, NI>::width;
  const int nslice = (D + SW - 1) / SW;
  const size_t lds = slice_lds_bytes(a.R1, SliceAcc<MODE, NI>::n, SW);
  static DeviceMask cap_raised;    // per kernel instantiation, per device
  {
    const int rc = raise_lds_cap(k_walk_slice<MODE, NI, false>, cap_raised);
    if (rc) return rc;
    if constexpr (MODE == MODE_FUSED) {
      static DeviceMask cap_raised_m;
      const int rc2 = raise_lds_cap(k_walk_slice<MODE, NI, true>, cap_raised_m);
      if (rc2) return rc2;
    }
  }
  // per XCD: items = (questions of the XCD) x slices, slots = 2 workgroups on each of its CUs
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  const int slots_per_xcd = cus >= 8 ? 2 * (cus / 8) : 2;
  const int items_x = ((csr->B + 7) / 8) * nslice;
  const int rem = items_x % slots_per_xcd;
  const int nfull = (GNNRAG_SLICE_SPLIT_TAIL && rem != 0) ? items_x - rem : items_x;
  int nblk = 8 * (nfull + 2 * (items_x - nfull));
  // at most 8 questions (one per XCD): cut every (question, slice) item into as many node ranges as still fit the XCD's
  // slots in one round (each part stages its own table slice; >= 16 node sets per part)
  int pl = 0;
  if (csr->B <= 8) {
    while (pl < 3 && (items_x << (pl + 1)) <= slots_per_xcd && ((a.N / 16) >> (pl + 1)) >= 16) ++pl;
    if (pl < 2) pl = 0;       // halves are what the tail split above already gives
    if (pl) nblk = 8 * (items_x << pl);
  }
  if (MODE == MODE_FUSED && a.merged) {
    if constexpr (MODE == MODE_FUSED)
      hipLaunchKernelGGL((k_walk_slice<MODE, NI, true>), dim3(nblk), dim3(kSliceThreads), lds, stream, a, (const int2*)pr, F,
                         nslice, nfull, pl);
  } else {
    hipLaunchKernelGGL((k_walk_slice<MODE, NI, false>), dim3(nblk), dim3(kSliceThreads), lds, stream, a, (const int2*)pr, F,
                       nslice, nfull, pl);
  }
  GNNRAG_LAUNCH_CHECK();
  return 0;   // hubs were walked inside the kernel, nothing to add afterwards
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" size_t gnnrag_aggregate_workspace_bytes(const gnnrag_csr* csr, int32_t D, int32_t I) {
  if (!csr || D <= 0 || I <= 0) return 0;
  return partial_bytes(csr, D, I < 3 ? I : 3) + prior_bytes(csr);
}

extern "C" int gnnrag_aggregate(const gnnrag_csr* csr, const float* dist, const float* ins,
                                const float* T_fwd, const float* T_inv, float* agg, int32_t D, int32_t I,
                                void* workspace, size_t workspace_bytes, gnnrag_stream_t stream) {
  if (!csr || !dist || !ins || !T_fwd || !T_inv || !agg || D <= 0 || I <= 0) return GNNRAG_E_BADARG;
  WalkArgs a;
  memset(&a, 0, sizeof(a));
  int rc = fill_common(a, csr, D, workspace, workspace_bytes, I < 3 ? I : 3);
  if (rc) return rc;
  a.w[0] = csr->w_gnn[0];
  a.w[1] = csr->w_gnn[1];
  a.T[0] = T_fwd;
  a.T[1] = T_inv;
  a.dist = dist;
  a.ins = ins;
  a.out = agg;
  a.I = I;
  // up to 3 instructions share one walk (their accumulators live in registers side by side)
  const bool lds_walk = slice_walk_fits(csr->R1, D) && GNNRAG_REASON_SLICE;
  for (int i0 = 0; i0 < I; i0 += 3) {
    a.i0 = i0;
    const int ni = (I - i0) < 3 ? (I - i0) : 3;
    if (lds_walk) {
      const int na_ws = I < 3 ? I : 3;
      if (ni == 1) rc = launch_slice<MODE_REASON, 1>(a, csr, workspace, workspace_bytes, na_ws, (hipStream_t)stream);
      else if (ni == 2) rc = launch_slice<MODE_REASON, 2>(a, csr, workspace, workspace_bytes, na_ws, (hipStream_t)stream);
      else rc = launch_slice<MODE_REASON, 3>(a, csr, workspace, workspace_bytes, na_ws, (hipStream_t)stream);
    } else {
      rc = launch_walk<MODE_REASON>(a, ni, (hipStream_t)stream);
    }
    if (rc) return rc;
  }
  return 0;
}

extern "C" int gnnrag_aggregate_fused(const gnnrag_csr* csr, const float* dist, const float* P, float* out,
                                      int32_t D, void* workspace, size_t workspace_bytes,
                                      gnnrag_stream_t stream_) {
  return gnnrag::aggregate_fused_dirs(csr, dist, P, out, D, 0, workspace, workspace_bytes, (hipStream_t)stream_);
}

// skip_dir: 0 both directions, 1 + d = leave direction d out (its tables are not read; LDS walk only - the gather walk
// walks both, the caller's tables for the other direction must then be zero)
// The kernel arguments of a fused aggregation call, incl. whether the hub kernels of the dense form are part of it
// (a.hub_w); shared by the call itself and by gnnrag_aggregate_fused_hub_form, which reports what the call decides.
static int prepare_fused(WalkArgs& a, const gnnrag_csr* csr, const float* dist, const float* P, float* out, int32_t D,
                         int32_t skip_dir, void* workspace, size_t workspace_bytes, int* variant_out) {
  memset(&a, 0, sizeof(a));
  const int rc = fill_common(a, csr, D, workspace, workspace_bytes, 1);
  if (rc) return rc;
  a.w[0] = csr->w_gnn[0];
  a.w[1] = csr->w_gnn[1];
  a.T[0] = P;
  a.T[1] = P + (size_t)csr->rel_total * D;
  for (int d = 0; d < 2; ++d) a.edge[d] = (const int2*)csr->edge_l[d];   // relation = compact index in the question
  a.rel_off = csr->rel_off;
  a.R1 = csr->rel_max;
  a.dist = dist;
  a.out = out;
  a.I = 1;
  a.skip_dir = skip_dir;
  // merged rows: whenever both directions are walked and the structure carries the merged positions
  a.merged = (GNNRAG_SLICE_MERGED && skip_dir == 0 && csr->edge_m && csr->m_from) ? 1 : 0;
  a.edge_m = (const int2*)csr->edge_m;
  a.m_from = csr->m_from;
  const int variant = gnnrag_aggregate_fused_variant(csr, D);
  if (variant == GNNRAG_WALK_L2_GATHER && (D & 3) == 0 && D <= 32 * kHubWaves && hub_dense_enabled() && csr->hub_sorted &&
      csr->hub_q_off[0] && csr->hub_wbase[0]) {
    // dense hub form: boundary records and weight blocks live where the LDS walk keeps its prior pairs (unused by the
    // gather walk); whether the blocks fit is decided on the device (hub_dense_on)
    const size_t used = partial_bytes(csr, D, 1), avail = prior_bytes(csr);
    const size_t bnd_bytes = align_up((size_t)2 * (size_t)csr->max_chunks * sizeof(int2), 256);
    if (workspace_bytes >= used + avail && avail > bnd_bytes + 4096) {
      for (int d = 0; d < 2; ++d) {
        a.hub_q_off[d] = csr->hub_q_off[d];
        a.hub_wbase[d] = csr->hub_wbase[d];
      }
      a.hub_bnd = (int2*)((char*)workspace + used);
      a.hub_w = (float*)((char*)workspace + used + bnd_bytes);
      const size_t cap = (avail - bnd_bytes) / sizeof(float);
      a.hub_w_cap = (long long)(cap < ((size_t)1 << 30) ? cap : ((size_t)1 << 30));
      // about two workgroups per CU when one direction has hubs in every question (BASELINE config 5)
      const int ks = (64 * GNNRAG_HUB_KS_MAX) / (2 * csr->B);
      a.hub_ks = ks < 1 ? 1 : ks > GNNRAG_HUB_KS_MAX ? GNNRAG_HUB_KS_MAX : ks;
    }
  }
  *variant_out = variant;
  return 0;
}

int gnnrag::prior_pairs_target(const gnnrag_csr* csr, int32_t D, void* workspace, size_t workspace_bytes,
                               PriorPairsTarget* out) {
  if (!csr || !out || !workspace || D <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  memset(out, 0, sizeof(*out));
  WalkArgs a;
  int variant = 0;
  // dist / P / out are not dereferenced here; any non-null value passes prepare_fused
  const int rc = prepare_fused(a, csr, (const float*)workspace, (const float*)workspace, (float*)workspace, D, 0, workspace,
                               workspace_bytes, &variant);
  if (rc) return rc;
  if (variant == GNNRAG_WALK_L2_GATHER || !a.merged || csr->F <= 0) return 0;
  if (workspace_bytes < partial_bytes(csr, D, 1) + prior_bytes(csr)) return GNNRAG_E_WORKSPACE;
  out->edge_m = a.edge_m;
  out->m_from = a.m_from;
  out->w0 = a.w[0];
  out->w1 = a.w[1];
  out->pairs = (int2*)((char*)workspace + partial_bytes(csr, D, 1));
  out->F = csr->F;
  out->ok = true;
  return 0;
}

int gnnrag::aggregate_fused_dirs(const gnnrag_csr* csr, const float* dist, const float* P, float* out, int32_t D,
                                 int32_t skip_dir, void* workspace, size_t workspace_bytes, hipStream_t stream,
                                 bool pairs_ready) {
  if (!csr || !dist || !P || !out || D <= 0 || csr->rel_total < 0 || skip_dir < 0 || skip_dir > 2) return GNNRAG_E_BADARG;
  WalkArgs a;
  int variant = 0;
  const int rc = prepare_fused(a, csr, dist, P, out, D, skip_dir, workspace, workspace_bytes, &variant);
  if (rc) return rc;
  switch (variant) {
    case GNNRAG_WALK_L2_GATHER: return launch_walk<MODE_FUSED>(a, 1, stream);   // tables too big for LDS
    // 32-column slices when two of them still fit a CU's LDS (questions that use up to ~300 relations): half
    // as many workgroups re-walk the question's facts
    case GNNRAG_WALK_LDS_32: return launch_slice<MODE_FUSED, 2>(a, csr, workspace, workspace_bytes, 1, stream, pairs_ready);
    default: return launch_slice<MODE_FUSED, 1>(a, csr, workspace, workspace_bytes, 1, stream, pairs_ready);
  }
}

// one thread: the predicate every hub kernel of the call evaluates, on the call's own arguments
__global__ void k_hub_form_report(const WalkArgs a, int32_t* form) {
  form[0] = a.hub_w == nullptr ? GNNRAG_HUB_FORM_NONE : hub_dense_on(a) ? GNNRAG_HUB_FORM_DENSE : GNNRAG_HUB_FORM_CHUNKED;
  form[1] = min(a.n_heavy[0], a.heavy_cap);
  form[2] = min(a.n_heavy[1], a.heavy_cap);
  form[3] = a.hub_ks;
}

extern "C" int gnnrag_aggregate_fused_hub_form(const gnnrag_csr* csr, int32_t D, void* workspace, size_t workspace_bytes,
                                               int32_t* form_dev, gnnrag_stream_t stream) {
  if (!csr || !workspace || !form_dev || D <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  WalkArgs a;
  int variant = 0;
  // dist / P / out are not dereferenced by the report kernel; any non-null value passes prepare_fused
  const int rc = prepare_fused(a, csr, (const float*)workspace, (const float*)workspace, (float*)workspace, D, 0, workspace,
                               workspace_bytes, &variant);
  if (rc) return rc;
  hipLaunchKernelGGL(k_hub_form_report, dim3(1), dim3(1), 0, (hipStream_t)stream, a, form_dev);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_aggregate_fused_variant(const gnnrag_csr* csr, int32_t D) {
  if (!csr || D <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  if (!slice_walk_fits(csr->rel_max, D)) return GNNRAG_WALK_L2_GATHER;
  if (GNNRAG_SLICE_WIDE && D > kSliceW && slice_lds_bytes(csr->rel_max, 2, 2 * kSliceW) <= GNNRAG_SLICE_WIDE_LDS_KB * 1024)
    return GNNRAG_WALK_LDS_32;
  return GNNRAG_WALK_LDS_16;
}

extern "C" int gnnrag_typelayer(const gnnrag_csr* csr, const float* T, int use_w_rel, float* h0, int32_t D,
                                void* workspace, size_t workspace_bytes, gnnrag_stream_t stream) {
  if (!csr || !T || !h0 || D <= 0) return GNNRAG_E_BADARG;
  if (use_w_rel && (!csr->w_rel[0] || !csr->w_rel[1])) return GNNRAG_E_BADARG;
  WalkArgs a;
  memset(&a, 0, sizeof(a));
  const int rc = fill_common(a, csr, D, workspace, workspace_bytes, 1);
  if (rc) return rc;
  a.w[0] = use_w_rel ? csr->w_rel[0] : nullptr;
  a.w[1] = use_w_rel ? csr->w_rel[1] : nullptr;
  a.T[0] = T;
  a.T[1] = T;
  a.out = h0;
  a.I = 1;
  return launch_walk<MODE_TYPE>(a, 1, (hipStream_t)stream);
}
