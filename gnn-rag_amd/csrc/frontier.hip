// The first layer of every ReaRev iteration starts from the SEED distribution (rearev.py:208: `self.curr_dist =
// current_dist` = the one-hot / few-hot seed_dist of dataset_load.py:249-257), so
//   fact_prior = head2fact . dist            (reasongnn.py:80 / :106)
// is zero for every fact whose source is not a seed, and `fact_val * fact_prior` (:82) contributes exact zeros.  The
// fused layer then needs
//   * relation-table rows P[d, (b, r), :] only for the relations r that some seed fact of question b carries,
//   * neighbour sums nbr[n, :] only for the nodes n that receive a seed fact (the FRONTIER); all other rows are 0,
//   * the self-block update for every node as before, with `nbr` read only where it is non-zero.
// Nothing here assumes that the prior IS a seed distribution: the frontier is derived on the device from whatever
// `dist` holds (nodes with dist != 0), so any prior gives correct results; a dense prior just makes the frontier the
// whole graph (and the caller should not ask for this form then).
//
//   k_frontier_build   one workgroup per question: nodes with dist != 0 (its seeds); their facts are row s of the
//                      OTHER direction's structure (facts with src_d = s are the facts whose dst_{1-d} = s), each
//                      marks its destination row and its compact relation row; then an ordered compaction of the
//                      two flag arrays into the question's OWN list segments (rows[g * N ..], trows[rel_off[g] ..]) with
//                      per-question counts - no global counter, nothing to zero between launches.
//   k_tables_frontier  P rows of the listed compact relation rows, both directions, written IN PLACE (same layout as
//                      the full table launch): one wave per 16 listed rows x 64 columns, exact fp32 MFMA, operands
//                      straight from L2, the relu(T_d[r] * ins[b, i]) operand generated in the loader (gemm_f32.hip
//                      AMODE_GEN does the same for the full tables).
//   k_walk_frontier    one workgroup per listed node: scans the node's merged fact run, keeps the facts with p != 0 in
//                      position order, sums p * P[d, row(b, rel), :] over them - in one chain when there are few
//                      (identical to the LDS walk's light-row arithmetic), by 4 waves + a fixed-order reduction else.
#include "dense_internal.h"

namespace gnnrag {

constexpr int kFrThreads = 1024;
constexpr int kFrSeedCap = 2048;          // seeds of one question kept in LDS; more -> the whole question is flagged
constexpr int kFrAltSeeds = 32;           // up to this many seeds of a question are also listed (ascending) for the walk

struct FrontierWs {                        // offsets into the caller's frontier workspace
  size_t counts, row_flag, rows, tflag, trows, seeds, seedinfo, total;
};

static FrontierWs frontier_ws(const gnnrag_csr* csr) {
  FrontierWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t BN = (size_t)csr->B * csr->N;
  const size_t RT = (size_t)(csr->rel_total > 0 ? csr->rel_total : 1);
  w.counts = take((size_t)2 * csr->B * sizeof(int32_t));      // [B][2]: listed nodes / relation rows of each question
  w.row_flag = take(BN + 64);              // one byte per node (+ padding: the update kernels read 4 flags at once)
  w.rows = take(BN * sizeof(int32_t));
  w.tflag = take(RT + 64);
  w.trows = take(RT * sizeof(int32_t));
  w.seeds = take((size_t)csr->B * kFrAltSeeds * sizeof(int32_t));   // a question's seeds in ascending order ...
  w.seedinfo = take((size_t)2 * csr->B * sizeof(int32_t));          // ... their number (-1: more than listed) and facts
  w.total = off;
  return w;
}

// ---- frontier of one question ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFrThreads) void k_frontier_build(
    const float* __restrict__ dist, const int32_t* __restrict__ rp0, const int32_t* __restrict__ rp1,
    const int2* __restrict__ el0, const int2* __restrict__ el1, const int32_t* __restrict__ rel_off, int N,
    uint8_t* __restrict__ row_flag, int32_t* __restrict__ rows, uint8_t* __restrict__ tflag,
    int32_t* __restrict__ trows, int32_t* __restrict__ counts, int32_t* __restrict__ seeds,
    int32_t* __restrict__ seedinfo, float* __restrict__ zero_a, long long zero_na, float* __restrict__ zero_b,
    long long zero_nb) {
  __shared__ int s_seed[kFrSeedCap];
  __shared__ int s_ns, s_wsum[kFrThreads / 64];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int roff = rel_off[g], Rg = rel_off[g + 1] - roff;
  const size_t n0 = (size_t)g * N;
  // side jobs of this launch (it runs in front of the layer's other kernels): zero the score buffer the bf16x3 update
  // accumulates onto, and the zero row behind nbr that unflagged rows read
  for (long long i = (long long)blockIdx.x * kFrThreads + tid; i < zero_na; i += (long long)gridDim.x * kFrThreads) zero_a[i] = 0.f;
  for (long long i = (long long)blockIdx.x * kFrThreads + tid; i < zero_nb; i += (long long)gridDim.x * kFrThreads) zero_b[i] = 0.f;
  if (tid == 0) s_ns = 0;
  if (g == (int)gridDim.x - 1 && tid < 64) row_flag[(size_t)gridDim.x * N + tid] = 0;     // the padding behind the flags
  for (int i = tid; i < N; i += kFrThreads) row_flag[n0 + i] = 0;
  for (int i = tid; i < Rg; i += kFrThreads) tflag[roff + i] = 0;
  __syncthreads();
  for (int i = tid; i < N; i += kFrThreads) {
    if (dist[n0 + i] != 0.f) {
      const int k = atomicAdd(&s_ns, 1);
      if (k < kFrSeedCap) s_seed[k] = (int)n0 + i;
    }
  }
  __syncthreads();
  const int ns = s_ns;
  if (tid == 0) {
    // the seeds in ascending order and the number of facts that touch them, for k_walk_frontier: a listed HUB finds its
    // few facts from a seed in the seeds' short rows instead of scanning its own (the LDS list is in atomic order)
    int cnt = -1, facts = 0;
    if (ns <= kFrAltSeeds) {
      cnt = ns;
      for (int i = 1; i < ns; ++i) {
        const int v = s_seed[i];
        int j = i - 1;
        for (; j >= 0 && s_seed[j] > v; --j) s_seed[j + 1] = s_seed[j];
        s_seed[j + 1] = v;
      }
      for (int i = 0; i < ns; ++i) {
        const int sn = s_seed[i];
        seeds[g * kFrAltSeeds + i] = sn;
        facts += (rp0[sn + 1] - rp0[sn]) + (rp1[sn + 1] - rp1[sn]);
      }
    }
    seedinfo[2 * g] = cnt;
    seedinfo[2 * g + 1] = facts;
  }
  __syncthreads();
  if (ns > kFrSeedCap) {                   // not a seed distribution: everything is frontier
    for (int i = tid; i < N; i += kFrThreads) row_flag[n0 + i] = 1;
    for (int i = tid; i < Rg; i += kFrThreads) tflag[roff + i] = 1;
  } else {
    for (int k = 0; k < ns; ++k) {
      const int s = s_seed[k];
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {     // row s of structure dd = the facts of direction 1 - dd that START at s
        const int32_t* rp = dd ? rp1 : rp0;
        const int2* el = dd ? el1 : el0;
        const int beg = rp[s], end = rp[s + 1];
        for (int j = beg + tid; j < end; j += kFrThreads) {
          const int2 e = el[j];            // (destination of the fact in direction 1 - dd, compact relation)
          row_flag[e.x] = 1;
          tflag[roff + e.y] = 1;
        }
      }
    }
  }
  __syncthreads();
  // ordered compaction (ballot + wave prefix) into the question's own list segment
  auto compact = [&](const uint8_t* flag, int n, int first, int32_t* list, int32_t* count_out) {
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += kFrThreads) {
      const int i = c0 + tid;
      const bool on = i < n && flag[i] != 0;
      const unsigned long long m = __ballot(on);
      if (lane == 0) s_wsum[wave] = __popcll(m);
      __syncthreads();
      int off = base, tot = 0;
      for (int w = 0; w < kFrThreads / 64; ++w) {
        const int c = s_wsum[w];
        if (w < wave) off += c;
        tot += c;
      }
      if (on) list[off + __popcll(m & ((1ull << lane) - 1))] = first + i;
      base += tot;
      __syncthreads();
    }
    if (tid == 0) *count_out = base;
  };
  compact(row_flag + n0, N, (int)n0, rows + n0, counts + 2 * g);
  compact(tflag + roff, Rg, roff, trows + roff, counts + 2 * g + 1);
}

// ---- relation-table rows of the listed compact rows -----------------------------------------------------------------
struct TabFrArgs {
  const float* T[2];          // [R1, D] relation projections, forward / inverse
  const float* ins;           // [B, I, D]
  const float* W;             // e2e_linear.weight [D, (2I+1) D]
  const int2* rel_rows;       // [rel_total] (question, relation id) of every compact row
  const int32_t* rel_off;     // [B + 1]
  const int32_t* list;        // listed compact rows: question g's at list[rel_off[g] ..]
  const int32_t* counts;      // [B][2] device: (.., listed relation rows) per question
  float* P;                   // [2, rel_total, D]
  int32_t D, I, rel_total, B;
  int32_t dense;              // != 0: no lists - every compact row is computed, one workgroup per (16-row tile, d, cg)
};

// Work item = (question, tile of 16 listed rows, direction, 64-column group); the FOUR WAVES of a workgroup split the
// k extent (K = I * D) between them and add their partial tiles through LDS in wave order: with ~10 listed rows per
// question the kernel is a handful of dependent L2 round trips long, so the k loop is what has to be short.
__global__ __launch_bounds__(256) void k_tables_frontier(TabFrArgs a, int tiles_per_q) {
  __shared__ __attribute__((aligned(16))) float s_part[3][4][64][4];     // [wave 1..3][column tile][lane][4 rows]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int D = a.D, K = a.I * D, ldw = (2 * a.I + 1) * D;
  const int ncg = (D + 63) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // One workgroup per (question, direction, column group) walks the question's tiles: with a seed frontier there is one
  // (a dozen listed relations); launching a workgroup per POSSIBLE tile cost more in dispatch than the work itself
  // (8192 mostly empty workgroups: 20 us at C2 against 6.6 us for one question).
  const int nitem = a.dense ? ((a.rel_total + 15) >> 4) * 2 * ncg : a.B * 2 * ncg;
  for (int item = blockIdx.x; item < nitem; item += gridDim.x)
  for (int tile = a.dense ? item / (2 * ncg) : 0; tile < (a.dense ? item / (2 * ncg) + 1 : tiles_per_q); ++tile) {
    int r = a.dense ? item % (2 * ncg) : item;
    const int cg = r % ncg; r /= ncg;
    const int d = r & 1; r >>= 1;
    const int g = a.dense ? 0 : r;
    const int cnt = a.dense ? a.rel_total : a.counts[2 * g + 1];
    if (tile * 16 >= cnt) break;                       // workgroup-uniform
    const int32_t* list = a.dense ? nullptr : a.list + a.rel_off[g];
    const int c0 = cg * 64;
    const int m = tile * 16 + fr;
    const int prow = a.dense ? (m < cnt ? m : cnt - 1) : list[m < cnt ? m : cnt - 1];
    const int2 br = a.rel_rows[prow];
    const float* trow = a.T[d] + (size_t)br.y * D;
    const float* qrow = a.ins + (size_t)br.x * a.I * D;
    const float* wrow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wrow[nt] = a.W + (size_t)min(c0 + nt * 16 + fr, D - 1) * ldw + (size_t)(1 + d) * D;
    f32x4 acc[4] = {zero4, zero4, zero4, zero4};
    // this wave's k groups of 16: kg = wave, wave + 4, ...; all operands of up to 4 groups are requested before the
    // first MFMA (k = i * D + kk; a lane's float4 never straddles an instruction boundary: D % 4 == 0)
    const int nkg = (K + 15) >> 4;
    for (int kg0 = wave; kg0 < nkg; kg0 += 16) {
      f32x4 av[4], bv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = (kg0 + 4 * u) * 16 + fg * 4;
        av[u] = zero4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[u][nt] = zero4;
        if (k < K) {
          const int i = k / D, kk = k - i * D;
          const f32x4 t = *reinterpret_cast<const f32x4*>(trow + kk);
          const f32x4 q = *reinterpret_cast<const f32x4*>(qrow + (size_t)i * D + kk);
          av[u] = __builtin_elementwise_max(t * q, zero4);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bv[u][nt] = *reinterpret_cast<const f32x4*>(wrow[nt] + (size_t)2 * i * D + kk);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], bv[u][nt][e], acc[nt], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(&s_part[wave - 1][nt][lane][0]) = acc[nt];
    }
    __syncthreads();
    if (wave == 0) {
      // C layout: lane (fr, fg) holds rows 4 fg + r of column slot fr
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 v = ((acc[nt] + *reinterpret_cast<const f32x4*>(&s_part[0][nt][lane][0])) +
                         *reinterpret_cast<const f32x4*>(&s_part[1][nt][lane][0])) +
                        *reinterpret_cast<const f32x4*>(&s_part[2][nt][lane][0]);
        const int col = c0 + nt * 16 + fr;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int mm = tile * 16 + fg * 4 + rr;
          if (mm < cnt && col < D) a.P[((size_t)d * a.rel_total + (a.dense ? mm : list[mm])) * D + col] = v[rr];
        }
      }
    }
    __syncthreads();
  }
}

// ---- neighbour sums of the listed nodes ------------------------------------------------------------------------------
struct WalkFrArgs {
  const int32_t* rp0;
  const int32_t* rp1;
  const int2* edge_m;         // [2F] merged records (source, compact relation; direction 1 offset by Rg + 1)
  const int32_t* m_from;      // [2F] d * F + sorted position (per-fact weights)
  const float* w0;
  const float* w1;
  const float* dist;
  const float* P;             // [2, rel_total, D]
  const int32_t* rel_off;
  const int32_t* rows;        // listed nodes: question g's at rows[g * N ..]
  const int32_t* counts;      // [B][2]
  float* out;                 // [BN + 1, D]; only the listed rows are written
  const int32_t* seeds;       // [B][kFrAltSeeds] a question's seeds, ascending
  const int32_t* seedinfo;    // [B][2] their number (-1: not listed) and the facts that touch them
  int64_t F;
  int32_t N, D, rel_total, B;
};

constexpr int kWfThreads = 256;
constexpr int kWfPer = 8;                  // records per thread and pass
constexpr int kWfCap = kWfThreads * kWfPer;
constexpr int kWfAltMin = 4096;            // rows longer than this may be resolved from the seeds' side

// Work item = (question, k-th listed node): blocks (g, k), k, k + per_q, ... - no global list, no atomics.
__global__ __launch_bounds__(kWfThreads) void k_walk_frontier(WalkFrArgs a, int per_q) {
  __shared__ int2 s_live[kWfCap];          // (p bits, row of P)
  __shared__ int s_wsum[kWfPer * 4];
  __shared__ __attribute__((aligned(16))) float s_red[3][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = a.D;
  const int g = blockIdx.x / per_q;
  const int cnt = a.counts[2 * g];
  const int roff = a.rel_off[g], Rg = a.rel_off[g + 1] - roff;
  const int col = 4 * lane;
  const bool cok = col < D;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int it = blockIdx.x % per_q; it < cnt; it += per_q) {
    const int n = a.rows[(size_t)g * a.N + it];
    const int beg = a.rp0[n] + a.rp1[n], end = a.rp0[n + 1] + a.rp1[n + 1];
    f32x4 acc = zero4;                     // every wave: its share of the live facts, columns 4 lane ..
    bool split = false;                    // workgroup-uniform: the live facts were shared out over the 4 waves
    // A listed HUB (BASELINE config 5: rows of 88 000 facts, of which a handful start at a seed): the same facts are
    // records of the SEEDS' rows with this node as the other end - a fact that arrives at a seed s from n in direction
    // dq arrives at n from s in direction 1 - dq.  When the seeds' rows are much shorter than the node's own, wave 0
    // scans those (seeds ascending, position order: a fixed order) instead of 43 passes over the hub's row.
    const int nseed = a.seedinfo[2 * g];
    const bool alt = nseed > 0 && end - beg > kWfAltMin && a.seedinfo[2 * g + 1] <= ((end - beg) >> 2);
    if (alt && wave == 0) {
      for (int si = 0; si < nseed; ++si) {
        const int sn = a.seeds[g * kFrAltSeeds + si];
        const float ps = a.dist[sn];
        const int sb = a.rp0[sn] + a.rp1[sn], se = a.rp0[sn + 1] + a.rp1[sn + 1];
        for (int c0 = sb; c0 < se; c0 += 64) {
          const int j = c0 + lane;
          const int2 e = j < se ? a.edge_m[j] : make_int2(-1, 0);
          float p = 0.f;
          if (e.x == n) {
            p = ps;
            if (a.w0) {
              const int f = a.m_from[j];
              p *= f < a.F ? a.w0[f] : a.w1[f - a.F];
            }
          }
          unsigned long long m = __ballot(p != 0.f);
          while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const float pl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), l));
            const int rm = __builtin_amdgcn_readlane(e.y, l);
            const int dq = rm > Rg ? 1 : 0;
            const int rl = dq ? rm - (Rg + 1) : rm;
            if (cok) acc += pl * *reinterpret_cast<const f32x4*>(a.P + ((size_t)(1 - dq) * a.rel_total + roff + rl) * D + col);
          }
        }
      }
    }
    for (int c0 = beg; c0 < (alt ? beg : end); c0 += kWfCap) {
      // kWfPer rounds of 256 coalesced records (position = c0 + 256 u + tid), all loads issued before any is looked
      // at: two dependent round trips (records, then dist[src]) per pass of 2048 records
      int2 e[kWfPer];
      float p[kWfPer];
#pragma unroll
      for (int u = 0; u < kWfPer; ++u) {
        const int j = c0 + u * kWfThreads + tid;
        e[u] = j < end ? a.edge_m[j] : make_int2(0, 0);
      }
#pragma unroll
      for (int u = 0; u < kWfPer; ++u) {
        const int j = c0 + u * kWfThreads + tid;
        p[u] = 0.f;
        if (j < end) {
          p[u] = a.dist[e[u].x];
          if (a.w0) {
            const int f = a.m_from[j];
            p[u] *= f < a.F ? a.w0[f] : a.w1[f - a.F];
          }
        }
      }
      // ordered compaction: position order = (round, wave, lane); every (round, wave) count goes to LDS, ONE barrier,
      // then each lane adds up the counts in front of it
      unsigned long long mk[kWfPer];
#pragma unroll
      for (int u = 0; u < kWfPer; ++u) {
        mk[u] = __ballot(p[u] != 0.f);
        if (lane == 0) s_wsum[u * 4 + wave] = __popcll(mk[u]);
      }
      __syncthreads();
      int nl = 0;
#pragma unroll
      for (int u = 0; u < kWfPer; ++u) {
        int off = nl;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int t = s_wsum[u * 4 + w];
          if (w < wave) off += t;
          nl += t;
        }
        if (p[u] != 0.f) {
          const int rm = e[u].y;
          const int d = rm > Rg ? 1 : 0;              // direction 1's relation index sits behind direction 0's zero row
          const int rl = d ? rm - (Rg + 1) : rm;
          s_live[off + __popcll(mk[u] & ((1ull << lane) - 1))] = make_int2(__float_as_int(p[u]), d * a.rel_total + roff + rl);
        }
      }
      __syncthreads();
      // few live facts: wave 0 sums them in position order, one chain (the LDS walk's light-row arithmetic); many:
      // wave w takes facts w, w + 4, ... and the four partial sums are added in wave order at the end
      if (nl <= 32 && !split) {
        if (wave == 0 && cok) {
          for (int k = 0; k < nl; ++k) {
            const int2 lv = s_live[k];
            const f32x4 t = *reinterpret_cast<const f32x4*>(a.P + (size_t)lv.y * D + col);
            acc += __int_as_float(lv.x) * t;
          }
        }
      } else {
        split = true;
        if (cok) {
          for (int k = wave; k < nl; k += 4) {
            const int2 lv = s_live[k];
            const f32x4 t = *reinterpret_cast<const f32x4*>(a.P + (size_t)lv.y * D + col);
            acc += __int_as_float(lv.x) * t;
          }
        }
      }
      __syncthreads();                     // s_live / s_wsum are rewritten by the next pass
    }
    if (split) {
      if (wave > 0) *reinterpret_cast<f32x4*>(&s_red[wave - 1][col]) = acc;
      __syncthreads();
      if (wave == 0)
        acc = ((acc + *reinterpret_cast<const f32x4*>(&s_red[0][col])) + *reinterpret_cast<const f32x4*>(&s_red[1][col])) +
              *reinterpret_cast<const f32x4*>(&s_red[2][col]);
      __syncthreads();
    }
    if (wave == 0 && cok) *reinterpret_cast<f32x4*>(a.out + (size_t)n * D + col) = acc;
  }
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" size_t gnnrag_frontier_workspace_bytes(const gnnrag_csr* csr) {
  if (!csr) return 0;
  return frontier_ws(csr).total;
}

extern "C" int gnnrag_frontier_supported(const gnnrag_csr* csr, int32_t D) {
  return csr && csr->edge_m && csr->m_from && D > 0 && D % 4 == 0 && D <= 256 ? 1 : 0;
}

int gnnrag::frontier_build_z(const gnnrag_csr* csr, const float* dist, void* fws, size_t fws_bytes, float* zero_a,
                             int64_t zero_na, float* zero_b, int64_t zero_nb, hipStream_t stream) {
  if (!csr || !dist || !fws) return GNNRAG_E_BADARG;
  const FrontierWs w = frontier_ws(csr);
  if (fws_bytes < w.total) return GNNRAG_E_WORKSPACE;
  char* base = (char*)fws;
  hipLaunchKernelGGL(k_frontier_build, dim3(csr->B), dim3(kFrThreads), 0, stream, dist, csr->row_ptr[0], csr->row_ptr[1],
                     (const int2*)csr->edge_l[0], (const int2*)csr->edge_l[1], csr->rel_off, csr->N,
                     (uint8_t*)(base + w.row_flag), (int32_t*)(base + w.rows), (uint8_t*)(base + w.tflag),
                     (int32_t*)(base + w.trows), (int32_t*)(base + w.counts), (int32_t*)(base + w.seeds),
                     (int32_t*)(base + w.seedinfo), zero_a, (long long)zero_na, zero_b,
                     (long long)zero_nb);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_frontier_build(const gnnrag_csr* csr, const float* dist, void* fws, size_t fws_bytes,
                                     gnnrag_stream_t stream) {
  return frontier_build_z(csr, dist, fws, fws_bytes, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

const uint8_t* gnnrag::frontier_row_flags(const gnnrag_csr* csr, const void* fws) {
  return (const uint8_t*)fws + frontier_ws(csr).row_flag;
}

// (the two frontier entry points below issue 16-byte accesses: misaligned operands are refused here, the layer driver
// then takes the regular fused path)
static bool fr_aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
                         const void* e = nullptr) {
  return ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15) == 0);
}

extern "C" int gnnrag_relation_tables_frontier(const gnnrag_csr* csr, const void* fws, const float* T_fwd,
                                               const float* T_inv, const float* ins, const float* W, float* P,
                                               int32_t D, int32_t I, gnnrag_stream_t stream) {
  if (!csr || !fws || !T_fwd || !T_inv || !ins || !W || !P || I <= 0) return GNNRAG_E_BADARG;
  if (!gnnrag_frontier_supported(csr, D) || !fr_aligned16(T_fwd, T_inv, ins, W, P)) return GNNRAG_E_UNSUPPORTED;
  if (csr->rel_total == 0) return 0;
  const FrontierWs w = frontier_ws(csr);
  const char* base = (const char*)fws;
  TabFrArgs a;
  a.T[0] = T_fwd; a.T[1] = T_inv; a.ins = ins; a.W = W;
  a.rel_rows = (const int2*)csr->rel_rows;
  a.rel_off = csr->rel_off;
  a.list = (const int32_t*)(base + w.trows);
  a.counts = (const int32_t*)(base + w.counts);
  a.P = P; a.D = D; a.I = I; a.rel_total = csr->rel_total; a.B = csr->B; a.dense = 0;
  const int tiles_per_q = (csr->rel_max + 15) / 16;             // worst case: every relation of a question listed
  const int64_t nitem = (int64_t)csr->B * 2 * ((D + 63) / 64);
  hipLaunchKernelGGL(k_tables_frontier, dim3((unsigned)(nitem < 4096 ? nitem : 4096)), dim3(256), 0, (hipStream_t)stream, a,
                     tiles_per_q);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_aggregate_fused_frontier(const gnnrag_csr* csr, const void* fws, const float* dist,
                                               const float* P, float* out, int32_t D, gnnrag_stream_t stream) {
  if (!csr || !fws || !dist || !P || !out) return GNNRAG_E_BADARG;
  if (!gnnrag_frontier_supported(csr, D) || !fr_aligned16(P, out) || csr->N <= 0) return GNNRAG_E_UNSUPPORTED;
  const FrontierWs w = frontier_ws(csr);
  const char* base = (const char*)fws;
  WalkFrArgs a;
  a.rp0 = csr->row_ptr[0]; a.rp1 = csr->row_ptr[1];
  a.edge_m = (const int2*)csr->edge_m; a.m_from = csr->m_from;
  a.w0 = csr->w_gnn[0]; a.w1 = csr->w_gnn[1];
  a.dist = dist; a.P = P; a.rel_off = csr->rel_off;
  a.rows = (const int32_t*)(base + w.rows);
  a.counts = (const int32_t*)(base + w.counts);
  a.seeds = (const int32_t*)(base + w.seeds);
  a.seedinfo = (const int32_t*)(base + w.seedinfo);
  a.out = out; a.F = csr->F; a.N = csr->N; a.D = D; a.rel_total = csr->rel_total; a.B = csr->B;
  // blocks per question: enough to give every listed node of a seed frontier its own block, bounded chip-wide
  // (about one block per listed node of a seed frontier - a dozen - not per possible node: empty workgroups cost
  // dispatch time, 4096 of them 10 us)
  int per_q = 1024 / csr->B;
  if (per_q < 4) per_q = 4;
  if (per_q > 256) per_q = 256;
  if (per_q > csr->N) per_q = csr->N;
  hipLaunchKernelGGL(k_walk_frontier, dim3((unsigned)(csr->B * per_q)), dim3(kWfThreads), 0, (hipStream_t)stream, a, per_q);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_frontier_read(const gnnrag_csr* csr, const void* fws, int32_t* counts2, uint8_t* row_flag_host,
                                    gnnrag_stream_t stream) {
  if (!csr || !fws || !counts2) return GNNRAG_E_BADARG;
  const FrontierWs w = frontier_ws(csr);
  const char* base = (const char*)fws;
  int32_t* per_q = new int32_t[(size_t)2 * csr->B];
  hipError_t e = hipMemcpyAsync(per_q, base + w.counts, (size_t)2 * csr->B * sizeof(int32_t), hipMemcpyDeviceToHost,
                                (hipStream_t)stream);
  if (e == hipSuccess && row_flag_host)
    e = hipMemcpyAsync(row_flag_host, base + w.row_flag, (size_t)csr->B * csr->N, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  counts2[0] = counts2[1] = 0;
  for (int g = 0; g < csr->B; ++g) {
    counts2[0] += per_q[2 * g];
    counts2[1] += per_q[2 * g + 1];
  }
  delete[] per_q;
  return e == hipSuccess ? 0 : (int)e;
}

// The same kernel over ALL compact rows (no frontier): the relation tables of small batches (one WebQSP question: ~600
// rows x K = I * D = 112) - on the k-tiled generated-operand GEMM that is 11 us of launch / staging / barriers for
// 8 MFLOP.  Exact fp32.  Returns GNNRAG_E_UNSUPPORTED outside its range (the caller takes the GEMM).
int gnnrag::tables_small_launch(const gnnrag_csr* csr, const float* T_fwd, const float* T_inv, const float* ins,
                                const float* W, float* P, int32_t D, int32_t I, hipStream_t stream) {
  if (D % 4 || D > 256 || csr->rel_total <= 0) return GNNRAG_E_UNSUPPORTED;
  if (2.0 * 2 * csr->rel_total * (double)I * D * D > 1.5e8) return GNNRAG_E_UNSUPPORTED;      // a job for the GEMM kernels
  if ((((uintptr_t)T_fwd | (uintptr_t)T_inv | (uintptr_t)ins | (uintptr_t)W) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  TabFrArgs a;
  a.T[0] = T_fwd; a.T[1] = T_inv; a.ins = ins; a.W = W;
  a.rel_rows = (const int2*)csr->rel_rows;
  a.rel_off = csr->rel_off;
  a.list = nullptr; a.counts = nullptr;
  a.P = P; a.D = D; a.I = I; a.rel_total = csr->rel_total; a.B = csr->B; a.dense = 1;
  const int nitem = ((csr->rel_total + 15) / 16) * 2 * ((D + 63) / 64);
  hipLaunchKernelGGL(k_tables_frontier, dim3(nitem), dim3(256), 0, stream, a, 0);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}
