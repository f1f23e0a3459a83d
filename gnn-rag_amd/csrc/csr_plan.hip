// Destination-sorted structure of a batch of question subgraphs (both directions).
//
// Replaces BaseGNNLayer.build_matrix (reference gnn/modules/kg_reasoning/base_gnn.py:19-51),
// which materialises 7 uncoalesced COO tensors from Python lists on the host.  Here the batch
// tuple's int32 arrays are uploaded once and sorted on the device:
//
//   key = destination node (tail for the forward direction, head for the inverse one),
//   value = fact id, stable LSD radix sort over ceil(log2(B*N)) bits  ->  facts of one
//   destination are contiguous and in ascending fact id (one fixed summation order); hub rows
//   (> kHeavyDeg facts) of large vocabularies are then put in (relation, fact id) order by a
//   second stable sort over (segment, relation) keys (see hub_sort_scratch) - also one fixed order.
//
// The sorts are rocPRIM's device radix sorts (plain library ops); everything around them
// (record gather, row pointers, ordered hub lists, relation compaction, merged stream) is hand
// written.  All of it is HBM-bound integer work: coalesced 4/8-byte streams; the only atomics are
// the per-question appends of the LDS walk's big-node lists (order irrelevant).
#include <cstdlib>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "gnnrag_common.h"

namespace gnnrag {

struct CsrLayout {
  size_t row_ptr[2], edge[2], perm[2], w_gnn[2], w_rel[2], heavy[2], chunk_off[2], n_heavy, big_cnt, big_nodes;
  size_t edge_l[2], rel_off, rel_rows, edge_m, m_from, m_dst, hub_q_off[2], hub_wbase[2], hub_qcnt, total;
  int32_t heavy_cap;
};

static CsrLayout csr_layout(int64_t F, int32_t B, int32_t N, int32_t R1, int has_w_gnn, int has_w_rel) {
  CsrLayout L;
  size_t off = 0;
  const size_t BN = (size_t)B * (size_t)N;
  const size_t Fp = (size_t)(F > 0 ? F : 1);
  L.heavy_cap = (int32_t)(Fp / kHeavyDeg + 1);
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  for (int d = 0; d < 2; ++d) L.row_ptr[d] = take((BN + 1) * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.edge[d] = take(Fp * 2 * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.perm[d] = take(Fp * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.w_gnn[d] = has_w_gnn ? take(Fp * sizeof(float)) : 0;
  for (int d = 0; d < 2; ++d) L.w_rel[d] = has_w_rel ? take(Fp * sizeof(float)) : 0;
  for (int d = 0; d < 2; ++d) L.heavy[d] = take((size_t)L.heavy_cap * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.chunk_off[d] = take(((size_t)L.heavy_cap + 1) * sizeof(int32_t));
  L.n_heavy = take(8 * sizeof(int32_t));   // n_heavy[2], n_chunks[2], rel_total, rel_max
  L.big_cnt = take((size_t)B * sizeof(int32_t));
  L.big_nodes = take(BN * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.edge_l[d] = take(Fp * 2 * sizeof(int32_t));
  L.rel_off = take(((size_t)B + 1) * sizeof(int32_t));
  const size_t BR = (size_t)B * (size_t)R1;
  L.rel_rows = take((BR < Fp ? BR : Fp) * 2 * sizeof(int32_t));   // every compact row has >= 1 fact
  L.edge_m = take(2 * Fp * 2 * sizeof(int32_t));
  L.m_from = take(2 * Fp * sizeof(int32_t));
  L.m_dst = take(2 * Fp * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.hub_q_off[d] = take(((size_t)B + 1) * sizeof(int32_t));
  for (int d = 0; d < 2; ++d) L.hub_wbase[d] = take(((size_t)B + 1) * sizeof(int32_t));
  L.hub_qcnt = take((size_t)B * sizeof(int32_t));
  L.total = off;
  return L;
}

static unsigned key_bits(size_t BN) {
  unsigned bits = 1;
  while (bits < 32 && ((size_t)1 << bits) < BN) ++bits;
  return bits;
}

static size_t sort_temp_bytes(int64_t F, unsigned bits) {
  size_t bytes = 0;
  const uint32_t* kin = nullptr;
  uint32_t* kout = nullptr;
  int32_t* vout = nullptr;
  rocprim::counting_iterator<int32_t> vin(0);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)F, 0u, bits,
                                           (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) {
    (void)hipGetLastError();
    bytes = (size_t)16 * (size_t)(F > 0 ? F : 1) + ((size_t)1 << 20);  // conservative bound (no device to ask)
  }
  return bytes;
}

// Hub rows (more than kHeavyDeg facts) are put in RELATION order at plan time (stable inside a relation): the walk adds
// the priors of a run of equal relations and gathers the run's table row once (k_heavy_partial) - a Freebase hub has
// far more facts than distinct relations (BASELINE config 5: 183 000 hub facts of a question, 56 000 distinct (hub,
// relation) pairs).  Every other row keeps the fact order.
//
// The hub rows are disjoint, ascending ranges of the destination-sorted positions, so ONE more stable radix sort does
// it: position i gets the key (segment(i) << relation bits) | relation, where the segments alternate between "the run of
// ordinary rows in front of hub j" (even ids, relation field 0: the stable sort leaves the run as it is) and "hub j" (odd
// ids).  ceil(log2(2 heavy_cap + 1)) + ceil(log2(R1)) bits - 25 at BASELINE config 5, four one-sweep passes over F pairs
// (~0.25 ms) where rocPRIM's segmented sort spent 0.98 ms on the same rows (one workgroup per segment: 37 long segments
// per question do not fill the chip).  The segmented sort stays as the form for keys wider than 32 bits.
// Scratch: two key arrays, a second fact-id array, the segment bounds and rocPRIM's own.
static size_t seg_sort_temp_bytes(int64_t F, int32_t segments, unsigned bits) {
  size_t bytes = 0;
  const uint32_t* kin = nullptr;
  uint32_t* kout = nullptr;
  const int32_t* vin = nullptr;
  int32_t* vout = nullptr;
  const int32_t* off = nullptr;
  hipError_t e = rocprim::segmented_radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (unsigned)F, (unsigned)segments,
                                                     off, off, 0u, bits, (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) {
    (void)hipGetLastError();
    bytes = (size_t)16 * (size_t)(F > 0 ? F : 1) + (size_t)64 * (size_t)segments + ((size_t)1 << 20);
  }
  return bytes;
}

static size_t pair_sort_temp_bytes(int64_t F, unsigned bits) {
  size_t bytes = 0;
  const uint32_t* kin = nullptr;
  uint32_t* kout = nullptr;
  const int32_t* vin = nullptr;
  int32_t* vout = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)F, 0u, bits, (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) {
    (void)hipGetLastError();
    bytes = (size_t)16 * (size_t)(F > 0 ? F : 1) + ((size_t)1 << 20);
  }
  return bytes;
}

// bits of the (segment, relation) key of the hub sort; > 32: the segmented form (GNNRAG_HUB_SORT=segmented asks for it
// at any size - the tests run both forms against the same numpy order)
static unsigned hub_key_bits(int32_t R1, int32_t heavy_cap) {
  const char* e = getenv("GNNRAG_HUB_SORT");        // read per call: a test switches it between two builds
  if (e && strcmp(e, "segmented") == 0) return 64;
  return key_bits((size_t)2 * (size_t)heavy_cap + 2) + key_bits((size_t)R1);
}

// A vocabulary of at most kHubSortMinR1 table rows always fits the LDS walk (aggregate.hip, slice_walk_fits: 2 (R1 + 1)
// 64-byte rows + 4.5 KB <= 159 KB), so the hub rows of such a structure are never walked by the gather kernels: they keep
// the fact order and the build skips the sort (0.4 ms of a 1.5 ms build at BASELINE config 2).
constexpr int kHubSortMinR1 = 1024;

struct HubSortScratch {
  size_t key_in, key_out, perm2, seg, temp, total;
};

static HubSortScratch hub_sort_scratch(int64_t F, int32_t R1, int32_t heavy_cap) {
  HubSortScratch H;
  const size_t Fp = (size_t)(F > 0 ? F : 1);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  if (R1 <= kHubSortMinR1) {
    H.key_in = H.key_out = H.perm2 = H.seg = H.temp = H.total = 0;
    return H;
  }
  H.key_in = take(Fp * sizeof(uint32_t));
  H.key_out = take(Fp * sizeof(uint32_t));
  H.perm2 = take(Fp * sizeof(int32_t));
  H.seg = take((size_t)2 * heavy_cap * sizeof(int32_t));
  const unsigned hb = hub_key_bits(R1, heavy_cap);
  H.temp = take(hb <= 32 ? pair_sort_temp_bytes(F, hb) : seg_sort_temp_bytes(F, heavy_cap, key_bits((size_t)R1)));
  H.total = off;
  return H;
}

// key[i] = relation of the fact at sorted position i
__global__ __launch_bounds__(256) void k_csr_relkey(const int32_t* __restrict__ perm, const int32_t* __restrict__ rels,
                                                    int64_t F, uint32_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < F) key[i] = (uint32_t)rels[perm[i]];
}

// key[i] = (segment of sorted position i, relation of the fact there if the segment is a hub row): see
// seg_sort_temp_bytes' comment.  seg_begin / seg_end are the hub rows in ascending order (k_csr_hub_segments); the
// position's segment is found by a binary search whose bounds the workgroup's first and last position fix in LDS.
__global__ __launch_bounds__(256) void k_csr_segkey(const int32_t* __restrict__ perm, const int32_t* __restrict__ rels,
                                                    int64_t F, const int32_t* __restrict__ seg_begin,
                                                    const int32_t* __restrict__ seg_end,
                                                    const int32_t* __restrict__ count, int32_t cap, unsigned rel_bits,
                                                    uint32_t* __restrict__ key) {
  __shared__ int s_lo, s_hi;
  const int nh = min(*count, cap);
  const int64_t i0 = (int64_t)blockIdx.x * 256;
  auto upper = [&](int64_t pos, int lo, int hi) {      // number of hub rows that begin at or before pos
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)seg_begin[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  if (threadIdx.x == 0) s_lo = upper(i0, 0, nh);
  if (threadIdx.x == 64) s_hi = upper(min(i0 + 255, F - 1), 0, nh);
  __syncthreads();
  const int64_t i = i0 + threadIdx.x;
  if (i >= F) return;
  const int j = upper(i, s_lo, s_hi);
  const bool in_hub = j > 0 && i < (int64_t)seg_end[j - 1];
  // (an out-of-range relation id - the build rejects the tuple afterwards - must not reach the segment bits)
  key[i] = in_hub ? (((uint32_t)(2 * j - 1) << rel_bits) | ((uint32_t)rels[perm[i]] & ((1u << rel_bits) - 1u)))
                  : ((uint32_t)(2 * j) << rel_bits);
}

// the hub rows as sort segments: [row_ptr[n], row_ptr[n + 1]) for the listed nodes, empty for the unused list entries
__global__ __launch_bounds__(256) void k_csr_hub_segments(const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ count, int32_t cap,
                                                          int32_t* __restrict__ seg_begin, int32_t* __restrict__ seg_end) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= cap) return;
  int b = 0, n = 0;
  if (e < min(*count, cap)) {
    const int node = list[e];
    b = row_ptr[node];
    n = row_ptr[node + 1];
  }
  seg_begin[e] = b;
  seg_end[e] = n;
}

// One thread per sorted position: gather the (source, relation) record and the weights of the
// fact that landed there.  Reads perm coalesced, gathers 3-5 words, writes coalesced.
__global__ __launch_bounds__(256) void k_csr_fill(const int32_t* __restrict__ perm,
                                                  const int32_t* __restrict__ src,
                                                  const int32_t* __restrict__ rels,
                                                  const float* __restrict__ wg,
                                                  const float* __restrict__ wr, int64_t F,
                                                  const int32_t* __restrict__ g2l, int64_t g2l_len, int32_t N,
                                                  int32_t R1, int2* __restrict__ edge, int2* __restrict__ edge_l,
                                                  float* __restrict__ wg_out, float* __restrict__ wr_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const int32_t f = perm[i];
  const int32_t s = src[f], r = rels[f];
  edge[i] = make_int2(s, r);
  // the same fact with its relation renumbered inside its question (fused path)
  // (an invalid tuple is rejected by the build after these kernels; keep the lookup in bounds meanwhile)
  int rl = 0;
  if ((unsigned)r < (unsigned)R1 && s >= 0) {
    const int64_t idx = (int64_t)(s / N) * R1 + r;
    if (idx < g2l_len) rl = g2l[idx];
  }
  edge_l[i] = make_int2(s, rl);
  if (wg_out) {
    const float v = wg[f];
    wg_out[i] = v * v;  // the weight enters head2fact AND fact2tail (base_gnn.py:44-47)
  }
  if (wr_out) wr_out[i] = wr[f];
}

// row_ptr[n] = first sorted position whose destination is >= n  (binary search, n in [0, BN]).
__global__ __launch_bounds__(256) void k_csr_row_ptr(const uint32_t* __restrict__ keys, int64_t F,
                                                     int64_t BN, int32_t* __restrict__ row_ptr) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > BN) return;
  int64_t lo = 0, hi = F;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < n) lo = mid + 1; else hi = mid;
  }
  row_ptr[n] = (int32_t)lo;
}

// The heavy rows (hubs) of a direction, listed in ascending node order - i.e. question by question - with the
// per-question offsets the dense hub form of the gather walk needs (aggregate.hip, k_hub_dense):
//   hub_q_off[b]  first list entry of question b          (hub_q_off[B] = n_heavy)
//   hub_wbase[b]  sum over earlier questions of hubs x relations-in-use (rounded up to 4): offset of question b's
//                 hub-by-relation weight block; saturates at INT32_MAX
// Two launches of one workgroup per question: count, then an ordered compaction behind the earlier questions' counts.
__global__ __launch_bounds__(256) void k_csr_hub_count(const int32_t* __restrict__ row_ptr, int32_t N,
                                                       int32_t heavy_deg, int32_t* __restrict__ qcnt) {
  __shared__ int s[4];
  const int b = blockIdx.x;
  const int32_t* rp = row_ptr + (int64_t)b * N;
  int c = 0;
  for (int j = threadIdx.x; j < N; j += 256) c += (rp[j + 1] - rp[j]) > heavy_deg ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) qcnt[b] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void k_csr_hub_fill(const int32_t* __restrict__ row_ptr, int32_t N, int32_t B,
                                                      int32_t heavy_deg, const int32_t* __restrict__ qcnt,
                                                      const int32_t* __restrict__ rel_off, int32_t* __restrict__ list,
                                                      int32_t cap, int32_t* __restrict__ count,
                                                      int32_t* __restrict__ hub_q_off, int32_t* __restrict__ hub_wbase) {
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {      // a batch has at most a few hundred questions: serial prefix
    int base = 0;
    long long wb = 0;
    for (int q = 0; q < b; ++q) {
      base += qcnt[q];
      wb += (long long)qcnt[q] * (((rel_off[q + 1] - rel_off[q]) + 3) & ~3);
    }
    s_base = base;
    hub_q_off[b] = base;
    hub_wbase[b] = wb > 0x7fffffffLL ? 0x7fffffff : (int)wb;
    if (b == B - 1) {
      base += qcnt[b];
      wb += (long long)qcnt[b] * (((rel_off[b + 1] - rel_off[b]) + 3) & ~3);
      hub_q_off[B] = base;
      hub_wbase[B] = wb > 0x7fffffffLL ? 0x7fffffff : (int)wb;
      *count = base;
    }
  }
  __syncthreads();
  int run = s_base;
  const int32_t* rp = row_ptr + (int64_t)b * N;
  for (int j0 = 0; j0 < N; j0 += 256) {
    const int j = j0 + tid;
    const bool on = j < N && (rp[j + 1] - rp[j]) > heavy_deg;
    const unsigned long long m = __ballot(on);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int off = run, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) off += s_w[w];
      tot += s_w[w];
    }
    if (on) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1));
      if (pos < cap) list[pos] = (int32_t)((int64_t)b * N + j);
    }
    run += tot;
    __syncthreads();
  }
}

// the merged record stream (see gnnrag.h: edge_m / m_from): node n's facts of direction 0 come first in the node's run
// [rp0[n] + rp1[n], rp0[n+1] + rp1[n+1]), then direction 1's, whose compact relation index is moved behind the first
// table slice (+ the question's relation count + 1)
__global__ __launch_bounds__(256) void k_csr_merge(const int32_t* __restrict__ perm0, const int32_t* __restrict__ perm1,
                                                   const int32_t* __restrict__ heads, const int32_t* __restrict__ tails,
                                                   const int32_t* __restrict__ rp0, const int32_t* __restrict__ rp1,
                                                   const int2* __restrict__ el0, const int2* __restrict__ el1,
                                                   const int32_t* __restrict__ rel_off, int32_t N, int64_t F,
                                                   int64_t BN, int2* __restrict__ edge_m, int32_t* __restrict__ m_from,
                                                   int32_t* __restrict__ m_dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= F) return;
  // (an invalid tuple - node id outside [0, B*N) - is rejected by the build AFTER these kernels have run: such a fact
  // must not index the row pointers.  Round 2's version did, and a negative id wrote edge_m far out of bounds.)
  const int nd = blockIdx.y == 0 ? tails[perm0[i]] : heads[perm1[i]];
  if (nd < 0 || nd >= BN) return;
  if (blockIdx.y == 0) {
    const int n = nd;                             // destination in direction 0
    const int64_t m = i + rp1[n];
    edge_m[m] = el0[i];
    m_from[m] = (int32_t)i;
    m_dst[m] = n;
  } else {
    const int n = nd;                             // destination in direction 1
    const int64_t m = i + rp0[n + 1];
    int2 e = el1[i];
    const int q = n / N;
    e.y += rel_off[q + 1] - rel_off[q] + 1;
    edge_m[m] = e;
    m_from[m] = (int32_t)(F + i);
    m_dst[m] = n;
  }
}

// Per-question list of "big" nodes (more than kBigDeg facts in a direction): the LDS walk hands them
// to whole waves / the whole workgroup instead of a 4-lane group.  Order inside a question's list is
// whatever the atomics give; results do not depend on it (each row's own sum order is fixed).
__global__ __launch_bounds__(256) void k_csr_big(const int32_t* __restrict__ rp0, const int32_t* __restrict__ rp1,
                                                 int64_t BN, int32_t N, int32_t big_deg,
                                                 int32_t* __restrict__ big_cnt, int32_t* __restrict__ big_nodes) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= BN) return;
  const int l0 = rp0[n + 1] - rp0[n], l1 = rp1[n + 1] - rp1[n];
  if (max(l0, l1) > big_deg) {
    const int b = (int)(n / N);
    const int pos = atomicAdd(&big_cnt[b], 1);
    big_nodes[(size_t)b * N + pos] = (int32_t)n;
  }
}

// Heavy rows are walked in chunks of kHeavyDeg facts: chunk_off[e] = first chunk of heavy entry e
// (exclusive prefix over ceil(deg/kHeavyDeg)), one 1024-thread workgroup per direction.
__global__ __launch_bounds__(1024) void k_csr_heavy_chunks(const int32_t* __restrict__ rp0,
                                                           const int32_t* __restrict__ rp1,
                                                           const int32_t* __restrict__ list0,
                                                           const int32_t* __restrict__ list1,
                                                           const int32_t* __restrict__ n_heavy, int32_t cap,
                                                           int32_t* __restrict__ off0, int32_t* __restrict__ off1,
                                                           int32_t* __restrict__ n_chunks) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int d = blockIdx.x;
  const int32_t* rp = d ? rp1 : rp0;
  const int32_t* list = d ? list1 : list0;
  int32_t* off = d ? off1 : off0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cnt = min(n_heavy[d], cap);
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < cnt; base += 1024) {
    const int i = base + tid;
    int v = 0;
    if (i < cnt) {
      const int n = list[i];
      v = (rp[n + 1] - rp[n] + kHeavyDeg - 1) / kHeavyDeg;
    }
    int x = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int wp = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) wp += wsum[w];
      total += wsum[w];
    }
    const int carry = carry_s;
    if (i < cnt) off[i] = carry + wp + x - v;
    __syncthreads();
    if (tid == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (tid == 0) {
    off[cnt] = carry_s;
    n_chunks[d] = carry_s;
  }
}

// ---- per-question relation compaction ---------------------------------------------------------
// exclusive prefix of one int per thread over a 1024-thread workgroup (wsum: 16 ints of LDS)
__device__ __forceinline__ int block_scan_excl(int v, int* wsum, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int wp = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < wave) wp += wsum[w];
    total += wsum[w];
  }
  __syncthreads();
  return wp + x - v;
}

// g2l[b*R1 + r] = 1 for every (question, relation) pair that occurs (benign write race)
// The same pass validates the tuple (the host no longer does: 1.2 ms of numpy per C2 batch): node ids in
// [0, B*N), relation ids in [0, R1), head and tail in the same question -> error bits, checked by the build.
__global__ __launch_bounds__(256) void k_rel_flag(const int32_t* __restrict__ heads,
                                                  const int32_t* __restrict__ rels,
                                                  const int32_t* __restrict__ tails, int64_t F, int32_t N,
                                                  int32_t R1, int32_t B, int32_t* __restrict__ g2l,
                                                  int32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const int32_t r = rels[i], h = heads[i], t = tails[i];
  const int64_t BN = (int64_t)B * N;
  int bad = 0;
  if (h < 0 || h >= BN || t < 0 || t >= BN) bad |= 1;
  if ((unsigned)r >= (unsigned)R1) bad |= 2;
  const int32_t b = h / N;
  if (!bad && b != t / N) bad |= 4;
  if (bad) atomicOr(err, bad);
  else g2l[(size_t)b * R1 + r] = 1;
}

// one workgroup per question: flags -> compact index (or -1), number of used relations -> cnt[b]
__global__ __launch_bounds__(1024) void k_rel_scan(int32_t* __restrict__ g2l, int32_t R1,
                                                   int32_t* __restrict__ cnt) {
  __shared__ int wsum[16];
  int32_t* row = g2l + (size_t)blockIdx.x * R1;
  int carry = 0;
  for (int base = 0; base < R1; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = (i < R1) ? row[i] : 0;
    int total;
    const int ex = block_scan_excl(v, wsum, total);
    if (i < R1) row[i] = v ? carry + ex : -1;
    carry += total;
  }
  if (threadIdx.x == 0) cnt[blockIdx.x] = carry;
}

// rel_off[1..B] holds the per-question counts on entry; prefix sum in place, totals to stats
__global__ __launch_bounds__(1024) void k_rel_off(int32_t* __restrict__ rel_off, int32_t B,
                                                  int32_t* __restrict__ stats) {
  __shared__ int wsum[16];
  __shared__ int mx;
  if (threadIdx.x == 0) mx = 0;
  __syncthreads();
  int carry = 0;
  for (int base = 0; base < B; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = (i < B) ? rel_off[i + 1] : 0;
    atomicMax(&mx, v);
    int total;
    const int ex = block_scan_excl(v, wsum, total);
    if (i < B) rel_off[i + 1] = carry + ex + v;
    carry += total;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    rel_off[0] = 0;
    stats[0] = carry;
    stats[1] = mx;
  }
}

__global__ __launch_bounds__(256) void k_rel_rows(const int32_t* __restrict__ g2l,
                                                  const int32_t* __restrict__ rel_off, int32_t R1, int64_t BR,
                                                  int2* __restrict__ rel_rows) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= BR) return;
  const int l = g2l[idx];
  if (l < 0) return;
  const int b = (int)(idx / R1);
  rel_rows[rel_off[b] + l] = make_int2(b, (int)(idx - (int64_t)b * R1));
}

// ---- facts ordered by (question, relation): the backward's gather structure ----------------------
// key[f] = compact relation row of fact f: position of rel_f in its question's sorted relation list
__global__ __launch_bounds__(256) void k_rel_key(const int32_t* __restrict__ heads, const int32_t* __restrict__ rels,
                                                 int64_t F, int32_t N, const int32_t* __restrict__ rel_off,
                                                 const int2* __restrict__ rel_rows, uint32_t* __restrict__ key) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int b = heads[f] / N, r = rels[f];
  int lo = rel_off[b], hi = rel_off[b + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rel_rows[mid].y < r) lo = mid + 1; else hi = mid;
  }
  key[f] = (uint32_t)lo;
}

__global__ __launch_bounds__(256) void k_relorder_fill(const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ heads,
                                                       const int32_t* __restrict__ tails,
                                                       const float* __restrict__ w, int64_t F,
                                                       int2* __restrict__ ht, float* __restrict__ w_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const int32_t f = perm[i];
  ht[i] = make_int2(heads[f], tails[f]);
  if (w_out) {
    const float v = w[f];
    w_out[i] = v * v;
  }
}

// chunk_ptr = exclusive prefix of ceil(len/256) over the rows (single workgroup; rows <= a few 10^5)
__global__ __launch_bounds__(1024) void k_relorder_chunks(const int32_t* __restrict__ row_ptr, int32_t R,
                                                          int32_t* __restrict__ chunk_ptr, int32_t* __restrict__ total) {
  __shared__ int wsum[16];
  int carry = 0;
  for (int base = 0; base < R; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = (i < R) ? (row_ptr[i + 1] - row_ptr[i] + kHeavyDeg - 1) / kHeavyDeg : 0;
    int tot;
    const int ex = block_scan_excl(v, wsum, tot);
    if (i < R) chunk_ptr[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    chunk_ptr[R] = carry;
    *total = carry;
  }
}

__global__ __launch_bounds__(256) void k_csr_permute_weight(const int32_t* __restrict__ perm,
                                                            const float* __restrict__ w, int64_t F, int square,
                                                            float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const float v = w[perm[i]];
  out[i] = square ? v * v : v;
}


// ---- batch structure = concatenation of cached per-question structures --------------------
// Questions are disjoint node ranges (heads / tails are offset by i * N, dataset_load.py:483) and the structure is
// sorted by destination node, so the batch's sorted order IS the questions' sorted orders one after the other: no sort,
// no search - every array is a copy with the question's node / fact / relation-row offset added.
constexpr int kConcatChunk = 24;          // questions per launch: their table travels as a kernel argument (< 4 KB)
struct ConcatPart {
  const int32_t* row_ptr[2];
  const int2* edge[2];
  const int2* edge_l[2];
  const int32_t* perm[2];
  const int2* edge_m;
  const int32_t* m_from;
  const int32_t* m_dst;
  const int2* rel_rows;
  int64_t foff;      // first fact of the question in the batch
  int32_t roff;      // first compact relation row of the question
  int32_t Fq, Rq, b;
};
struct ConcatArgs {
  ConcatPart part[kConcatChunk];
  int32_t n;         // questions in this launch
  int32_t N;
  int64_t F;         // facts of the whole batch
  int32_t* row_ptr[2];
  int2* edge[2];
  int2* edge_l[2];
  int32_t* perm[2];
  int2* edge_m;
  int32_t* m_from;
  int32_t* m_dst;
  int2* rel_rows;
  int32_t* rel_off;
};

// blockIdx.y = question of the chunk; blockIdx.x strides over the question's items; blockIdx.z: 0 / 1 = the per-
// direction arrays, 2 = the merged stream (2 Fq records), 3 = row pointers + relation rows
__global__ __launch_bounds__(256) void k_csr_concat(const ConcatArgs a) {
  const ConcatPart& q = a.part[blockIdx.y];
  const int what = blockIdx.z;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int node0 = q.b * a.N;
  if (what < 2) {
    const int d = what;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < q.Fq; i += stride) {
      int2 e = q.edge[d][i], el = q.edge_l[d][i];
      e.x += node0;
      el.x += node0;
      a.edge[d][q.foff + i] = e;
      a.edge_l[d][q.foff + i] = el;
      a.perm[d][q.foff + i] = q.perm[d][i] + (int32_t)q.foff;
    }
  } else if (what == 2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * (int64_t)q.Fq; i += stride) {
      int2 e = q.edge_m[i];
      e.x += node0;
      const int f = q.m_from[i];                       // d * Fq + position in direction d
      a.edge_m[2 * q.foff + i] = e;
      a.m_from[2 * q.foff + i] = f < q.Fq ? f + (int32_t)q.foff : (int32_t)(a.F + (f - q.Fq) + q.foff);
      a.m_dst[2 * q.foff + i] = q.m_dst[i] + node0;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.N; i += stride) {
      a.row_ptr[0][node0 + i] = q.row_ptr[0][i] + (int32_t)q.foff;
      a.row_ptr[1][node0 + i] = q.row_ptr[1][i] + (int32_t)q.foff;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < q.Rq; i += stride)
      a.rel_rows[q.roff + i] = make_int2(q.b, q.rel_rows[i].y);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.rel_off[q.b] = q.roff;
  }
}

__global__ void k_csr_concat_tail(int32_t* rp0, int32_t* rp1, int32_t* rel_off, int64_t BN, int32_t B, int32_t F,
                                  int32_t rel_total) {
  rp0[BN] = F;
  rp1[BN] = F;
  rel_off[B] = rel_total;
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_csr_permute_weight(const gnnrag_csr* csr, const float* w_per_fact, int square,
                                         float* out_fwd, float* out_inv, gnnrag_stream_t stream) {
  if (!csr || !out_fwd || !out_inv) return GNNRAG_E_BADARG;
  if (csr->F == 0) return 0;
  if (!w_per_fact) return GNNRAG_E_BADARG;
  const int nb = (int)((csr->F + 255) / 256);
  float* outs[2] = {out_fwd, out_inv};
  for (int d = 0; d < 2; ++d) {
    hipLaunchKernelGGL(k_csr_permute_weight, dim3(nb), dim3(256), 0, (hipStream_t)stream, csr->perm[d],
                       w_per_fact, csr->F, square, outs[d]);
    GNNRAG_LAUNCH_CHECK();
  }
  return 0;
}

struct RelLayout { size_t ht, perm, w, row_ptr, chunk_ptr, total_cnt, total; };

static RelLayout rel_layout(const gnnrag_csr* csr, int has_w) {
  RelLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t Fp = (size_t)(csr->F > 0 ? csr->F : 1);
  const size_t R = (size_t)(csr->rel_total > 0 ? csr->rel_total : 0);
  L.ht = take(Fp * 2 * sizeof(int32_t));
  L.perm = take(Fp * sizeof(int32_t));
  L.w = has_w ? take(Fp * sizeof(float)) : 0;
  L.row_ptr = take((R + 1) * sizeof(int32_t));
  L.chunk_ptr = take((R + 1) * sizeof(int32_t));
  L.total_cnt = take(sizeof(int32_t));
  L.total = off;
  return L;
}

extern "C" size_t gnnrag_relorder_bytes(const gnnrag_csr* csr, int has_w) {
  return csr ? rel_layout(csr, has_w).total : 0;
}

extern "C" size_t gnnrag_relorder_scratch_bytes(const gnnrag_csr* csr) {
  if (!csr) return 0;
  const size_t Fp = (size_t)(csr->F > 0 ? csr->F : 1);
  const unsigned bits = key_bits((size_t)(csr->rel_total > 1 ? csr->rel_total : 2));
  // unsorted keys, sorted keys, sort temporaries
  return 2 * align_up(Fp * sizeof(uint32_t), 256) + align_up(sort_temp_bytes(csr->F, bits), 256);
}

extern "C" int gnnrag_relorder_build(const gnnrag_csr* csr, const int32_t* heads, const int32_t* rels,
                                     const int32_t* tails, const float* w_per_fact, void* mem, size_t mem_bytes,
                                     void* scratch, size_t scratch_bytes, gnnrag_relorder* out,
                                     gnnrag_stream_t stream_) {
  if (!csr || !out || !mem || csr->rel_total < 0) return GNNRAG_E_BADARG;
  const int64_t F = csr->F;
  if (F > 0 && (!heads || !rels || !tails || !scratch)) return GNNRAG_E_BADARG;
  hipStream_t stream = (hipStream_t)stream_;
  const RelLayout L = rel_layout(csr, w_per_fact != nullptr);
  if (mem_bytes < L.total) return GNNRAG_E_WORKSPACE;
  if (scratch_bytes < gnnrag_relorder_scratch_bytes(csr)) return GNNRAG_E_WORKSPACE;
  char* base = (char*)mem;
  memset(out, 0, sizeof(*out));
  out->F = F;
  out->rel_total = csr->rel_total;
  out->ht = (int32_t*)(base + L.ht);
  out->perm = (int32_t*)(base + L.perm);
  out->w = w_per_fact ? (float*)(base + L.w) : nullptr;
  out->row_ptr = (int32_t*)(base + L.row_ptr);
  out->chunk_ptr = (int32_t*)(base + L.chunk_ptr);
  int32_t* total = (int32_t*)(base + L.total_cnt);
  const int32_t R = csr->rel_total;
  if (F == 0 || R == 0) {
    GNNRAG_HIP(hipMemsetAsync(out->row_ptr, 0, ((size_t)R + 1) * sizeof(int32_t), stream));
    GNNRAG_HIP(hipMemsetAsync(out->chunk_ptr, 0, ((size_t)R + 1) * sizeof(int32_t), stream));
    out->n_chunks = 0;
    return 0;
  }
  const size_t kb = align_up((size_t)F * sizeof(uint32_t), 256);
  uint32_t* key = (uint32_t*)scratch;
  uint32_t* key_sorted = (uint32_t*)((char*)scratch + kb);
  int32_t* perm = out->perm;
  void* temp = (char*)scratch + 2 * kb;
  const unsigned bits = key_bits((size_t)(R > 1 ? R : 2));
  const int nb = (int)((F + 255) / 256);
  hipLaunchKernelGGL(k_rel_key, dim3(nb), dim3(256), 0, stream, heads, rels, F, csr->N, csr->rel_off,
                     (const int2*)csr->rel_rows, key);
  GNNRAG_LAUNCH_CHECK();
  rocprim::counting_iterator<int32_t> iota(0);
  size_t tb = align_up(sort_temp_bytes(F, bits), 256);
  GNNRAG_HIP(rocprim::radix_sort_pairs(temp, tb, (const uint32_t*)key, key_sorted, iota, perm, (size_t)F, 0u, bits,
                                       stream, false));
  hipLaunchKernelGGL(k_relorder_fill, dim3(nb), dim3(256), 0, stream, perm, heads, tails, w_per_fact, F,
                     (int2*)out->ht, out->w);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_csr_row_ptr, dim3((R + 1 + 255) / 256), dim3(256), 0, stream, key_sorted, F, (int64_t)R,
                     out->row_ptr);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_relorder_chunks, dim3(1), dim3(1024), 0, stream, out->row_ptr, R, out->chunk_ptr, total);
  GNNRAG_LAUNCH_CHECK();
  int32_t n = 0;
  GNNRAG_HIP(hipMemcpyAsync(&n, total, sizeof(n), hipMemcpyDeviceToHost, stream));
  GNNRAG_HIP(hipStreamSynchronize(stream));
  out->n_chunks = n;
  return 0;
}

// Host helper of the boundary: the reference's tuple holds int64 ids (dataset_load.py:527), the structure int32.
// Narrows the three arrays into one [3, F] int32 block with a few threads and checks on the way that every id
// fits (numpy's astype + three max passes cost 18 ms per C5 batch of 7 M facts - more than the GPU step).
extern "C" int gnnrag_narrow_tuple(const int64_t* heads, const int64_t* rels, const int64_t* tails, int64_t F,
                                   int32_t* out, int32_t nthreads) {
  if (F < 0 || (F > 0 && (!heads || !rels || !tails || !out))) return GNNRAG_E_BADARG;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 16) nthreads = 16;
  if (F < (1 << 16)) nthreads = 1;
  const int64_t* src[3] = {heads, rels, tails};
  std::vector<int> bad((size_t)nthreads, 0);
  auto work = [&](int t) {
    const int64_t lo = F * t / nthreads, hi = F * (t + 1) / nthreads;
    uint64_t acc = 0;
    for (int a = 0; a < 3; ++a) {
      const int64_t* s = src[a];
      int32_t* d = out + (size_t)a * F;
      for (int64_t i = lo; i < hi; ++i) {
        const int64_t v = s[i];
        acc |= (uint64_t)v;
        d[i] = (int32_t)v;
      }
    }
    bad[t] = (acc >> 31) != 0;        // negative or >= 2^31 somewhere
  };
  if (nthreads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < nthreads; ++t)
    if (bad[t]) return GNNRAG_E_TUPLE;
  return 0;
}

static int hub_lists(gnnrag_csr* out, int d, int32_t* qcnt, hipStream_t stream) {
  hipLaunchKernelGGL(k_csr_hub_count, dim3(out->B), dim3(256), 0, stream, out->row_ptr[d], out->N, (int32_t)kHeavyDeg, qcnt);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_csr_hub_fill, dim3(out->B), dim3(256), 0, stream, out->row_ptr[d], out->N, out->B,
                     (int32_t)kHeavyDeg, qcnt, out->rel_off, out->heavy[d], out->heavy_cap, out->n_heavy + d,
                     out->hub_q_off[d], out->hub_wbase[d]);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gnnrag_csr_bytes(int64_t F, int32_t B, int32_t N, int32_t R1, int has_w_gnn, int has_w_rel) {
  if (F < 0 || B <= 0 || N <= 0 || R1 <= 0) return 0;
  return csr_layout(F, B, N, R1, has_w_gnn, has_w_rel).total;
}

extern "C" size_t gnnrag_csr_scratch_bytes(int64_t F, int32_t B, int32_t N, int32_t R1) {
  if (F < 0 || B <= 0 || N <= 0 || R1 <= 0) return 0;
  const size_t Fp = (size_t)(F > 0 ? F : 1);
  const unsigned bits = key_bits((size_t)B * (size_t)N);
  return align_up(Fp * sizeof(uint32_t), 256) + align_up(sort_temp_bytes(F, bits), 256) +
         align_up((size_t)B * (size_t)R1 * sizeof(int32_t), 256) +
         hub_sort_scratch(F, R1, (int32_t)(Fp / kHeavyDeg + 1)).total;
}

extern "C" int gnnrag_csr_build(const int32_t* heads, const int32_t* rels, const int32_t* tails,
                                const float* w_gnn, const float* w_rel, int64_t F, int32_t B, int32_t N,
                                int32_t R1, void* csr_mem, size_t csr_bytes, void* scratch,
                                size_t scratch_bytes, gnnrag_csr* out, gnnrag_stream_t stream_) {
  return gnnrag_csr_build_counts(heads, rels, tails, w_gnn, w_rel, F, B, N, R1, -1, -1, csr_mem, csr_bytes, scratch,
                                 scratch_bytes, out, stream_);
}

extern "C" int gnnrag_csr_status(const gnnrag_csr* csr, gnnrag_stream_t stream_) {
  if (!csr || !csr->n_heavy) return GNNRAG_E_BADARG;
  int32_t stats[3] = {0, 0, 0};     // rel_total, rel_max, validation error bits (as the build left them on the device)
  GNNRAG_HIP(hipMemcpyAsync(stats, csr->n_heavy + 4, sizeof(stats), hipMemcpyDeviceToHost, (hipStream_t)stream_));
  GNNRAG_HIP(hipStreamSynchronize((hipStream_t)stream_));
  if (stats[2]) return GNNRAG_E_TUPLE;
  if (csr->F > 0 && (stats[0] != csr->rel_total || stats[1] != csr->rel_max)) return GNNRAG_E_BADARG;
  return 0;
}

extern "C" int gnnrag_csr_build_counts(const int32_t* heads, const int32_t* rels, const int32_t* tails,
                                       const float* w_gnn, const float* w_rel, int64_t F, int32_t B, int32_t N,
                                       int32_t R1, int32_t rel_total, int32_t rel_max, void* csr_mem, size_t csr_bytes,
                                       void* scratch, size_t scratch_bytes, gnnrag_csr* out, gnnrag_stream_t stream_) {
  if (!out || !csr_mem || B <= 0 || N <= 0 || R1 <= 0 || F < 0) return GNNRAG_E_BADARG;
  if (F > 0 && (!heads || !rels || !tails || !scratch)) return GNNRAG_E_BADARG;
  const int64_t BN = (int64_t)B * N;
  if (BN >= ((int64_t)1 << 31) || F >= ((int64_t)1 << 31)) return GNNRAG_E_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  if ((int64_t)B * R1 >= ((int64_t)1 << 31)) return GNNRAG_E_UNSUPPORTED;
  const CsrLayout L = csr_layout(F, B, N, R1, w_gnn != nullptr, w_rel != nullptr);
  if (csr_bytes < L.total) return GNNRAG_E_WORKSPACE;
  char* base = (char*)csr_mem;
  memset(out, 0, sizeof(*out));
  out->B = B; out->N = N; out->R1 = R1; out->F = F;
  out->heavy_deg = kHeavyDeg;
  out->heavy_cap = L.heavy_cap;
  out->hub_sorted = R1 > kHubSortMinR1 ? 1 : 0;
  for (int d = 0; d < 2; ++d) {
    out->row_ptr[d] = (int32_t*)(base + L.row_ptr[d]);
    out->edge[d] = (int32_t*)(base + L.edge[d]);
    out->perm[d] = (int32_t*)(base + L.perm[d]);
    out->w_gnn[d] = w_gnn ? (float*)(base + L.w_gnn[d]) : nullptr;
    out->w_rel[d] = w_rel ? (float*)(base + L.w_rel[d]) : nullptr;
    out->heavy[d] = (int32_t*)(base + L.heavy[d]);
    out->chunk_off[d] = (int32_t*)(base + L.chunk_off[d]);
    out->edge_l[d] = (int32_t*)(base + L.edge_l[d]);
  }
  out->rel_off = (int32_t*)(base + L.rel_off);
  out->edge_m = (int32_t*)(base + L.edge_m);
  out->m_from = (int32_t*)(base + L.m_from);
  out->m_dst = (int32_t*)(base + L.m_dst);
  for (int d = 0; d < 2; ++d) {
    out->hub_q_off[d] = (int32_t*)(base + L.hub_q_off[d]);
    out->hub_wbase[d] = (int32_t*)(base + L.hub_wbase[d]);
  }
  out->rel_rows = (int32_t*)(base + L.rel_rows);
  out->n_heavy = (int32_t*)(base + L.n_heavy);
  out->n_chunks = out->n_heavy + 2;
  out->big_cnt = (int32_t*)(base + L.big_cnt);
  out->big_nodes = (int32_t*)(base + L.big_nodes);
  out->big_deg = kBigDeg;
  GNNRAG_HIP(hipMemsetAsync(out->big_cnt, 0, (size_t)B * sizeof(int32_t), stream));
  out->max_chunks = 2 * L.heavy_cap;   // sum ceil(deg/256) over rows with deg > 256 < F/256 + F/257
  GNNRAG_HIP(hipMemsetAsync(out->n_heavy, 0, 8 * sizeof(int32_t), stream));
  GNNRAG_HIP(hipMemsetAsync(out->rel_off, 0, ((size_t)B + 1) * sizeof(int32_t), stream));
  int32_t* rel_stats = out->n_heavy + 4;

  const unsigned bits = key_bits((size_t)BN);
  const size_t keys_bytes = align_up((size_t)(F > 0 ? F : 1) * sizeof(uint32_t), 256);
  size_t temp_bytes = 0;
  if (F > 0) {
    temp_bytes = sort_temp_bytes(F, bits);
    temp_bytes = align_up(temp_bytes, 256);
    if (scratch_bytes < gnnrag_csr_scratch_bytes(F, B, N, R1)) return GNNRAG_E_WORKSPACE;
  }
  uint32_t* keys_sorted = (uint32_t*)scratch;
  void* temp = (char*)scratch + keys_bytes;
  int32_t* g2l = (int32_t*)((char*)scratch + keys_bytes + temp_bytes);
  const HubSortScratch H = hub_sort_scratch(F, R1, L.heavy_cap);
  char* hub_base = (char*)scratch + keys_bytes + temp_bytes + align_up((size_t)B * (size_t)R1 * sizeof(int32_t), 256);

  // relations each question uses -> compact numbering (before the fills, which store it per fact)
  if (F > 0) {
    const int64_t BR = (int64_t)B * R1;
    GNNRAG_HIP(hipMemsetAsync(g2l, 0, (size_t)BR * sizeof(int32_t), stream));
    hipLaunchKernelGGL(k_rel_flag, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, stream, heads, rels, tails, F,
                       N, R1, B, g2l, rel_stats + 2);
    GNNRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rel_scan, dim3(B), dim3(1024), 0, stream, g2l, R1, out->rel_off + 1);
    GNNRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rel_off, dim3(1), dim3(1024), 0, stream, out->rel_off, B, rel_stats);
    GNNRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rel_rows, dim3((unsigned)((BR + 255) / 256)), dim3(256), 0, stream, g2l, out->rel_off,
                       R1, BR, (int2*)out->rel_rows);
    GNNRAG_LAUNCH_CHECK();
  }

  const int nb_rows = (int)((BN + 1 + 255) / 256);
  for (int d = 0; d < 2; ++d) {
    const int32_t* dst = d == 0 ? tails : heads;
    const int32_t* src = d == 0 ? heads : tails;
    if (F > 0) {
      rocprim::counting_iterator<int32_t> iota(0);
      size_t tb = temp_bytes;
      GNNRAG_HIP(rocprim::radix_sort_pairs(temp, tb, (const uint32_t*)dst, keys_sorted, iota,
                                           out->perm[d], (size_t)F, 0u, bits, stream, false));
    }
    hipLaunchKernelGGL(k_csr_row_ptr, dim3(nb_rows), dim3(256), 0, stream, keys_sorted, F, BN,
                       out->row_ptr[d]);
    GNNRAG_LAUNCH_CHECK();
    GNNRAG_RC(hub_lists(out, d, (int32_t*)(base + L.hub_qcnt), stream));
    if (F > 0 && R1 > kHubSortMinR1) {
      // hub rows in relation order (see hub_sort_scratch): fact ids of the hub rows re-sorted by relation, stable
      const int nb = (int)((F + 255) / 256);
      uint32_t* key_in = (uint32_t*)(hub_base + H.key_in);
      uint32_t* key_out = (uint32_t*)(hub_base + H.key_out);
      int32_t* perm2 = (int32_t*)(hub_base + H.perm2);
      int32_t* seg_begin = (int32_t*)(hub_base + H.seg);
      int32_t* seg_end = seg_begin + L.heavy_cap;
      hipLaunchKernelGGL(k_csr_hub_segments, dim3((L.heavy_cap + 255) / 256), dim3(256), 0, stream, out->row_ptr[d],
                         out->heavy[d], out->n_heavy + d, L.heavy_cap, seg_begin, seg_end);
      GNNRAG_LAUNCH_CHECK();
      const unsigned rb = key_bits((size_t)R1), hb = hub_key_bits(R1, L.heavy_cap);
      size_t stb = H.total - H.temp;
      if (hb <= 32) {
        hipLaunchKernelGGL(k_csr_segkey, dim3(nb), dim3(256), 0, stream, out->perm[d], rels, F, seg_begin, seg_end,
                           out->n_heavy + d, L.heavy_cap, rb, key_in);
        GNNRAG_LAUNCH_CHECK();
        GNNRAG_HIP(rocprim::radix_sort_pairs(hub_base + H.temp, stb, (const uint32_t*)key_in, key_out,
                                             (const int32_t*)out->perm[d], perm2, (size_t)F, 0u, hb, stream, false));
      } else {
        hipLaunchKernelGGL(k_csr_relkey, dim3(nb), dim3(256), 0, stream, out->perm[d], rels, F, key_in);
        GNNRAG_LAUNCH_CHECK();
        GNNRAG_HIP(hipMemcpyAsync(perm2, out->perm[d], (size_t)F * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        GNNRAG_HIP(rocprim::segmented_radix_sort_pairs(hub_base + H.temp, stb, (const uint32_t*)key_in, key_out,
                                                       (const int32_t*)out->perm[d], perm2, (unsigned)F,
                                                       (unsigned)L.heavy_cap, (const int32_t*)seg_begin,
                                                       (const int32_t*)seg_end, 0u, rb, stream, false));
      }
      GNNRAG_HIP(hipMemcpyAsync(out->perm[d], perm2, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    }
    if (F > 0) {
      const int nb = (int)((F + 255) / 256);
      hipLaunchKernelGGL(k_csr_fill, dim3(nb), dim3(256), 0, stream, out->perm[d], src, rels, w_gnn,
                         w_rel, F, g2l, (int64_t)B * R1, N, R1, (int2*)out->edge[d], (int2*)out->edge_l[d],
                         out->w_gnn[d], out->w_rel[d]);
      GNNRAG_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(k_csr_heavy_chunks, dim3(2), dim3(1024), 0, stream, out->row_ptr[0], out->row_ptr[1],
                     out->heavy[0], out->heavy[1], out->n_heavy, out->heavy_cap, out->chunk_off[0],
                     out->chunk_off[1], out->n_chunks);
  GNNRAG_LAUNCH_CHECK();
  if (F > 0) {
    hipLaunchKernelGGL(k_csr_merge, dim3((unsigned)((F + 255) / 256), 2), dim3(256), 0, stream, out->perm[0],
                       out->perm[1], heads, tails, out->row_ptr[0], out->row_ptr[1], (const int2*)out->edge_l[0],
                       (const int2*)out->edge_l[1], out->rel_off, N, F, BN, (int2*)out->edge_m, out->m_from,
                       out->m_dst);
    GNNRAG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_csr_big, dim3((int)((BN + 255) / 256)), dim3(256), 0, stream, out->row_ptr[0],
                     out->row_ptr[1], BN, N, (int32_t)kBigDeg, out->big_cnt, out->big_nodes);
  GNNRAG_LAUNCH_CHECK();
  // the relation counts size the fused path's tables and launches.  A caller that knows them (the per-question counts of
  // a fact cache: sum and maximum) passes them in and the build does NOT wait for its stream; the device-side copies and
  // the validation bits stay behind the structure for gnnrag_csr_status.
  if (rel_total >= 0 && rel_max >= 0) {
    out->rel_total = rel_total;
    out->rel_max = rel_max;
    return 0;
  }
  int32_t stats[3] = {0, 0, 0};     // rel_total, rel_max, validation error bits
  GNNRAG_HIP(hipMemcpyAsync(stats, rel_stats, sizeof(stats), hipMemcpyDeviceToHost, stream));
  GNNRAG_HIP(hipStreamSynchronize(stream));
  if (stats[2]) {                   // out-of-range ids or a fact that connects two questions
    out->rel_total = -1;            // the structure must not be used
    return GNNRAG_E_TUPLE;
  }
  out->rel_total = stats[0];
  out->rel_max = stats[1];
  return 0;
}

// Pointers of a gnnrag_csr inside the caller's memory block (shared by gnnrag_csr_build and gnnrag_csr_concat).
static void csr_bind(gnnrag_csr* out, char* base, const CsrLayout& L, int64_t F, int32_t B, int32_t N, int32_t R1,
                     bool has_w_gnn, bool has_w_rel) {
  memset(out, 0, sizeof(*out));
  out->B = B; out->N = N; out->R1 = R1; out->F = F;
  out->heavy_deg = kHeavyDeg;
  out->heavy_cap = L.heavy_cap;
  for (int d = 0; d < 2; ++d) {
    out->row_ptr[d] = (int32_t*)(base + L.row_ptr[d]);
    out->edge[d] = (int32_t*)(base + L.edge[d]);
    out->perm[d] = (int32_t*)(base + L.perm[d]);
    out->w_gnn[d] = has_w_gnn ? (float*)(base + L.w_gnn[d]) : nullptr;
    out->w_rel[d] = has_w_rel ? (float*)(base + L.w_rel[d]) : nullptr;
    out->heavy[d] = (int32_t*)(base + L.heavy[d]);
    out->chunk_off[d] = (int32_t*)(base + L.chunk_off[d]);
    out->edge_l[d] = (int32_t*)(base + L.edge_l[d]);
  }
  out->rel_off = (int32_t*)(base + L.rel_off);
  out->edge_m = (int32_t*)(base + L.edge_m);
  out->m_from = (int32_t*)(base + L.m_from);
  out->m_dst = (int32_t*)(base + L.m_dst);
  for (int d = 0; d < 2; ++d) {
    out->hub_q_off[d] = (int32_t*)(base + L.hub_q_off[d]);
    out->hub_wbase[d] = (int32_t*)(base + L.hub_wbase[d]);
  }
  out->rel_rows = (int32_t*)(base + L.rel_rows);
  out->n_heavy = (int32_t*)(base + L.n_heavy);
  out->n_chunks = out->n_heavy + 2;
  out->big_cnt = (int32_t*)(base + L.big_cnt);
  out->big_nodes = (int32_t*)(base + L.big_nodes);
  out->big_deg = kBigDeg;
  out->max_chunks = 2 * L.heavy_cap;   // sum ceil(deg/256) over rows with deg > 256 < F/256 + F/257
}

extern "C" int gnnrag_csr_concat(const gnnrag_csr* const* parts, int32_t B, int32_t N, int32_t R1, void* csr_mem,
                                 size_t csr_bytes, gnnrag_csr* out, gnnrag_stream_t stream_) {
  if (!parts || !out || !csr_mem || B <= 0 || N <= 0 || R1 <= 0) return GNNRAG_E_BADARG;
  int64_t F = 0, RT = 0;
  int32_t rmax = 0;
  for (int b = 0; b < B; ++b) {
    const gnnrag_csr* p = parts[b];
    if (!p || p->B != 1 || p->N != N || p->R1 != R1 || p->rel_total < 0 || p->F < 0 || !p->edge_m) return GNNRAG_E_BADARG;
    F += p->F;
    RT += p->rel_total;
    rmax = p->rel_total > rmax ? p->rel_total : rmax;
  }
  const int64_t BN = (int64_t)B * N;
  if (BN >= ((int64_t)1 << 31) || F >= ((int64_t)1 << 31) || RT >= ((int64_t)1 << 31)) return GNNRAG_E_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const CsrLayout L = csr_layout(F, B, N, R1, false, false);
  if (csr_bytes < L.total) return GNNRAG_E_WORKSPACE;
  csr_bind(out, (char*)csr_mem, L, F, B, N, R1, false, false);
  out->hub_sorted = 1;
  for (int b = 0; b < B; ++b) out->hub_sorted &= parts[b]->hub_sorted;
  GNNRAG_HIP(hipMemsetAsync(out->big_cnt, 0, (size_t)B * sizeof(int32_t), stream));
  GNNRAG_HIP(hipMemsetAsync(out->n_heavy, 0, 8 * sizeof(int32_t), stream));
  ConcatArgs a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.F = F;
  for (int d = 0; d < 2; ++d) {
    a.row_ptr[d] = out->row_ptr[d];
    a.edge[d] = (int2*)out->edge[d];
    a.edge_l[d] = (int2*)out->edge_l[d];
    a.perm[d] = out->perm[d];
  }
  a.edge_m = (int2*)out->edge_m; a.m_from = out->m_from; a.m_dst = out->m_dst; a.rel_rows = (int2*)out->rel_rows; a.rel_off = out->rel_off;
  int64_t foff = 0;
  int32_t roff = 0;
  for (int b0 = 0; b0 < B; b0 += kConcatChunk) {
    const int n = B - b0 < kConcatChunk ? B - b0 : kConcatChunk;
    int64_t fmax = N;
    for (int k = 0; k < n; ++k) {
      const gnnrag_csr* p = parts[b0 + k];
      ConcatPart& q = a.part[k];
      for (int d = 0; d < 2; ++d) {
        q.row_ptr[d] = p->row_ptr[d];
        q.edge[d] = (const int2*)p->edge[d];
        q.edge_l[d] = (const int2*)p->edge_l[d];
        q.perm[d] = p->perm[d];
      }
      q.edge_m = (const int2*)p->edge_m; q.m_from = p->m_from; q.m_dst = p->m_dst; q.rel_rows = (const int2*)p->rel_rows;
      q.foff = foff; q.roff = roff; q.Fq = (int32_t)p->F; q.Rq = p->rel_total; q.b = b0 + k;
      foff += p->F;
      roff += p->rel_total;
      fmax = 2 * p->F > fmax ? 2 * p->F : fmax;
    }
    a.n = n;
    int gx = (int)((fmax + 255) / 256);
    gx = gx < 1 ? 1 : gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(k_csr_concat, dim3(gx, n, 4), dim3(256), 0, stream, a);
    GNNRAG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_csr_concat_tail, dim3(1), dim3(1), 0, stream, out->row_ptr[0], out->row_ptr[1], out->rel_off, BN, B,
                     (int32_t)F, (int32_t)RT);
  GNNRAG_LAUNCH_CHECK();
  for (int d = 0; d < 2; ++d) {
    GNNRAG_RC(hub_lists(out, d, (int32_t*)((char*)csr_mem + L.hub_qcnt), stream));
  }
  hipLaunchKernelGGL(k_csr_heavy_chunks, dim3(2), dim3(1024), 0, stream, out->row_ptr[0], out->row_ptr[1],
                     out->heavy[0], out->heavy[1], out->n_heavy, out->heavy_cap, out->chunk_off[0],
                     out->chunk_off[1], out->n_chunks);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_csr_big, dim3((int)((BN + 255) / 256)), dim3(256), 0, stream, out->row_ptr[0],
                     out->row_ptr[1], BN, N, (int32_t)kBigDeg, out->big_cnt, out->big_nodes);
  GNNRAG_LAUNCH_CHECK();
  out->rel_total = (int32_t)RT;      // known on the host: this call does not wait for the stream
  out->rel_max = rmax;
  return 0;
}
