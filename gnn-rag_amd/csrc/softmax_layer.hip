// Per-question masked softmax, the one-call layer driver and small utilities.
#include "gnnrag_common.h"
#include "dense_internal.h"

#include <map>
#include <vector>

#ifndef GNNRAG_OVERLAP_TABLES_DEFAULT
#define GNNRAG_OVERLAP_TABLES_DEFAULT 0
#endif
#ifndef GNNRAG_OVERLAP_PROJ_DEFAULT
#define GNNRAG_OVERLAP_PROJ_DEFAULT 0
#endif

namespace gnnrag {

// dist[g,:] = softmax(score[g,:])  (reasongnn.py:169).  One 1024-thread workgroup per question.
// Masked slots hold exactly -1e11 (fp32), so exp(-1e11 - max) == 0 exactly; a question whose
// slots are all masked gets exactly 1/N everywhere, like the reference.
// ITEMS > 0: the N <= 1024*ITEMS scores are read ONCE into registers (the kernel is pure latency:
// 64 workgroups on a 256-CU chip); ITEMS == 0: any N, three passes over L1/L2-resident data.
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_masked_softmax(const float* __restrict__ score,
                                                         float* __restrict__ dist, int N) {
  __shared__ float red[16];
  __shared__ float bcast;
  const float* s = score + (size_t)blockIdx.x * N;
  float* o = dist + (size_t)blockIdx.x * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[ITEMS > 0 ? ITEMS : 1];

  float m = -INFINITY;
  if constexpr (ITEMS > 0) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int i = threadIdx.x + k * 1024;
      v[k] = i < N ? s[i] : -INFINITY;
      m = fmaxf(m, v[k]);
    }
  } else {
    for (int i = threadIdx.x; i < N; i += 1024) m = fmaxf(m, s[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = red[0];
    for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
    bcast = mm;
  }
  __syncthreads();
  m = bcast;
  __syncthreads();

  float sum = 0.f;
  if constexpr (ITEMS > 0) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      v[k] = expf(v[k] - m);            // exp(-inf) = 0 for the padding lanes
      sum += v[k];
    }
  } else {
    for (int i = threadIdx.x; i < N; i += 1024) sum += expf(s[i] - m);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    bcast = t;
  }
  __syncthreads();
  const float total = bcast;
  if constexpr (ITEMS > 0) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int i = threadIdx.x + k * 1024;
      if (i < N) o[i] = v[k] / total;
    }
  } else {
    for (int i = threadIdx.x; i < N; i += 1024) o[i] = expf(s[i] - m) / total;
  }
}

// The softmax above AND the next layer's fact priors in one launch: layer l + 1's dense-prior walk starts by turning
// dist_l into one (prior, relation) pair per fact and direction (k_fact_prior_merged: a pass over 2F records that gathers
// dist[src], 7.8 us at C2) - here the workgroups that just normalised a question's scores keep its distribution in LDS
// and write the pairs of the question's own facts (a contiguous range of the merged stream), so the walk finds them
// ready.  Workgroup (g, part): every part repeats the question's softmax (8 KB of scores; same arithmetic, same
// reduction order as k_masked_softmax<ITEMS>, so the distribution is bit-identical), part 0 writes it, and the parts
// share the question's facts.  GNNRAG_SOFTMAX_PAIRS=0 keeps the two launches apart (diagnostic).
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_softmax_pairs(const float* __restrict__ score, float* __restrict__ dist, int N,
                                                        const int32_t* __restrict__ rp0, const int32_t* __restrict__ rp1,
                                                        const int2* __restrict__ em, const int32_t* __restrict__ from,
                                                        const float* __restrict__ w0, const float* __restrict__ w1,
                                                        int64_t F, int2* __restrict__ pr) {
  extern __shared__ float s_dist[];        // [N]
  __shared__ float red[16];
  __shared__ float bcast;
  const int g = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  const float* s = score + (size_t)g * N;
  float* o = dist + (size_t)g * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[ITEMS];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = threadIdx.x + k * 1024;
    v[k] = i < N ? s[i] : -INFINITY;
    m = fmaxf(m, v[k]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = red[0];
    for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
    bcast = mm;
  }
  __syncthreads();
  m = bcast;
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    v[k] = expf(v[k] - m);
    sum += v[k];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    bcast = t;
  }
  __syncthreads();
  const float total = bcast;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = threadIdx.x + k * 1024;
    if (i < N) {
      const float d = v[k] / total;
      s_dist[i] = d;
      if (part == 0) o[i] = d;
    }
  }
  __syncthreads();
  // the question's facts: merged positions [rp0[gN] + rp1[gN], rp0[(g+1)N] + rp1[(g+1)N])
  const int64_t n0 = (int64_t)g * N;
  const int64_t beg = (int64_t)rp0[n0] + rp1[n0], end = (int64_t)rp0[n0 + N] + rp1[n0 + N];
  const int64_t lo = beg + (end - beg) * part / nparts, hi = beg + (end - beg) * (part + 1) / nparts;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  for (int64_t j = lo + threadIdx.x; j < hi; j += 1024) {
    const i32x2 e = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(em) + j);
    // a fact stays inside its question; a tuple that was never validated (gnnrag_csr_build_counts defers the check to
    // gnnrag_csr_status) must still read a DEFINED value: a source outside the question contributes prior 0
    const unsigned s = (unsigned)(e.x - (int)n0);
    float p = s < (unsigned)N ? s_dist[s] : 0.f;
    if (w0) {
      const int f = from[j];
      p *= f < F ? w0[f] : w1[f - F];
    }
    pr[j] = make_int2(__float_as_int(p), e.y);
  }
}

// float4 copy that measures the streaming ceiling of the box (bench.py's `measured_copy_ceiling_GBps`): four 16-byte loads
// per thread in flight before the first store, every block iteration moves 16 KB of contiguous data (round 3's one
// load -> one store loop reached 4.95 TB/s where the guide's float4 copy reaches 6.29)
__global__ __launch_bounds__(256) void k_stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                     int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < n4; base += stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) v[u] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) __builtin_nontemporal_store(v[u], dst + i);
    }
  }
}

struct LayerWs { size_t T_fwd, T_inv, planes, agg, P, nbr, partial, partial_bytes, fws, fws_bytes, total; };

// flops of the dense part per layer call: unfused = one [BN,(2I+1)D]x[(2I+1)D,D] GEMM; fused = the
// per-question relation tables [2*rel_total, I*D]x[I*D, D] (rel_total = sum over questions of the
// relations each one uses) plus the self block [BN,D]x[D,D]
static bool fused_is_cheaper(int64_t B, int64_t N, int64_t rel_total, int64_t D, int64_t I) {
  const double unfused = (double)B * N * (2 * I + 1) * D * D;
  const double fused = 2.0 * rel_total * I * D * D + (double)B * N * D * D;
  return fused < 0.8 * unfused;
}

static bool path_ok(int32_t path) {
  const int base = path & 0xf, flags = path & ~0xf & ~GNNRAG_PATH_SEED_PRIOR & ~GNNRAG_PATH_REUSE_PROJ;
  if (base < GNNRAG_PATH_AUTO || base > GNNRAG_PATH_FUSED) return false;
  return flags == 0 || flags == GNNRAG_PATH_ONLY_FWD || flags == GNNRAG_PATH_ONLY_INV;
}

static LayerWs layer_ws(const gnnrag_csr* csr, int32_t D, int32_t I) {
  LayerWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t BN = (size_t)csr->B * csr->N;
  w.T_fwd = take((size_t)2 * csr->R1 * D * sizeof(float));          // [2][R1][D]: forward, inverse
  w.T_inv = w.T_fwd + (size_t)csr->R1 * D * sizeof(float);
  // bf16 planes of relu(+-T) for the V form of the relation tables (one layer), where that kernel's shapes apply
  w.planes = take(tables_vq_shape_ok(D, I) ? tables_vq_planes_bytes(csr->R1) : 0);
  // the two paths never run in the same call: their big buffers share one region
  const size_t a_bytes = BN * 2 * I * D * sizeof(float);
  const size_t p_bytes = align_up((size_t)2 * (csr->rel_total > 0 ? csr->rel_total : 1) * D * sizeof(float), 256);
  const size_t n_bytes = (BN + 1) * D * sizeof(float);              // + the zero row frontier layers read for rows off the frontier
  const size_t big = take(a_bytes > p_bytes + n_bytes ? a_bytes : p_bytes + n_bytes);
  w.agg = big;
  w.P = big;
  w.nbr = big + p_bytes;
  w.partial_bytes = gnnrag_aggregate_workspace_bytes(csr, D, I);
  w.partial = take(w.partial_bytes);
  w.fws_bytes = gnnrag_frontier_workspace_bytes(csr);
  w.fws = take(w.fws_bytes);
  w.total = off;
  return w;
}

// ---- the stack driver's side stream (relation tables of layers 1.. under the layers in front of them) ---------------
// P of layer j >= 1 depends on the relation planes, the instructions and e2e_linear{j}.weight only - on nothing the
// layers 0 .. j-1 compute.  With GNNRAG_OVERLAP_TABLES (default: see overlap_enabled) the whole-iteration call forks a
// side stream behind its relation projections, runs the L - 1 table launches there into L - 1 table buffers of their
// own, and the caller's stream waits for layer j's tables right before layer j's walk: the MFMA-bound table kernel runs
// beside the issue-bound walk / the latency-bound frontier launches of the layers in front of it.  Same kernels'
// arithmetic in the same order per output element: results are bit-identical to the serial sequence.
// Stream and events are cached per host thread and device (an event shared by two host threads could hand one thread's
// wait the other thread's record).
struct OverlapRes {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr;
  std::vector<hipEvent_t> done;
};

static int overlap_res(int n_done, OverlapRes** out) {
  static thread_local std::map<int, OverlapRes> per_device;
  int dev = 0;
  GNNRAG_HIP(hipGetDevice(&dev));
  OverlapRes& r = per_device[dev];
  if (!r.side) {
    GNNRAG_HIP(hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking));
    GNNRAG_HIP(hipEventCreateWithFlags(&r.fork, hipEventDisableTiming));
  }
  while ((int)r.done.size() < n_done) {
    hipEvent_t e = nullptr;
    GNNRAG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    r.done.push_back(e);
  }
  *out = &r;
  return 0;
}

static bool overlap_enabled() {
  const char* env = getenv("GNNRAG_OVERLAP_TABLES");      // read per call: tests compare both forms in one process
  return env ? env[0] != '0' : GNNRAG_OVERLAP_TABLES_DEFAULT != 0;
}

// shapes for which the side-stream form exists: the V-form table kernel's (planes) and the LDS walk behind it
static bool overlap_shape_ok(const gnnrag_csr* csr, int32_t L, int32_t D, int32_t I) {
  return L > 1 && tables_vq_shape_ok(D, I) && csr->rel_total >= 1024 &&
         gnnrag_aggregate_fused_variant(csr, D) != GNNRAG_WALK_L2_GATHER;
}

static size_t overlap_p_bytes(const gnnrag_csr* csr, int32_t D) {
  return align_up((size_t)2 * (csr->rel_total > 0 ? csr->rel_total : 1) * D * sizeof(float), 256);
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_masked_softmax(const float* score, float* dist, int32_t B, int32_t N,
                                     gnnrag_stream_t stream) {
  if (!score || !dist || B <= 0 || N <= 0) return GNNRAG_E_BADARG;
  if (N <= 2048) hipLaunchKernelGGL(k_masked_softmax<2>, dim3(B), dim3(1024), 0, (hipStream_t)stream, score, dist, N);
  else if (N <= 8192) hipLaunchKernelGGL(k_masked_softmax<8>, dim3(B), dim3(1024), 0, (hipStream_t)stream, score, dist, N);
  else hipLaunchKernelGGL(k_masked_softmax<0>, dim3(B), dim3(1024), 0, (hipStream_t)stream, score, dist, N);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_stream_copy(const float* src, float* dst, int64_t n, gnnrag_stream_t stream) {
  if (!src || !dst || n < 0 || (n & 3)) return GNNRAG_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_stream_copy, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src,
                     (f32x4*)dst, n / 4);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gnnrag_layer_workspace_bytes(const gnnrag_csr* csr, int32_t D, int32_t I) {
  if (!csr || D <= 0 || I <= 0) return 0;
  return layer_ws(csr, D, I).total;
}

// the layer's last launch: softmax of the scores - and, when the caller says the NEXT layer's dense-prior LDS walk will
// read this distribution (want_pairs), that walk's (prior, relation) pairs with it (k_softmax_pairs).  *pairs_done tells
// whether they were written.
static int finish_softmax(const gnnrag_csr* csr, const LayerWs& w, char* base, const float* score, float* dist_out, int32_t D,
                          bool want_pairs, bool* pairs_done, gnnrag_stream_t stream) {
  if (pairs_done) *pairs_done = false;
  const char* env = getenv("GNNRAG_SOFTMAX_PAIRS");      // read per call: the test compares both forms in one process
  const bool enabled = !(env && env[0] == '0');
  if (want_pairs && pairs_done && enabled && csr->N <= 8192 && csr->edge_m) {
    PriorPairsTarget t;
    const int rc = prior_pairs_target(csr, D, base + w.partial, w.partial_bytes, &t);
    if (rc) return rc;
    if (t.ok) {
      // about two workgroups per CU over the batch; a question's facts are shared by at most 8
      int parts = (512 + csr->B - 1) / csr->B;
      parts = parts < 1 ? 1 : parts > 8 ? 8 : parts;
      const size_t lds = (size_t)csr->N * sizeof(float);
      if (csr->N <= 2048)
        hipLaunchKernelGGL(k_softmax_pairs<2>, dim3(csr->B, parts), dim3(1024), lds, (hipStream_t)stream, score, dist_out,
                           csr->N, csr->row_ptr[0], csr->row_ptr[1], t.edge_m, t.m_from, t.w0, t.w1, t.F, t.pairs);
      else
        hipLaunchKernelGGL(k_softmax_pairs<8>, dim3(csr->B, parts), dim3(1024), lds, (hipStream_t)stream, score, dist_out,
                           csr->N, csr->row_ptr[0], csr->row_ptr[1], t.edge_m, t.m_from, t.w0, t.w1, t.F, t.pairs);
      GNNRAG_LAUNCH_CHECK();
      *pairs_done = true;
      return 0;
    }
  }
  return gnnrag_masked_softmax(score, dist_out, csr->B, csr->N, stream);
}

// one layer behind its relation projections T_fwd / T_inv (already computed)
static int layer_body(const gnnrag_csr* csr, const LayerWs& w, char* base, const float* h, const float* dist,
                      const float* ins, const float* T_fwd, const float* T_inv, const void* planes, const float* W_e2e,
                      const float* b_e2e, const float* w_score, const float* b_score, const float* mask,
                      float* h_out, float* score_out, float* dist_out, int32_t D, int32_t I, int32_t path,
                      int32_t math, gnnrag_stream_t stream, bool pairs_ready = false, bool* pairs_for_next = nullptr,
                      float* P_done = nullptr, hipEvent_t t_ready = nullptr) {
  // t_ready != nullptr: the relation projections (T, planes) are being computed on the stack driver's side stream; the
  // caller's stream waits for them in front of the first launch that reads them - behind the frontier build in the
  // seed-prior form (which reads only the prior and the structure), at the top otherwise
  // P_done != nullptr: this layer's relation tables were already computed into P_done (and its score buffer zeroed) by
  // the stack driver's side stream, and the caller's stream has waited for them - the table launch is skipped
  // pairs_ready: the previous layer's softmax launch left this layer's (prior, relation) pairs in the workspace;
  // pairs_for_next != nullptr: the caller will run another layer on dist_out - *pairs_for_next reports whether this
  // layer's last launch wrote that layer's pairs
  if (pairs_for_next) *pairs_for_next = false;
  const int64_t BN = (int64_t)csr->B * csr->N;
  int rc;
  // one-direction layers (NSM): only that direction's tables are built and walked where the kernels at hand can
  // leave a direction out (V-form tables + LDS walk); otherwise both directions run and the caller's zero weight
  // block makes the other one contribute exactly 0
  const int only = (path & GNNRAG_PATH_ONLY_FWD) ? 0 : (path & GNNRAG_PATH_ONLY_INV) ? 1 : -1;
  const bool seed_prior = (path & GNNRAG_PATH_SEED_PRIOR) != 0;
  path &= 0xf;
  if (path == GNNRAG_PATH_AUTO)
    path = fused_is_cheaper(csr->B, csr->N, csr->rel_total, D, I) ? GNNRAG_PATH_FUSED : GNNRAG_PATH_UNFUSED;
  if (path == GNNRAG_PATH_FUSED) {
    float* P = P_done ? P_done : (float*)(base + w.P);
    float* nbr = (float*)(base + w.nbr);
    const bool one_dir = only >= 0 && gnnrag_aggregate_fused_variant(csr, D) != GNNRAG_WALK_L2_GATHER;
    // the frontier kernels issue 16-byte loads / stores on ins, W, T, P (workspace) and the node rows: a misaligned view
    // takes the regular fused path below, whose launchers check alignment themselves
    const bool fr_aligned = ((((uintptr_t)ins | (uintptr_t)W_e2e | (uintptr_t)T_fwd | (uintptr_t)T_inv | (uintptr_t)P |
                               (uintptr_t)nbr | (uintptr_t)dist) & 15) == 0) && csr->N > 0;
    const bool frontier_form = seed_prior && only < 0 && fr_aligned && gnnrag_frontier_supported(csr, D) && csr->rel_total > 0;
    if (t_ready && !frontier_form) GNNRAG_HIP(hipStreamWaitEvent((hipStream_t)stream, t_ready, 0));
    if (frontier_form) {
      // The caller says `dist` is a seed distribution (first layer of a ReaRev iteration, rearev.py:208): only the
      // seeds' facts have a prior, so only the relation-table rows those facts use and the neighbour sums of the nodes
      // they reach are computed (frontier.hip; the frontier itself is derived from `dist` on the device, so a prior
      // that is NOT sparse still gives the right result, slowly); the update reads `nbr` through the row gates.
      void* fws = base + w.fws;
      const bool gated = update_rows_supported(h, nbr, W_e2e, h_out, BN, D, I, math);
      if (!gated) GNNRAG_HIP(hipMemsetAsync(nbr, 0, (size_t)BN * D * sizeof(float), (hipStream_t)stream));
      rc = frontier_build_z(csr, dist, fws, w.fws_bytes, score_out, BN, nbr + (size_t)BN * D, D, (hipStream_t)stream);
      if (t_ready) GNNRAG_HIP(hipStreamWaitEvent((hipStream_t)stream, t_ready, 0));     // (also joins after an error)
      if (rc) return rc;
      rc = gnnrag_relation_tables_frontier(csr, fws, T_fwd, T_inv, ins, W_e2e, P, D, I, stream);
      if (rc) return rc;
      rc = gnnrag_aggregate_fused_frontier(csr, fws, dist, P, nbr, D, stream);
      if (rc) return rc;
      rc = update_score_fused_rows(h, nbr, gated ? frontier_row_flags(csr, fws) : nullptr, W_e2e, b_e2e, w_score, b_score,
                                   mask, h_out, score_out, BN, D, I, math, (hipStream_t)stream, true);
      if (rc) return rc;
      return finish_softmax(csr, w, base, score_out, dist_out, D, only < 0, pairs_for_next, stream);
    }
    rc = GNNRAG_E_UNSUPPORTED;
    bool score_zeroed = false;    // the V-form table kernel also zeroes the score the update accumulates onto
    if (P_done) {
      rc = 0;
      score_zeroed = true;
    } else if (planes && math != GNNRAG_MATH_FP32 && csr->rel_total > 0) {
      const char* menv = getenv("GNNRAG_TABLES_LITE_MAIN");      // experiment knob: the side-stream kernel in the serial sequence
      rc = ((menv && menv[0] == '1') ? tables_vq_lite_launch_z : tables_vq_launch_z)(
          csr, planes, ins, W_e2e, P, D, I, one_dir ? only : -1, score_out, BN, (hipStream_t)stream);
      score_zeroed = rc == 0;
    }
    if (rc == GNNRAG_E_UNSUPPORTED) rc = gnnrag_relation_tables(csr, T_fwd, T_inv, ins, W_e2e, P, D, I, math, stream);
    if (rc) return rc;
    rc = aggregate_fused_dirs(csr, dist, P, nbr, D, one_dir ? 2 - only : 0, base + w.partial, w.partial_bytes,
                              (hipStream_t)stream, pairs_ready && !one_dir);
    if (rc) return rc;
    rc = update_score_fused_z(h, nbr, W_e2e, b_e2e, w_score, b_score, mask, h_out, score_out, BN, D, I, math,
                              (hipStream_t)stream, score_zeroed);
    if (rc) return rc;
    return finish_softmax(csr, w, base, score_out, dist_out, D, only < 0, pairs_for_next, stream);
  } else {
    if (t_ready) GNNRAG_HIP(hipStreamWaitEvent((hipStream_t)stream, t_ready, 0));
    float* agg = (float*)(base + w.agg);
    rc = gnnrag_aggregate(csr, dist, ins, T_fwd, T_inv, agg, D, I, base + w.partial, w.partial_bytes, stream);
    if (rc) return rc;
    rc = gnnrag_update_score(h, agg, W_e2e, b_e2e, w_score, b_score, mask, h_out, score_out, BN, D, I, math,
                             stream);
    if (rc) return rc;
  }
  return gnnrag_masked_softmax(score_out, dist_out, csr->B, csr->N, stream);
}

// T[j][d] = rel_linear{j}(rel_features_d) (+ pos_emb{j}_d) for n layers into T [n][2][R1][D]: once per relation
// row, not once per fact; all layers in one launch when the float4 kernel applies (rel_transform.hip)
static int rel_projections(const gnnrag_csr* csr, int32_t n, const gnnrag_layer_params* layers,
                           const float* relfeat_fwd, const float* relfeat_inv, int32_t pos_rows, float* T,
                           void* planes, bool* planes_written, int32_t D, int32_t math, gnnrag_stream_t stream) {
  *planes_written = false;
  if ((D & 3) == 0) {
    const int rc = gnnrag_rel_transform(relfeat_fwd, relfeat_inv, csr->R1, D, n, layers, pos_rows, T, planes, stream);
    if (rc == 0) *planes_written = planes != nullptr;
    // unaligned operands: the k-tiled kernel below (its scalar loaders take any address); the planes stay unwritten,
    // which the caller learns through *planes_written and then builds the tables from T (gnnrag_relation_tables)
    if (rc != GNNRAG_E_UNSUPPORTED) return rc;
  }
  const size_t RD = (size_t)csr->R1 * D;
  for (int j = 0; j < n; ++j) {
    const gnnrag_layer_params& p = layers[j];
    const int rc = gnnrag_linear_pair(relfeat_fwd, relfeat_inv, csr->R1, D, p.W_rel, p.b_rel, p.pos_fwd, p.pos_inv,
                                      p.pos_fwd ? pos_rows : 0, T + 2 * j * RD, T + (2 * j + 1) * RD, D, math, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" size_t gnnrag_stack_workspace_bytes(const gnnrag_csr* csr, int32_t L, int32_t D, int32_t I) {
  if (!csr || L <= 0 || D <= 0 || I <= 0) return 0;
  // one layer's workspace + the relation projections of all L layers (one contiguous block, computed up front)
  // + their bf16 planes where the V-form tables kernel applies
  // + L - 1 relation-table buffers where the side-stream form applies (layer j's tables are computed while layer j - 1
  // still reads its own)
  return layer_ws(csr, D, I).total + align_up((size_t)L * 2 * csr->R1 * D * sizeof(float), 256) +
         align_up(tables_vq_shape_ok(D, I) ? (size_t)L * tables_vq_planes_bytes(csr->R1) : 0, 256) +
         (overlap_shape_ok(csr, L, D, I) ? (size_t)(L - 1) * overlap_p_bytes(csr, D) : 0);
}

extern "C" int gnnrag_reason_layer(const gnnrag_csr* csr, const float* h, const float* dist, const float* ins,
                                   const float* relfeat_fwd, const float* relfeat_inv, const float* W_rel,
                                   const float* b_rel, const float* pos_fwd, const float* pos_inv,
                                   int32_t pos_rows, const float* W_e2e, const float* b_e2e,
                                   const float* w_score, const float* b_score, const float* mask,
                                   float* h_out, float* score_out, float* dist_out, void* workspace,
                                   size_t workspace_bytes, int32_t D, int32_t I, int32_t path, int32_t math,
                                   gnnrag_stream_t stream) {
  if (!csr || !h || !dist || !ins || !relfeat_fwd || !relfeat_inv || !W_rel || !b_rel || !W_e2e || !b_e2e ||
      !w_score || !b_score || !mask || !h_out || !score_out || !dist_out || !workspace || D <= 0 || I <= 0)
    return GNNRAG_E_BADARG;
  if (!path_ok(path)) return GNNRAG_E_BADARG;
  const LayerWs w = layer_ws(csr, D, I);
  if (workspace_bytes < w.total) return GNNRAG_E_WORKSPACE;
  char* base = (char*)workspace;
  float* T_fwd = (float*)(base + w.T_fwd);
  float* T_inv = (float*)(base + w.T_inv);
  const gnnrag_layer_params p = {W_rel, b_rel, pos_fwd, pos_inv, W_e2e, b_e2e};
  // the planes are only worth writing when the fused path with a bf16x3 product will read them
  const bool fused = (path & 0xf) == GNNRAG_PATH_FUSED ||
                     ((path & 0xf) == GNNRAG_PATH_AUTO && fused_is_cheaper(csr->B, csr->N, csr->rel_total, D, I));
  void* planes = fused && math != GNNRAG_MATH_FP32 && tables_vq_shape_ok(D, I) && csr->rel_total >= 1024
                     ? (void*)(base + w.planes) : nullptr;
  bool planes_written = false;
  const int rc = rel_projections(csr, 1, &p, relfeat_fwd, relfeat_inv, pos_rows, T_fwd, planes, &planes_written, D, math,
                                 stream);
  if (rc) return rc;
  if (!planes_written) planes = nullptr;
  return layer_body(csr, w, base, h, dist, ins, T_fwd, T_inv, planes, W_e2e, b_e2e, w_score, b_score, mask, h_out, score_out,
                    dist_out, D, I, path, math, stream);
}

extern "C" int gnnrag_reason_stack(const gnnrag_csr* csr, int32_t L, const gnnrag_layer_params* layers,
                                   const float* h0, const float* dist0, const float* ins,
                                   const float* relfeat_fwd, const float* relfeat_inv, int32_t pos_rows,
                                   const float* w_score, const float* b_score, const float* mask, float* h_out,
                                   float* score_out, float* dist_out, void* workspace, size_t workspace_bytes,
                                   int32_t D, int32_t I, int32_t path, int32_t math, gnnrag_stream_t stream) {
  if (!csr || L <= 0 || !layers || !h0 || !dist0 || !ins || !relfeat_fwd || !relfeat_inv || !w_score || !b_score ||
      !mask || !h_out || !score_out || !dist_out || !workspace || D <= 0 || I <= 0)
    return GNNRAG_E_BADARG;
  if (!path_ok(path)) return GNNRAG_E_BADARG;
  for (int j = 0; j < L; ++j)
    if (!layers[j].W_rel || !layers[j].b_rel || !layers[j].W_e2e || !layers[j].b_e2e) return GNNRAG_E_BADARG;
  const LayerWs w = layer_ws(csr, D, I);
  if (workspace_bytes < w.total) return GNNRAG_E_WORKSPACE;
  char* base = (char*)workspace;
  const size_t BN = (size_t)csr->B * csr->N;
  const size_t RD = (size_t)csr->R1 * D;
  // with a gnnrag_stack_workspace_bytes workspace the relation projections of ALL layers are computed first, in one
  // launch (they depend on neither the node state nor the distribution), into the [L][2][R1][D] block behind the
  // layer workspace; with a layer-sized workspace each layer projects its own in front of its kernels
  const bool upfront = L > 1 && workspace_bytes >= gnnrag_stack_workspace_bytes(csr, L, D, I);
  float* T0 = (float*)(base + w.T_fwd);
  float* Tall = (float*)(base + w.total);
  const bool fused = (path & 0xf) == GNNRAG_PATH_FUSED ||
                     ((path & 0xf) == GNNRAG_PATH_AUTO && fused_is_cheaper(csr->B, csr->N, csr->rel_total, D, I));
  const bool want_planes = fused && math != GNNRAG_MATH_FP32 && tables_vq_shape_ok(D, I) && csr->rel_total >= 1024;
  const size_t plane_bytes = tables_vq_planes_bytes(csr->R1);
  char* planes_all = base + w.total + align_up((size_t)L * 2 * RD * sizeof(float), 256);
  bool planes_written = false;
  hipEvent_t t_ready = nullptr;      // set while the relation projections run on the side stream (layer 0 joins them)
  const bool reuse = upfront && (path & GNNRAG_PATH_REUSE_PROJ) != 0;
  path &= ~GNNRAG_PATH_REUSE_PROJ;
  if (reuse) {
    // the previous call on this workspace left T (and the planes, where this shape writes them) in place
    // - exactly when the projecting call's kernel accepted the SAME operands (incl. the outputs inside this workspace)
    planes_written = want_planes && rel_transform_accepts(relfeat_fwd, relfeat_inv, csr->R1, D, L, layers, pos_rows, Tall,
                                                          planes_all);
  } else if (upfront) {
    // GNNRAG_OVERLAP_PROJ (default: GNNRAG_OVERLAP_PROJ_DEFAULT): the projections on the side stream, so that layer 0's
    // frontier build - which reads only the prior and the structure - runs beside them instead of behind them
    const char* penv = getenv("GNNRAG_OVERLAP_PROJ");
    const bool proj_side = fused && (path & GNNRAG_PATH_SEED_PRIOR) && (penv ? penv[0] != '0' : GNNRAG_OVERLAP_PROJ_DEFAULT != 0);
    OverlapRes* pr = nullptr;
    if (proj_side) {
      const int rc0 = overlap_res(L, &pr);
      if (rc0) return rc0;
      GNNRAG_HIP(hipEventRecord(pr->fork, (hipStream_t)stream));
      GNNRAG_HIP(hipStreamWaitEvent(pr->side, pr->fork, 0));
    }
    const int rc = rel_projections(csr, L, layers, relfeat_fwd, relfeat_inv, pos_rows, Tall,
                                   want_planes ? planes_all : nullptr, &planes_written, D, math,
                                   proj_side ? (gnnrag_stream_t)pr->side : stream);
    if (proj_side) {
      GNNRAG_HIP(hipEventRecord(pr->done[0], pr->side));
      if (rc) (void)hipStreamWaitEvent((hipStream_t)stream, pr->done[0], 0);
      else t_ready = pr->done[0];
    }
    if (rc) return rc;
  }
  // side-stream tables (see OverlapRes): only in the full-workspace form with the planes in place, both directions
  const bool overlap = upfront && want_planes && planes_written && fused && (path & ~0xf & ~GNNRAG_PATH_SEED_PRIOR) == 0 &&
                       overlap_shape_ok(csr, L, D, I) && overlap_enabled() &&
                       ((((uintptr_t)ins | (uintptr_t)planes_all) & 15) == 0);
  OverlapRes* ov = nullptr;
  char* P_extra = planes_all + align_up(tables_vq_shape_ok(D, I) ? (size_t)L * plane_bytes : 0, 256);
  const size_t p_bytes = overlap_p_bytes(csr, D);
  bool side_ok[64] = {false};
  // gate (GNNRAG_OVERLAP_GATE, default on): layer j + 1's tables are not enqueued at the fork but right in front of layer
  // j's walk, so that they run BESIDE that walk (complementary pipes) instead of competing with layer j - 1's update for CUs
  const char* genv = getenv("GNNRAG_OVERLAP_GATE");
  const bool gate = !(genv && genv[0] == '0');
  const char* lenv = getenv("GNNRAG_TABLES_LITE");           // 0: the side stream launches k_tables_vq itself (A/B)
  const bool lite = !(lenv && lenv[0] == '0');
  auto side_tables = [&](int j) -> int {                     // layer j's tables on the side stream
    if (j >= 64 || (((uintptr_t)layers[j].W_e2e) & 15) != 0) return 0;
    const int rc = (lite ? tables_vq_lite_launch_z : tables_vq_launch_z)(
        csr, planes_all + (size_t)j * plane_bytes, ins, layers[j].W_e2e, (float*)(P_extra + (size_t)(j - 1) * p_bytes), D, I, -1,
        score_out + (size_t)j * BN, (int64_t)BN, ov->side);
    if (rc == GNNRAG_E_UNSUPPORTED) return 0;
    if (rc) return rc;
    side_ok[j] = true;
    GNNRAG_HIP(hipEventRecord(ov->done[j], ov->side));
    return 0;
  };
  auto side_join_all = [&](int from) {       // whatever happened: a forked side stream is joined again (captures!)
    if (!ov) return;
    (void)hipEventRecord(ov->done[0], ov->side);
    (void)hipStreamWaitEvent((hipStream_t)stream, ov->done[0], 0);
    (void)from;
  };
  if (overlap && L <= 64) {
    { const int rc = overlap_res(L, &ov); if (rc) return rc; }
    GNNRAG_HIP(hipEventRecord(ov->fork, (hipStream_t)stream));
    GNNRAG_HIP(hipStreamWaitEvent(ov->side, ov->fork, 0));
    int rc_side = 0;
    for (int j = 1; j < (gate ? 2 : L) && !rc_side; ++j) rc_side = side_tables(j);
    if (rc_side) {
      side_join_all(0);
      return rc_side;
    }
  }
  const float* h = h0;
  const float* dist = dist0;
  bool pairs_ready = false;      // layer j - 1's softmax launch wrote layer j's (prior, relation) pairs
  for (int j = 0; j < L; ++j) {
    const gnnrag_layer_params& p = layers[j];
    float* hj = h_out + (size_t)j * BN * D;
    float* sj = score_out + (size_t)j * BN;
    float* dj = dist_out + (size_t)j * BN;
    float* T = upfront ? Tall + (size_t)j * 2 * RD : T0;
    void* planes = !want_planes ? nullptr : upfront ? (void*)(planes_all + (size_t)j * plane_bytes) : (void*)(base + w.planes);
    if (!upfront) {
      const int rc = rel_projections(csr, 1, &p, relfeat_fwd, relfeat_inv, pos_rows, T, planes, &planes_written, D, math,
                                     stream);
      if (rc) return rc;
    }
    if (!planes_written) planes = nullptr;
    // GNNRAG_PATH_SEED_PRIOR describes dist0, i.e. layer 0 only (every later layer starts from a softmax output)
    bool pairs_next = false;
    float* P_done = nullptr;
    if (ov && j < 64 && side_ok[j]) {
      GNNRAG_HIP(hipStreamWaitEvent((hipStream_t)stream, ov->done[j], 0));     // joins the side stream up to layer j's tables
      P_done = (float*)(P_extra + (size_t)(j - 1) * p_bytes);
    }
    if (ov && gate && j >= 1 && j + 1 < L) {
      // layer j + 1's tables start where layer j's walk starts
      GNNRAG_HIP(hipEventRecord(ov->fork, (hipStream_t)stream));
      GNNRAG_HIP(hipStreamWaitEvent(ov->side, ov->fork, 0));
      const int rc_side = side_tables(j + 1);
      if (rc_side) {
        side_join_all(j);
        return rc_side;
      }
    }
    const int rc = layer_body(csr, w, base, h, dist, ins, T, T + RD, planes, p.W_e2e, p.b_e2e, w_score, b_score, mask, hj, sj,
                              dj, D, I, j == 0 ? path : (path & ~GNNRAG_PATH_SEED_PRIOR), math, stream, pairs_ready,
                              j + 1 < L ? &pairs_next : nullptr, P_done, j == 0 ? t_ready : nullptr);
    if (rc) {
      if (j == 0 && t_ready) (void)hipStreamWaitEvent((hipStream_t)stream, t_ready, 0);
      side_join_all(j);       // join what is still outstanding on the side stream before reporting the error
      return rc;
    }
    pairs_ready = pairs_next;
    h = hj;
    dist = dj;
  }
  return 0;
}

struct gnnrag_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" int gnnrag_reason_stack_capture(const gnnrag_csr* csr, int32_t L, const gnnrag_layer_params* layers,
                                           const float* h0, const float* dist0, const float* ins,
                                           const float* relfeat_fwd, const float* relfeat_inv, int32_t pos_rows,
                                           const float* w_score, const float* b_score, const float* mask,
                                           float* h_out, float* score_out, float* dist_out, void* workspace,
                                           size_t workspace_bytes, int32_t D, int32_t I, int32_t path, int32_t math,
                                           gnnrag_stream_t stream_, gnnrag_graph** out) {
  if (!out) return GNNRAG_E_BADARG;
  *out = nullptr;
  hipStream_t stream = (hipStream_t)stream_;
  GNNRAG_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
  const int rc = gnnrag_reason_stack(csr, L, layers, h0, dist0, ins, relfeat_fwd, relfeat_inv, pos_rows, w_score,
                                     b_score, mask, h_out, score_out, dist_out, workspace, workspace_bytes, D, I, path,
                                     math, stream);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(stream, &graph);      // always end the capture, also after an error
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t exec = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e2 != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return (int)e2;
  }
  gnnrag_graph* g = new gnnrag_graph{graph, exec};
  *out = g;
  return 0;
}

extern "C" int gnnrag_graph_launch(gnnrag_graph* graph, gnnrag_stream_t stream) {
  if (!graph) return GNNRAG_E_BADARG;
  GNNRAG_HIP(hipGraphLaunch(graph->exec, (hipStream_t)stream));
  return 0;
}

extern "C" int gnnrag_graph_destroy(gnnrag_graph* graph) {
  if (!graph) return 0;
  (void)hipGraphExecDestroy(graph->exec);
  (void)hipGraphDestroy(graph->graph);
  delete graph;
  return 0;
}

extern "C" int gnnrag_abi_version(void) { return GNNRAG_ABI_VERSION; }

extern "C" const char* gnnrag_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case GNNRAG_E_BADARG: return "gnnrag: bad argument (null pointer, negative or inconsistent size)";
    case GNNRAG_E_UNSUPPORTED: return "gnnrag: shape outside the compiled kernel set";
    case GNNRAG_E_WORKSPACE: return "gnnrag: caller-provided buffer too small";
    case GNNRAG_E_TUPLE:
      return "gnnrag: edge tuple out of range: node ids must lie in [0, B*N), relation ids in [0, R1), and a fact "
             "may not connect two questions";
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "gnnrag: unknown error";
}
