// Relation-typed neighbour aggregation over the destination-sorted structure.
//
//   agg[n, 2i+d, :] = sum_{f : dst_d(f) = n}  w_f * dist[src_d(f)] * relu( T_d[rel_f, :] * ins[n/N, i, :] )
//
// = ReasonGNNLayer.reason_layer / reason_layer_inv of the reference
// (gnn/modules/kg_reasoning/reasongnn.py:61-89 / :91-116), with rel_linear hoisted from the
// gathered per-fact rows [F,D] to the relation table [R1,D] (it is row-wise, so this is the
// same arithmetic per row), and with the two torch.sparse.mm products (prior gather :80/:106,
// scatter-add :84/:111) replaced by a CSR walk.  No [F,D] temporary ever exists and there are
// no floating-point atomics: one (sub-)wavefront owns one destination node and accumulates its
// facts in ascending fact id in registers.
//
// Mapping to CDNA4 (wave64):
//  * a group of LPN lanes (16/32/64, chosen from D) owns one node; lane `sub` holds the
//    VEC-wide column chunks sub, sub+LPN, ... of the D-vector (float4 chunks for D%4==0);
//  * the group reads LPN (src, rel) records with ONE coalesced 8-byte load per lane, gathers
//    dist[src] per lane, then broadcasts (p, rel) fact by fact with v_readlane / ds_bpermute;
//  * facts whose prior is exactly 0 are skipped (they contribute exact zeros) - on the first
//    layer of every iteration dist is the seed distribution, so only the seeds' facts are live;
//  * the T_d row gather (D*4 bytes, contiguous) is served from L2: both tables are 2*R1*D*4
//    bytes (0.96 MB at R1=602, D=200) against 4 MB of L2 per XCD;
//  * one wave writes the complete [2I*D] output row of its node (3200 B at D=200, I=2), so
//    HBM sees full 128-byte lines; this write stream is the kernel's compulsory traffic;
//  * destination nodes with more than kHeavyDeg facts (Freebase hubs) are skipped here and
//    walked by a 16-wave workgroup that splits the fact range and reduces through LDS in a
//    fixed order (k_heavy) - deterministic, no atomics.
//
// TypeLayer.forward (gnn/modules/layer_init.py:25-62) is the same walk with p = v_f (or 1),
// no instruction, both directions summed and a final ReLU.
#include "gnnrag_common.h"

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
__device__ __forceinline__ typename VecT<VEC>::type vzero() {
  typename VecT<VEC>::type z = {};
  return z;
}
__device__ __forceinline__ float vrelu(float x) { return fmaxf(x, 0.f); }
__device__ __forceinline__ f32x2 vrelu(f32x2 x) { return __builtin_elementwise_max(x, (f32x2){0.f, 0.f}); }
__device__ __forceinline__ f32x4 vrelu(f32x4 x) {
  return __builtin_elementwise_max(x, (f32x4){0.f, 0.f, 0.f, 0.f});
}

enum { MODE_REASON = 0, MODE_TYPE = 1 };

struct WalkArgs {
  const int32_t* row_ptr[2];
  const int2* edge[2];
  const float* w[2];       // per-fact weight in sorted order or nullptr
  const float* T[2];       // relation tables [R1,D]
  const float* dist;       // [BN] (MODE_REASON)
  const float* ins;        // [B,I,D] (MODE_REASON)
  float* out;              // REASON: [BN,2I*D]; TYPE: [BN,D]
  const int32_t* heavy[2];
  const int32_t* n_heavy;
  int32_t heavy_cap;
  int32_t heavy_deg;
  int32_t BN, N, D, I, i0;
};

template <int LPN>
__device__ __forceinline__ float bcast_f(float v, int j) {
  if constexpr (LPN == 64) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
  } else {
    return __shfl(v, j, LPN);
  }
}
template <int LPN>
__device__ __forceinline__ int bcast_i(int v, int j) {
  if constexpr (LPN == 64) {
    return __builtin_amdgcn_readlane(v, j);
  } else {
    return __shfl(v, j, LPN);
  }
}

// Walks `len` facts starting at sorted position `beg` of one direction for the node owned by this
// lane group and accumulates into acc.  `maxlen` >= len is uniform over the wavefront (groups of
// one wave may own rows of different length).
template <int MODE, int VEC, int LPN, int CPL, int NI>
__device__ __forceinline__ void walk_row(const int2* __restrict__ edge, const float* __restrict__ w,
                                         const float* __restrict__ dist, const float* __restrict__ T,
                                         int D, int beg, int len, int maxlen, int sub,
                                         const int (&col)[CPL], const bool (&cv)[CPL],
                                         const typename VecT<VEC>::type (&q)[NI][CPL],
                                         typename VecT<VEC>::type (&acc)[NI][CPL]) {
  typedef typename VecT<VEC>::type V;
  for (int base = 0; base < maxlen; base += LPN) {
    float p = 0.f;
    int r = 0;
    if (base + sub < len) {
      const int idx = beg + base + sub;
      const int2 e = edge[idx];
      r = e.y;
      if constexpr (MODE == MODE_REASON) {
        p = dist[e.x];
        if (w) p *= w[idx];
      } else {
        p = w ? w[idx] : 1.f;
      }
    }
    const int cnt = min(LPN, maxlen - base);
    if constexpr (LPN == 64) {
      // one node per wave: (p, rel) are wave-uniform -> scalar control flow, skip dead facts
      unsigned long long live = __ballot(p != 0.f);
      while (live) {
        const int j = __builtin_ctzll(live);
        live &= live - 1;
        const float pj = bcast_f<64>(p, j);
        const int rj = bcast_i<64>(r, j);
        const float* trow = T + (size_t)rj * D;
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
          if (cv[m]) {
            const V t = vload<VEC>(trow + col[m]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              if constexpr (MODE == MODE_REASON) acc[i][m] += pj * vrelu(t * q[i][m]);
              else acc[i][m] += pj * t;
            }
          }
        }
      }
    } else {
      for (int j = 0; j < cnt; ++j) {
        const float pj = bcast_f<LPN>(p, j);
        const int rj = bcast_i<LPN>(r, j);
        if (pj != 0.f) {
          const float* trow = T + (size_t)rj * D;
#pragma unroll
          for (int m = 0; m < CPL; ++m) {
            if (cv[m]) {
              const V t = vload<VEC>(trow + col[m]);
#pragma unroll
              for (int i = 0; i < NI; ++i) {
                if constexpr (MODE == MODE_REASON) acc[i][m] += pj * vrelu(t * q[i][m]);
                else acc[i][m] += pj * t;
              }
            }
          }
        }
      }
    }
  }
}

template <int LPN>
__device__ __forceinline__ int wave_max_over_groups(int v) {
#pragma unroll
  for (int o = LPN; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- light rows: one LPN-lane group per destination node, both directions ------------------
template <int MODE, int VEC, int LPN, int CPL, int NI>
__global__ __launch_bounds__(256) void k_walk_light(const WalkArgs a) {
  typedef typename VecT<VEC>::type V;
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  const int sub = threadIdx.x & (LPN - 1);
  int n = gtid / LPN;
  const bool live = n < a.BN;
  if (!live) n = a.BN - 1;  // keep every lane in the shuffles
  const int D = a.D;
  int col[CPL];
  bool cv[CPL];
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    col[m] = (sub + m * LPN) * VEC;
    cv[m] = col[m] < D;
  }
  V q[NI][CPL];
  if constexpr (MODE == MODE_REASON) {
    const int b = n / a.N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int m = 0; m < CPL; ++m)
        q[i][m] = cv[m] ? vload<VEC>(a.ins + ((size_t)b * a.I + a.i0 + i) * D + col[m]) : vzero<VEC>();
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int m = 0; m < CPL; ++m) q[i][m] = vzero<VEC>();
  }

  V acc[NI][CPL];
  bool any_heavy = false;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (MODE == MODE_REASON || d == 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int m = 0; m < CPL; ++m) acc[i][m] = vzero<VEC>();
    }
    const int beg = a.row_ptr[d][n];
    int len = a.row_ptr[d][n + 1] - beg;
    const bool heavy = len > a.heavy_deg;
    if (!live || heavy) len = 0;
    any_heavy |= heavy;
    const int maxlen = wave_max_over_groups<LPN>(len);
    walk_row<MODE, VEC, LPN, CPL, NI>(a.edge[d], a.w[d], a.dist, a.T[d], D, beg, len, maxlen, sub, col,
                                      cv, q, acc);
    if constexpr (MODE == MODE_REASON) {
      if (live && !heavy) {
        float* orow = a.out + (size_t)n * (2 * a.I) * D;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int m = 0; m < CPL; ++m)
            if (cv[m]) vstore<VEC>(orow + (size_t)(2 * (a.i0 + i) + d) * D + col[m], acc[i][m]);
      }
    }
  }
  if constexpr (MODE == MODE_TYPE) {
    // both directions summed, then ReLU (layer_init.py:57).  A node with a heavy direction is
    // finished by k_walk_heavy, which re-walks both of its rows.
    if (live && !any_heavy) {
#pragma unroll
      for (int m = 0; m < CPL; ++m)
        if (cv[m]) vstore<VEC>(a.out + (size_t)n * D + col[m], vrelu(acc[0][m]));
    }
  }
}

// ---- heavy rows: a 16-wave workgroup per destination node ------------------------------------
// grid = (blocks, 2 directions).  Each wave walks one contiguous 1/16 slice of the row; partials
// go through LDS and are summed in wave order (fixed order => deterministic).
template <int MODE, int VEC, int CPL, int NI>
__global__ __launch_bounds__(1024) void k_walk_heavy(const WalkArgs a) {
  typedef typename VecT<VEC>::type V;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16][NI][D]
  const int d = blockIdx.y;
  const int wave = threadIdx.x >> 6;
  const int sub = threadIdx.x & 63;
  const int D = a.D;
  int cnt = a.n_heavy[d];
  if (cnt > a.heavy_cap) cnt = a.heavy_cap;
  int col[CPL];
  bool cv[CPL];
#pragma unroll
  for (int m = 0; m < CPL; ++m) {
    col[m] = (sub + m * 64) * VEC;
    cv[m] = col[m] < D;
  }
  for (int h = blockIdx.x; h < cnt; h += gridDim.x) {
    const int n = a.heavy[d][h];
    V q[NI][CPL];
    if constexpr (MODE == MODE_REASON) {
      const int b = n / a.N;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int m = 0; m < CPL; ++m)
          q[i][m] = cv[m] ? vload<VEC>(a.ins + ((size_t)b * a.I + a.i0 + i) * D + col[m]) : vzero<VEC>();
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int m = 0; m < CPL; ++m) q[i][m] = vzero<VEC>();
    }
    V acc[NI][CPL];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int m = 0; m < CPL; ++m) acc[i][m] = vzero<VEC>();

    // MODE_TYPE sums both directions of node n; a node heavy in both directions appears in both
    // lists, so only the lower-numbered heavy direction finishes it.
    bool mine = true;
    if constexpr (MODE == MODE_TYPE) {
      if (d == 1) {
        const int l0 = a.row_ptr[0][n + 1] - a.row_ptr[0][n];
        if (l0 > a.heavy_deg) mine = false;
      }
    }
    if (mine) {
      const int ndir = (MODE == MODE_TYPE) ? 2 : 1;
      for (int dd = 0; dd < ndir; ++dd) {
        const int dir = (MODE == MODE_TYPE) ? dd : d;
        const int beg = a.row_ptr[dir][n];
        const int len = a.row_ptr[dir][n + 1] - beg;
        int per = (len + 15) / 16;
        per = (per + 63) & ~63;                      // whole 64-fact batches per wave
        const int wbeg = min(len, wave * per);
        const int wlen = min(len, wbeg + per) - wbeg;
        walk_row<MODE, VEC, 64, CPL, NI>(a.edge[dir], a.w[dir], a.dist, a.T[dir], D, beg + wbeg, wlen, wlen,
                                         sub, col, cv, q, acc);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int m = 0; m < CPL; ++m)
          if (cv[m]) vstore<VEC>(red + ((size_t)wave * NI + i) * D + col[m], acc[i][m]);
    }
    __syncthreads();
    if (mine) {
      for (int e = threadIdx.x; e < NI * D; e += 1024) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) s += red[(size_t)wv * NI * D + e];
        const int i = e / D, c = e - i * D;
        if constexpr (MODE == MODE_REASON)
          a.out[(size_t)n * (2 * a.I) * D + (size_t)(2 * (a.i0 + i) + d) * D + c] = s;
        else
          a.out[(size_t)n * D + c] = fmaxf(s, 0.f);
      }
    }
    __syncthreads();
  }
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
static int launch_one(const WalkArgs& a, hipStream_t stream) {
  const int groups_per_block = 256 / LPN;
  const int nblk = (a.BN + groups_per_block - 1) / groups_per_block;
  hipLaunchKernelGGL((k_walk_light<MODE, VEC, LPN, CPL, NI>), dim3(nblk), dim3(256), 0, stream, a);
  GNNRAG_LAUNCH_CHECK();
  const size_t lds = (size_t)16 * NI * a.D * sizeof(float);
  if (lds > 160 * 1024) return GNNRAG_E_UNSUPPORTED;
  hipLaunchKernelGGL((k_walk_heavy<MODE, VEC, CPL, NI>), dim3(128, 2), dim3(1024), lds, stream, a);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

template <int MODE, int VEC, int LPN, int CPL>
static int launch_ni(const WalkArgs& a, int ni, hipStream_t stream) {
  if constexpr (MODE == MODE_TYPE) {
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

static void fill_common(WalkArgs& a, const gnnrag_csr* csr, int D) {
  for (int d = 0; d < 2; ++d) {
    a.row_ptr[d] = csr->row_ptr[d];
    a.edge[d] = (const int2*)csr->edge[d];
    a.heavy[d] = csr->heavy[d];
  }
  a.n_heavy = csr->n_heavy;
  a.heavy_cap = csr->heavy_cap;
  a.heavy_deg = csr->heavy_deg;
  a.BN = csr->B * csr->N;
  a.N = csr->N;
  a.D = D;
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" int gnnrag_aggregate(const gnnrag_csr* csr, const float* dist, const float* ins,
                                const float* T_fwd, const float* T_inv, float* agg, int32_t D, int32_t I,
                                gnnrag_stream_t stream) {
  if (!csr || !dist || !ins || !T_fwd || !T_inv || !agg || D <= 0 || I <= 0) return GNNRAG_E_BADARG;
  WalkArgs a;
  memset(&a, 0, sizeof(a));
  fill_common(a, csr, D);
  a.w[0] = csr->w_gnn[0];
  a.w[1] = csr->w_gnn[1];
  a.T[0] = T_fwd;
  a.T[1] = T_inv;
  a.dist = dist;
  a.ins = ins;
  a.out = agg;
  a.I = I;
  // up to 3 instructions share one walk (their accumulators live in registers side by side)
  for (int i0 = 0; i0 < I; i0 += 3) {
    a.i0 = i0;
    const int ni = (I - i0) < 3 ? (I - i0) : 3;
    const int rc = launch_walk<MODE_REASON>(a, ni, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int gnnrag_typelayer(const gnnrag_csr* csr, const float* T, int use_w_rel, float* h0, int32_t D,
                                gnnrag_stream_t stream) {
  if (!csr || !T || !h0 || D <= 0) return GNNRAG_E_BADARG;
  if (use_w_rel && (!csr->w_rel[0] || !csr->w_rel[1])) return GNNRAG_E_BADARG;
  WalkArgs a;
  memset(&a, 0, sizeof(a));
  fill_common(a, csr, D);
  a.w[0] = use_w_rel ? csr->w_rel[0] : nullptr;
  a.w[1] = use_w_rel ? csr->w_rel[1] : nullptr;
  a.T[0] = T;
  a.T[1] = T;
  a.out = h0;
  a.I = 1;
  a.i0 = 0;
  return launch_walk<MODE_TYPE>(a, 1, (hipStream_t)stream);
}
