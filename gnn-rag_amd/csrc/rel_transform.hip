// Relation projections of ALL layers of one iteration in one launch:
//
//   T[j][d][r, :] = rel_linear{j}(rel_features_d[r, :]) (+ pos_emb{j}_d[r, :])     reasongnn.py:75-79 / :102-105
//
// for layer j = 0..L-1 and direction d (forward / inverse relation features).  None of them depends on the node
// state or on the distribution, so the L x 2 skinny products ([R1, D] x [D, D], R1 = a few hundred to a few thousand
// relation rows) that used to head every layer call - each a ~13-25 us launch of a few dozen workgroups waiting on a
// dependent k loop - become one launch of (row blocks x column tiles x 2L) small workgroups in front of the layer
// sequence.  Exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32).
//
// Workgroup = 4 waves x (1 or 4) 16-row tiles x columns [64 by, +64) of T[z] as four MFMA tiles that share the A
// fragment.  The weight slice W[64 by .. +63][:] is staged ONCE per workgroup into LDS in fragment order (51 KB at
// K = 200); lane (fr, fg) loads the float4 A[row fr][16 c + 4 fg ..] of k group c straight from global memory
// (kRtUnroll groups in flight) and reads, per tile a, the float4 W[column 64 by + 4 fr + a][16 c + 4 fg ..] from LDS;
// component s of the float4s feeds MFMA s (the k order inside a group is a permutation both operands share).  The
// column slots are interleaved (slot fr of tile a = column 4 fr + a), so the epilogue stores 4 consecutive columns
// per lane - T as float4, the bf16 planes as 8-byte pieces.
#include "gnnrag_common.h"
#include "dense_internal.h"

namespace gnnrag {

constexpr int kRtMaxL = 8;        // layers per launch (more layers: more launches)
constexpr int kRtUnroll = 7;      // k groups (16 columns each) of A in flight per wave: K = 200 -> 13 groups -> 2 rounds

struct RelTArgs {
  const float* A[2];              // rel_features, rel_features_inv  [M, K]
  const float* W[kRtMaxL];        // rel_linear{j}.weight [N, K]
  const float* b[kRtMaxL];        // rel_linear{j}.bias [N] or null
  const float* add[kRtMaxL][2];   // pos_emb{j} / pos_emb_inv{j} [add_rows, N] or null
  float* C;                       // [L][2][M][N]
  unsigned short* planes;         // null or [L][2][3][M][kPlaneRow] bf16: planes of relu(T) | relu(-T), see below
  int M, K, N, add_rows;
};

// Planes for the "V form" of the relation tables (tables_b3.hip: k_tables_vq): per (layer, direction) three bf16
// planes (exact 3-way split) of [relu(T[r, :]) | relu(-T[r, :])], each half padded with zeros to 7 k blocks of 32.
constexpr int kPlaneHalf = 224;
constexpr int kPlaneRow = 2 * kPlaneHalf;     // bf16 elements per plane row (896 bytes)

// TPW: 16-row tiles per wave (1: few relation rows, many small workgroups; 4: large vocabularies, the weight slice
// staged once per 256 rows)
template <int TPW>
__global__ __launch_bounds__(256) void k_rel_transform(RelTArgs g) {
  extern __shared__ __attribute__((aligned(16))) float Wf[];  // the weight slice in FRAGMENT order, see below
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int j = blockIdx.z >> 1, d = blockIdx.z & 1;
  const int colg = blockIdx.y * 64;                          // this workgroup's 64 columns: four MFMA tiles
  const int K = g.K;
  const int nch = (K + 15) >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool compute = colg < g.N;                           // (column groups past N only write the planes' zero padding)
  // column slot fr of tile a stands for column colg + 4 fr + a: a lane ends up with 4 CONSECUTIVE columns of its rows
  // (16-byte stores of T, 8-byte stores of the bf16 planes).  LDS holds W[colg .. colg+63][:] as the lanes will read
  // it: float4 number ((c * 4 + a) * 64 + lane) = W[colg + 4 fr + a][16 c + 4 fg ..] - one contiguous KB per
  // (k group c, tile a) and wave instruction, conflict free.
  if (compute) {
    const int total = 64 * nch * 4;                          // float4 pieces: 64 columns x (K/4 padded to 4 nch)
    for (int x = tid; x < total; x += 256) {
      const int col = x / (nch * 4), kq = x - col * (nch * 4);       // kq = k / 4: consecutive threads, consecutive k
      f32x4 v = zero4;
      if (colg + col < g.N && 4 * kq < K) v = *reinterpret_cast<const f32x4*>(g.W[j] + (size_t)(colg + col) * K + 4 * kq);
      const int c = kq >> 2, fgq = kq & 3, a = col & 3, frq = col >> 2;
      *reinterpret_cast<f32x4*>(Wf + ((size_t)((c * 4 + a) * 64 + fgq * 16 + frq)) * 4) = v;
    }
  }
  __syncthreads();
  const int c4 = colg + 4 * fr;
  const bool in_T = c4 < g.N;                                // (N % 4 == 0: a lane's 4 columns are all in or all out)
  const bool in_pad = !in_T && c4 < kPlaneHalf;
  f32x4 bias = zero4;
  if (in_T && g.b[j]) bias = *reinterpret_cast<const f32x4*>(g.b[j] + c4);
  const float* __restrict__ addp = g.add[j][d];
  float* __restrict__ cp = g.C + ((size_t)(2 * j + d) * g.M) * g.N;
  unsigned short* pp = g.planes ? g.planes + ((size_t)(2 * j + d) * 3 * g.M) * kPlaneRow : nullptr;

#pragma unroll 1
  for (int t = 0; t < TPW; ++t) {
    const int row0 = (blockIdx.x * TPW + t) * 64 + wave * 16;
    if (row0 >= g.M) break;
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = zero4;
    if (compute) {
      // rows past the end read the last valid one (never stored); the A fragments come straight from global memory
      const float* __restrict__ ap = g.A[d] + (size_t)min(row0 + fr, g.M - 1) * K;
      for (int c0 = 0; c0 < nch; c0 += kRtUnroll) {
        f32x4 av[kRtUnroll];
#pragma unroll
        for (int u = 0; u < kRtUnroll; ++u)    // k groups past K: a clamped (valid) address, zeroed at use - no branch around a load
          av[u] = *reinterpret_cast<const f32x4*>(ap + min(16 * (c0 + u) + 4 * fg, K - 4));
#pragma unroll
        for (int u = 0; u < kRtUnroll; ++u) {
          const int c = min(c0 + u, nch - 1);
          const bool live = 16 * (c0 + u) + 4 * fg < K;
          const f32x4 x = live ? av[u] : zero4;
          f32x4 wv[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) wv[a] = *reinterpret_cast<const f32x4*>(Wf + ((size_t)((c * 4 + a) * 64 + lane)) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < 4; ++a)
              acc[a][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], wv[a][e], acc[a][e & 1], 0, 0, 0);
        }
      }
    }
    // accumulator element r of lane (fr, fg), tile a = C[row0 + 4 fg + r][colg + 4 fr + a]
    if (!in_T && !(pp && in_pad)) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * fg + r;
      if (row >= g.M) continue;
      f32x4 v = zero4;
      if (in_T) {
#pragma unroll
        for (int a = 0; a < 4; ++a) v[a] = acc[a][0][r] + acc[a][1][r];
        v += bias;
        if (addp && row < g.add_rows) v += *reinterpret_cast<const f32x4*>(addp + (size_t)row * g.N + c4);
        *reinterpret_cast<f32x4*>(cp + (size_t)row * g.N + c4) = v;
      }
      if (pp) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          // exact 3-way bf16 split of relu(+-T) (gnnrag_common.h: split3); the padding columns get zeros
          const Split3 sp = split3(__builtin_elementwise_max(half ? -v : v, zero4));
          const size_t e = (size_t)row * kPlaneRow + half * kPlaneHalf + c4;
          *reinterpret_cast<uint2*>(pp + e) = sp.hi;
          *reinterpret_cast<uint2*>(pp + (size_t)g.M * kPlaneRow + e) = sp.mid;
          *reinterpret_cast<uint2*>(pp + (size_t)2 * g.M * kPlaneRow + e) = sp.lo;
        }
      }
    }
  }
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" size_t gnnrag_rel_planes_bytes(int64_t R1, int32_t D, int32_t L) {
  if (R1 <= 0 || D <= 0 || D > kPlaneHalf || L <= 0) return 0;
  return (size_t)L * 2 * 3 * R1 * kPlaneRow * sizeof(unsigned short);
}

// THE acceptance test of gnnrag_rel_transform (shapes, offsets, alignment of every operand incl. the outputs, LDS) as
// one function: the layer stack's GNNRAG_PATH_REUSE_PROJ call asks it, with the real pointers, whether the projecting
// call took this kernel (and so wrote the planes) or fell back to the k-tiled projection.
bool gnnrag::rel_transform_accepts(const float* relfeat_fwd, const float* relfeat_inv, int64_t R1, int32_t D, int32_t L,
                                   const gnnrag_layer_params* layers, int32_t pos_rows, const float* T_out,
                                   const void* planes_out) {
  if (D <= 0 || (D & 3)) return false;                                  // float4 operand loads
  if (planes_out && D > kPlaneHalf) return false;
  if (R1 >= ((int64_t)1 << 31) / D) return false;
  uintptr_t align = (uintptr_t)relfeat_fwd | (uintptr_t)relfeat_inv | (uintptr_t)T_out | (uintptr_t)planes_out;
  for (int j = 0; j < L; ++j) {
    align |= (uintptr_t)layers[j].W_rel | (uintptr_t)layers[j].b_rel;
    if (pos_rows > 0) align |= (uintptr_t)layers[j].pos_fwd | (uintptr_t)layers[j].pos_inv;
  }
  if (align & 15) return false;                                         // float4 accesses
  return (size_t)64 * ((D + 15) / 16) * 16 * sizeof(float) <= 160 * 1024;   // 64 columns x K padded to k groups in LDS
}

extern "C" int gnnrag_rel_transform(const float* relfeat_fwd, const float* relfeat_inv, int64_t R1, int32_t D,
                                    int32_t L, const gnnrag_layer_params* layers, int32_t pos_rows, float* T_out,
                                    void* planes_out, gnnrag_stream_t stream) {
  if (!relfeat_fwd || !relfeat_inv || !layers || !T_out || R1 < 0 || D <= 0 || L <= 0 || pos_rows < 0)
    return GNNRAG_E_BADARG;
  for (int j = 0; j < L; ++j) {
    if (!layers[j].W_rel) return GNNRAG_E_BADARG;
    if ((layers[j].pos_fwd == nullptr) != (layers[j].pos_inv == nullptr)) return GNNRAG_E_BADARG;
  }
  if (!gnnrag::rel_transform_accepts(relfeat_fwd, relfeat_inv, R1, D, L, layers, pos_rows, T_out, planes_out))
    return GNNRAG_E_UNSUPPORTED;
  if (R1 == 0) return 0;
  for (int j0 = 0; j0 < L; j0 += kRtMaxL) {
    const int n = L - j0 < kRtMaxL ? L - j0 : kRtMaxL;
    RelTArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = relfeat_fwd;
    g.A[1] = relfeat_inv;
    for (int j = 0; j < n; ++j) {
      const gnnrag_layer_params& p = layers[j0 + j];
      g.W[j] = p.W_rel;
      g.b[j] = p.b_rel;
      g.add[j][0] = pos_rows > 0 ? p.pos_fwd : nullptr;
      g.add[j][1] = pos_rows > 0 ? p.pos_inv : nullptr;
    }
    g.C = T_out + (size_t)j0 * 2 * R1 * D;
    g.planes = planes_out ? (unsigned short*)planes_out + (size_t)j0 * 2 * 3 * R1 * kPlaneRow : nullptr;
    g.M = (int)R1; g.K = D; g.N = D;
    g.add_rows = (int)(pos_rows < R1 ? pos_rows : R1);
    // with planes the column groups run on to 224 columns: the part past D only writes the zero padding
    const unsigned gy = (unsigned)(((planes_out ? kPlaneHalf : D) + 63) / 64), gz = (unsigned)(2 * n);
    const size_t lds = (size_t)64 * ((D + 15) / 16) * 16 * sizeof(float);      // 64 columns x K padded to k groups
    static DeviceMask cap1, cap4;
    if (R1 > 2048) {
      if (lds > 64 * 1024) { const int rc = raise_lds_cap(k_rel_transform<4>, cap4); if (rc) return rc; }
      hipLaunchKernelGGL(k_rel_transform<4>, dim3((unsigned)((R1 + 255) / 256), gy, gz), dim3(256), lds,
                         (hipStream_t)stream, g);
    } else {
      if (lds > 64 * 1024) { const int rc = raise_lds_cap(k_rel_transform<1>, cap1); if (rc) return rc; }
      hipLaunchKernelGGL(k_rel_transform<1>, dim3((unsigned)((R1 + 63) / 64), gy, gz), dim3(256), lds,
                         (hipStream_t)stream, g);
    }
    GNNRAG_LAUNCH_CHECK();
  }
  return 0;
}
