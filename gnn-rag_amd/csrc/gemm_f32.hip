// Exact-fp32 dense projections on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   C[M, Nout] = act( A[M, K] . W[Nout, K]^T + bias + add )       (W is an nn.Linear weight)
//
// used for
//   * T_d = rel_linear_step(rel_features_d) (+ pos_emb)            reasongnn.py:75-79 / :102-105
//   * kb_self_linear(rel_features)                                 layer_init.py:47-49
//   * h' = relu(e2e_linear_step(cat(h, agg))), fused with
//     score = score_func(h') + (1 - mask) * -1e11                  reasongnn.py:161-168
//
// The reference needs 1e-4 fp32 parity, so bf16/fp8 MFMA are out; gfx950 has no xf32.  The
// f32-input MFMA is bit-for-bit an fmaf chain and runs at the fp32 vector peak (157 TF), about
// 2.4x what a VALU GEMM reaches (guide: cdna_hip_programming.md section 3).
//
// Tiling (wave64): workgroup = 4 waves = 128 rows x all Nout columns (Nout = D <= 208 for the
// fused update, so A - the big operand, [B*N, (2I+1)D] - streams from HBM exactly once);
// wave w owns rows [32w, 32w+32) as 2 x NT accumulator tiles of 16x16 (NT = ceil(Nout/16)).
// K is consumed in 32-wide tiles staged through LDS (row stride 40 floats: conflict-free for the
// ds_read_b128 fragment reads below and for the ds_write_b128 staging writes).  A lane reads a
// float4 = 4 consecutive k of its row; register s of that float4 feeds MFMA sub-step s, so the
// k index a lane group supplies is 16c + 4*(lane>>4) + s for both operands - a permutation of
// the k order inside the tile that A and W share, which the sum does not see.  Global loads of
// tile t+1 are issued before the MFMAs of tile t (register prefetch), two workgroups per CU.
#include "gnnrag_common.h"
#include "dense_internal.h"

// Variants that were A/B-tested on MI355X and did NOT pay (removed; see DESIGN.md section 3.5 and git history):
// branch-free clamped tile loads (-8 %), source-level fragment double buffering / sched_group_barrier
// pinning (+-0), A tiles requested two k-tiles ahead (+-0), prefetching all A tiles of a short-K problem
// (register blow-up), direct stores from the MFMA layout instead of the LDS-staged epilogue (-25 % on the
// self-block update).
#define GNNRAG_GEMM_MT1_NW 8     // the one-row-tile-per-wave variant may use 8-wave (128-row) workgroups
#ifndef GNNRAG_UPDATE_B3
#define GNNRAG_UPDATE_B3 1       // bf16x3 self-block update on the W-resident kernel of tables_b3.hip
#endif
#ifndef GNNRAG_TABLES_WRES
#define GNNRAG_TABLES_WRES 1     // bf16x3 relation tables on the W-resident kernel of tables_b3.hip
#endif
#ifndef GNNRAG_GEMM_WRES
#define GNNRAG_GEMM_WRES 1       // short-K problems in exact fp32 run the W-resident kernel (k_gemm_wres)
#endif
#ifndef GNNRAG_UPDATE_SKINNY
#define GNNRAG_UPDATE_SKINNY 1   // self-block update of small batches (< 4096 rows) on the one-wave-per-tile kernel
#endif

namespace gnnrag {


constexpr int kBK = 32;    // k per LDS tile
constexpr int kSkinnyMaxM = 16384;   // up to here a problem runs on k_gemm_skinny (one wave per 16 x 64 tile, no LDS)

enum { EPI_LINEAR = 0, EPI_UPDATE = 1 };

struct GemmArgs {
  const float* A0;      // [M, K0]  (plain: K0 = K; update: h, K0 = D)
  const float* A1;      // [M, K-K0] or nullptr (update: agg)
  const float* A0b;     // AMODE_GEN: table of direction 1 (A0 = direction 0); skinny pair: second problem's A
  const float* add_b;   // skinny pair: second problem's add / C (blockIdx.z = 1)
  float* C_b;
  const float* W;       // [Nout, K]
  const float* bias;    // [Nout] or nullptr
  const float* add;     // [add_rows, Nout] or nullptr
  float* C;             // [M, Nout]
  const float* w_s;     // EPI_UPDATE: score_func.weight [Nout]
  const float* b_s;     // EPI_UPDATE: score_func.bias [1]
  const float* mask;    // EPI_UPDATE: [M]
  float* score;         // EPI_UPDATE: [M]
  int32_t M, K, K0, Nout, add_rows, relu;
  int32_t n0;           // first output column of this launch's column block (EPI_LINEAR, Nout > 208)
  int32_t ldw;          // row stride of W in floats (>= K; e2e_linear column blocks are used in place)
  int32_t wc0;          // first column of W used (AMODE_PLAIN)
  // AMODE_GEN (per-question relation tables): row m = compact row (b, r) = gen_rows[m], column k = (i, kk):
  //   A[m,k] = relu(A0[r*D + kk] * A1[(b*I + i)*D + kk]),  W column = (1 + 2i + gen_dir)*D + kk
  const int2* gen_rows;
  int32_t gen_D, gen_I, gen_dir;
  int32_t v4out;        // rows of C/add are 16-byte aligned and Nout % 4 == 0: float4 epilogue
  const uint8_t* add_flag;   // k_gemm_wres<.., AFL>: row gates of `add` [M + 4]; a row whose byte is 0 reads the zero row
                             // `add + M * Nout` behind the buffer (frontier layers, frontier.hip)
};

enum { AMODE_PLAIN = 0, AMODE_GEN = 1 };

// 4 consecutive k of logical row m of A (= [A0 | A1], or generated); zero beyond M / K.
template <bool V4, int AMODE>
__device__ __forceinline__ f32x4 load_a4(const GemmArgs& g, int m, int k) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (m >= g.M) return v;
  if constexpr (AMODE == AMODE_GEN) {
    // relu(T_d[r,:] * ins[b,i,:]) generated on the fly: the [B*R1, I*D] operand never exists in HBM
    if (k >= g.K) return v;
    const int2 br = g.gen_rows[m];
    const int b = br.x, r = br.y;
    const int i = k / g.gen_D, kk = k - i * g.gen_D;
    const float* tp = g.A0 + (size_t)r * g.gen_D + kk;
    const float* qp = g.A1 + ((size_t)b * g.gen_I + i) * g.gen_D + kk;
    if constexpr (V4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(tp);
      const f32x4 q = *reinterpret_cast<const f32x4*>(qp);
      v = __builtin_elementwise_max(t * q, v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k2 = k + j;
        if (k2 < g.K) {
          const int i2 = k2 / g.gen_D, kk2 = k2 - i2 * g.gen_D;
          v[j] = fmaxf(g.A0[(size_t)r * g.gen_D + kk2] * g.A1[((size_t)b * g.gen_I + i2) * g.gen_D + kk2], 0.f);
        }
      }
    }
    return v;
  } else if constexpr (V4) {
    // K0, K-K0 multiples of 4 and 16-byte aligned bases: a float4 never straddles the split.
    if (k < g.K0) v = *reinterpret_cast<const f32x4*>(g.A0 + (size_t)m * g.K0 + k);
    else if (k < g.K) v = *reinterpret_cast<const f32x4*>(g.A1 + (size_t)m * (g.K - g.K0) + (k - g.K0));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k + j;
      if (kk < g.K0) v[j] = g.A0[(size_t)m * g.K0 + kk];
      else if (kk < g.K) v[j] = g.A1[(size_t)m * (g.K - g.K0) + (kk - g.K0)];
    }
  }
  return v;
}

// column of W that multiplies logical k
template <int AMODE>
__device__ __forceinline__ int w_col(const GemmArgs& g, int k) {
  if constexpr (AMODE == AMODE_GEN) {
    const int i = k / g.gen_D;
    return (1 + 2 * i + g.gen_dir) * g.gen_D + (k - i * g.gen_D);
  } else {
    return g.wc0 + k;
  }
}

template <bool V4, int AMODE>
__device__ __forceinline__ f32x4 load_w4(const GemmArgs& g, int j, int k) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if constexpr (V4) {
    if (j < g.Nout && k < g.K) v = *reinterpret_cast<const f32x4*>(g.W + (size_t)j * g.ldw + w_col<AMODE>(g, k));
  } else {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (j < g.Nout && k + jj < g.K) v[jj] = g.W[(size_t)j * g.ldw + w_col<AMODE>(g, k + jj)];
  }
  return v;
}



// MATH = 0: v_mfma_f32_16x16x4_f32 (bit-exact fmaf chains).
// MATH = 1: every fp32 operand is split exactly into three bf16 planes (hi, mid, lo) when its tile is
//   staged to LDS and the product is formed from the six plane pairs that matter (hi*hi, hi*mid,
//   mid*hi, hi*lo, lo*hi, mid*mid; the dropped ones are <= 3 * 2^-24 relative) with
//   v_mfma_f32_16x16x32_bf16, fp32 accumulate: fp32-class accuracy at 6/16 of the fp32 MFMA time.
template <int NT, int MT, bool V4, int EPI, int AMODE, int MATH, int NW>
__global__ __launch_bounds__(NW * 64, (MT == 2 ? 2 : NW == 8 ? 4 : 3))
void k_gemm_f32(GemmArgs g) {
  constexpr int RPR = NW * 8;                               // staging rows per round (8 threads per row)
  constexpr int kLS = MATH ? 56 : 40;         // LDS row stride in floats (both conflict-free for ds_read_b128)
  constexpr int BM = 16 * MT * NW;            // rows per workgroup: NW waves x MT accumulator row-tiles of 16
  constexpr int AR = BM / RPR;                // A staging rounds
  constexpr int WR = (NT * 16 + RPR - 1) / RPR;   // W staging rounds
  constexpr int LD = NT * 16 + 4;             // epilogue staging row stride (floats)
  constexpr int kTileFloats = (BM + NT * 16) * kLS;
  constexpr int kStageFloats = NW * 8 * LD;    // epilogue: 8 rows per wave at a time
  constexpr int kSmemFloats = kTileFloats > kStageFloats ? kTileFloats : kStageFloats;
  __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];
  float* As = smem;
  float* Ws = smem + BM * kLS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lr = tid >> 3;   // staging row 0..RPR-1
  const int kq = tid & 7;    // staging float4 within the 32-wide k tile
  const int m0 = blockIdx.x * BM;
  const int n0 = g.n0;

  // AMODE_GEN: blockIdx.y = direction; per-thread row bases are fixed for the whole K loop and the
  // (instruction i, column kk) of the thread's k position advances incrementally - no divisions
  // in the loop.
  const float* gen_t[AR];
  const float* gen_q[AR];
  int gen_i = 0, gen_kk = 0;
  if constexpr (AMODE == AMODE_GEN) {
    const int dir = blockIdx.y;
    g.gen_dir = dir;
    if (dir) g.A0 = g.A0b;
    g.C += (size_t)dir * g.M * g.Nout;
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const int m = m0 + lr + RPR * r;
      const int2 br = g.gen_rows[m < g.M ? m : 0];     // (question, relation) of this compact row
      gen_t[r] = (m < g.M) ? g.A0 + (size_t)br.y * g.gen_D : nullptr;
      gen_q[r] = g.A1 + (size_t)br.x * g.gen_I * g.gen_D;
    }
    gen_i = (kq * 4) / g.gen_D;
    gen_kk = kq * 4 - gen_i * g.gen_D;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[AR], rw[WR];
  const int nT = (g.K + kBK - 1) / kBK;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto gload = [&](int t) {
    const int k = t * kBK + kq * 4;
    if constexpr (AMODE == AMODE_GEN && V4) {
      const bool kok = k < g.K;
      const int wcol = (1 + 2 * gen_i + g.gen_dir) * g.gen_D + gen_kk;
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        ra[r] = zero4;
        if (kok && gen_t[r]) {
          const f32x4 tv = *reinterpret_cast<const f32x4*>(gen_t[r] + gen_kk);
          const f32x4 qv = *reinterpret_cast<const f32x4*>(gen_q[r] + (size_t)gen_i * g.gen_D + gen_kk);
          ra[r] = __builtin_elementwise_max(tv * qv, zero4);
        }
      }
#pragma unroll
      for (int r = 0; r < WR; ++r) {
        const int j = lr + RPR * r;
        rw[r] = zero4;
        if (j < NT * 16 && n0 + j < g.Nout && kok)
          rw[r] = *reinterpret_cast<const f32x4*>(g.W + (size_t)(n0 + j) * g.ldw + wcol);
      }
      gen_kk += kBK;                       // next tile's column state
      while (gen_kk >= g.gen_D) { gen_kk -= g.gen_D; ++gen_i; }
    } else {
#pragma unroll
      for (int r = 0; r < AR; ++r) ra[r] = load_a4<V4, AMODE>(g, m0 + lr + RPR * r, k);
#pragma unroll
      for (int r = 0; r < WR; ++r) {
        const int j = lr + RPR * r;
        rw[r] = (j < NT * 16) ? load_w4<V4, AMODE>(g, n0 + j, k) : zero4;
      }
    }
  };
  auto sstore = [&]() {
    if constexpr (MATH == 0) {
#pragma unroll
      for (int r = 0; r < AR; ++r) *reinterpret_cast<f32x4*>(&As[(lr + RPR * r) * kLS + kq * 4]) = ra[r];
#pragma unroll
      for (int r = 0; r < WR; ++r) {
        const int j = lr + RPR * r;
        if (j < NT * 16) *reinterpret_cast<f32x4*>(&Ws[j * kLS + kq * 4]) = rw[r];
      }
    } else {
      // row image: [hi: 32 bf16 | mid: 32 bf16 | lo: 32 bf16 | pad] = 224 bytes; this thread owns k 4kq..4kq+3
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        const Split3 sp = split3(ra[r]);
        float* row = &As[(lr + RPR * r) * kLS + kq * 2];
        *reinterpret_cast<uint2*>(row) = sp.hi;
        *reinterpret_cast<uint2*>(row + 16) = sp.mid;
        *reinterpret_cast<uint2*>(row + 32) = sp.lo;
      }
#pragma unroll
      for (int r = 0; r < WR; ++r) {
        const int j = lr + RPR * r;
        if (j < NT * 16) {
          const Split3 sp = split3(rw[r]);
          float* row = &Ws[j * kLS + kq * 2];
          *reinterpret_cast<uint2*>(row) = sp.hi;
          *reinterpret_cast<uint2*>(row + 16) = sp.mid;
          *reinterpret_cast<uint2*>(row + 32) = sp.lo;
        }
      }
    }
  };

  gload(0);
  sstore();
  __syncthreads();

  const int fr = lane & 15;  // fragment row (A) / column (W) inside a 16x16 tile
  const int fg = lane >> 4;  // k group
  for (int t = 0; t < nT; ++t) {
    if (t + 1 < nT) gload(t + 1);      // tile t+1 is in flight while tile t (in LDS) is multiplied
    if constexpr (MATH == 0) {
#pragma unroll
      for (int c = 0; c < kBK / 16; ++c) {
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[mt] = *reinterpret_cast<const f32x4*>(&As[(wave * 16 * MT + mt * 16 + fr) * kLS + c * 16 + fg * 4]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(&Ws[(nt * 16 + fr) * kLS + c * 16 + fg * 4]);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s], b[s], acc[mt][nt], 0, 0, 0);
          }
        }
      }
    } else {
      // one 16x16x32 MFMA spans the whole 32-wide k tile: lane (fr, fg) supplies k = 8*fg..8*fg+7
      bf16x8 a[MT][3];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[mt][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
              &As[(wave * 16 * MT + mt * 16 + fr) * kLS + pl * 16 + fg * 4]));
      // the six plane products of one accumulator form a dependent chain: keep two accumulators
      // in flight (both row tiles, or two column tiles when MT == 1)
      constexpr int NG = (MT == 1) ? 2 : 1;
#pragma unroll
      for (int nt0 = 0; nt0 < NT; nt0 += NG) {
        bf16x8 b[NG][3];
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[q][pl] = (nt0 + q < NT) ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
                                            &Ws[((nt0 + q) * 16 + fr) * kLS + pl * 16 + fg * 4]))
                                      : a[0][pl];
        // (A plane, B plane) pairs, smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
        constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < NG; ++q)
              if (nt0 + q < NT)
                acc[mt][nt0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][PA[p]], b[q][PB[p]],
                                                                           acc[mt][nt0 + q], 0, 0, 0);
      }
    }
    __syncthreads();
    if (t + 1 < nT) {
      sstore();
      __syncthreads();
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  // The MFMA C layout (column = lane & 15, row = (lane >> 4) * 4 + reg) would give 64-byte global
  // pieces.  Each wave instead transposes one 16-row tile at a time through its own LDS region
  // (the A/W tiles are dead: every wave has passed the loop's last barrier) and finishes row by
  // row: lane l owns columns 4l..4l+3, so bias/add/ReLU/score and the store are full-row,
  // 16 bytes per lane, coalesced.  One wave's LDS operations complete in order, so no barrier.
  float* stg = smem + wave * (8 * LD);
  const int colv = n0 + 4 * lane;
  const bool lane_cols = lane < NT * 4 && colv < g.Nout;
  const bool v4out = g.v4out != 0;
  f32x4 bias4 = zero4, ws4 = zero4;
  if (lane_cols) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (colv + e < g.Nout) {
        if (g.bias) bias4[e] = g.bias[colv + e];
        if (EPI == EPI_UPDATE) ws4[e] = g.w_s[colv + e];
      }
    }
  }
  const float bs = (EPI == EPI_UPDATE) ? g.b_s[0] : 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row0 = m0 + wave * 16 * MT + mt * 16;
    float mk = 0.f;                                // mask of row row0 + lane (lanes 0..15)
    if (EPI == EPI_UPDATE && lane < 16 && row0 + lane < g.M) mk = g.mask[row0 + lane];
#pragma unroll
    for (int rb = 0; rb < 16; rb += 8) {
      // rows rb..rb+7 of the tile live in the lanes with (lane >> 5) == rb / 8
      if ((fg >> 1) == (rb >> 3)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[((fg & 1) * 4 + r) * LD + nt * 16 + fr] = acc[mt][nt][r];
      }
      // the 8 rows' `add` operands are requested together, ahead of their use
      f32x4 addv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = row0 + rb + u;
        addv[u] = zero4;
        if (v4out && lane_cols && g.add && row < g.add_rows && row < g.M)
          addv[u] = *reinterpret_cast<const f32x4*>(g.add + (size_t)row * g.Nout + colv);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = rb + u;
        const int row = row0 + rr;
        if (row < g.M) {                             // wave-uniform
          float part = 0.f;
          if (lane_cols) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&stg[u * LD + 4 * lane]) + bias4;
            float* crow = g.C + (size_t)row * g.Nout + colv;
            if (v4out) {
              v += addv[u];
              if (EPI == EPI_UPDATE || g.relu) v = __builtin_elementwise_max(v, zero4);
              *reinterpret_cast<f32x4*>(crow) = v;
            } else {
              const bool has_add = g.add && row < g.add_rows;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (colv + e < g.Nout) {
                  float x = v[e];
                  if (has_add) x += g.add[(size_t)row * g.Nout + colv + e];
                  if (EPI == EPI_UPDATE || g.relu) x = fmaxf(x, 0.f);
                  v[e] = x;
                  crow[e] = x;
                } else {
                  v[e] = 0.f;
                }
              }
            }
            if (EPI == EPI_UPDATE) part = v[0] * ws4[0] + v[1] * ws4[1] + v[2] * ws4[2] + v[3] * ws4[3];
          }
          if constexpr (EPI == EPI_UPDATE) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            const float mrow = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mk), rr));
            if (lane == 0) {
              // fp32 on purpose: score - 1e11 rounds to exactly -1e11, as in the reference
              g.score[row] = (part + bs) + (1.0f - mrow) * kVeryNeg;
            }
          }
        }
      }
    }
  }
}

// ---- cross-lane helpers of the epilogue: VALU only (DPP row operations), no LDS crossbar round trips (the first
// version reduced every row's score with six dependent ds_bpermute shuffles: 96 LDS round trips per wave and tile)
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_f(float old, float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ float lane_xor1(float x) { return dpp_f<0xB1, 0xf>(x, x); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float x) { return dpp_f<0x4E, 0xf>(x, x); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor4(float x) {
  const float t = dpp_f<0x104, 0x5>(x, x);      // row_shl:4 into lanes 0-3, 8-11 of a 16-lane row (they read lane + 4)
  return dpp_f<0x114, 0xA>(t, x);               // row_shr:4 into lanes 4-7, 12-15 (they read lane - 4)
}
__device__ __forceinline__ float lane_xor8(float x) { return dpp_f<0x128, 0xf>(x, x); }  // row_ror:8

// Sums 4 values per lane over the 16 lanes of a DPP row with 6 DPP moves (instead of 4 x 4 shuffles): two
// reduce-scatter steps over lane bits 0 and 1 halve the values a lane carries, two all-reduce steps over bits 2
// and 3 finish.  Returns the total of value index 2*(lane&1) + ((lane>>1)&1); one fixed summation order.
__device__ __forceinline__ float row16_sum4(const float (&v)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float w2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = (b0 ? v[j + 2] : v[j]) + lane_xor1(b0 ? v[j] : v[j + 2]);
  float w = (b1 ? w2[1] : w2[0]) + lane_xor2(b1 ? w2[0] : w2[1]);
  w += lane_xor4(w);
  w += lane_xor8(w);
  return w;
}
__device__ __forceinline__ int row16_sum4_index(int lane) { return 2 * (lane & 1) + ((lane >> 1) & 1); }

// Output-column permutation.  The MFMA C layout gives lane (fr = lane & 15, fg = lane >> 4) the entries
// (row 4*fg + q, column slot fr) of every 16x16 tile nt.  WHICH output column a slot stands for is free: it is
// the W row staged at LDS row nt*16 + fr.  Column tiles are taken in groups of four (64 columns) and slot
// (4a + b, fr) is given column 64a + 4*fr + b, so a lane holds FOUR CONSECUTIVE columns of its rows in the
// registers acc[4a..4a+3][q]: the epilogue loads `add` and stores C straight from the MFMA layout in 16-byte
// pieces, 256 contiguous bytes per 16 lanes - no transposition through LDS.  A trailing tile (NT % 4 == 1)
// keeps the plain order (one column per lane).
template <int NT> struct ColMap {
  static_assert(NT % 4 <= 1, "column tiles come in groups of four plus at most one");
  static constexpr int NFULL = NT / 4;
  static constexpr int NGRP = (NT + 3) / 4;
  // LDS row of W row j (0 <= j < NT*16)
  static __host__ __device__ __forceinline__ int lds_row(int j) {
    if (j < NFULL * 64) {
      const int a = j >> 6, w = j & 63;
      return ((a << 2) + (w & 3)) * 16 + (w >> 2);
    }
    return j;
  }
};

// Opaque copy of a lane-dependent value: address arithmetic derived from it is redone where it is used instead of
// being hoisted out of the row-tile loop and kept in (spilled) registers across the k loop.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// ---- W-resident kernel: the whole weight block lives in LDS, A fragments come straight from global memory ------
// For C[M, Nout] = A[M, K] . W^T with a SHORT K and a huge M (the self-block update: M = B*N = 128 000,
// K = Nout = D = 200) the k-tiled kernel above restages the same 160 KB of W for every 128 rows (as much L2 -> LDS
// traffic as A itself), synchronises its waves twice per 32 k, and its loads run one k tile (a few us) ahead of
// their use while the loaded HBM latency is of that order: its k loop is latency bound, not MFMA bound (phase
// timeline).  Here
//  * one 8-wave workgroup per CU copies W [Nout, K] fp32 into LDS ONCE (160 000 B at D = 200: row stride K
//    floats = 50 float4 chunks, which is bank-conflict free for the ds_read_b128 fragment reads as it is; rows in
//    ColMap order) - afterwards there is no barrier and no LDS write in the kernel;
//  * every wave owns a contiguous run of 16-row tiles of A (M/16 tiles over 2048 waves: 3.9 each at C2) and
//    reads a tile's A fragments in the MFMA layout directly from global memory (lane (fr, fg) loads the float4
//    A[row fr][16c + 4fg ..]: 64-byte pieces, consecutive c complete the 128-byte lines), a WHOLE TILE AHEAD:
//    the fragment registers of k group c are refilled with the next tile's data as soon as group c has been
//    multiplied, i.e. ~10 us before their next use.  The epilogue's `add` rows are requested at the start of the
//    tile.  256 VGPRs per lane (2 waves per SIMD) hold accumulators (52), fragments (52) and add rows (52);
//  * the two waves of a SIMD are independent, so one's epilogue overlaps the other's MFMAs by itself.
// Exact fp32 (v_mfma_f32_16x16x4_f32), alternating accumulators.
// KGUARD: K <= 16 * (NC - 1), i.e. k groups in front of the last one reach beyond K as well (hidden sizes between the
// compiled NC values: 32, 100, 160 ...): every group then clamps its A / W addresses and zeroes the A values beyond K.
// LR = LDS rows of the W block: the ColMap order scatters the W rows of a partly filled 64-column group over up to 64
// LDS rows (Nout = 56: rows up to 61), so the block is sized by the largest row in use, not by Nout.
template <int NT, int NC, int EPI, bool HAS_ADD, bool KGUARD, bool AFL>
__global__ __launch_bounds__(512, 2) void k_gemm_wres(GemmArgs g, int S, int LR) {
  typedef ColMap<NT> CM;
  extern __shared__ __attribute__((aligned(16))) float Wl[];     // [Nout rows in ColMap order][S float4 chunks]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int K = g.K, KC = K >> 2, Nout = g.Nout;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // W -> LDS (float4 copies; row j lands at LDS row ColMap::lds_row(j)).  Twenty requests per thread are in flight
  // before the first one is written (a one-at-a-time copy of 160 KB cost ~25 us of exposed L2 latency per launch).
  {
    const int total = Nout * KC;
    constexpr int UN = 20;
    for (int base = 0; base < total; base += 512 * UN) {
      f32x4 v[UN];
      int dst[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * 512 + tid;
        const int idc = idx < total ? idx : total - 1;
        const int j = idc / KC, kc = idc - j * KC;
        v[u] = *reinterpret_cast<const f32x4*>(g.W + (size_t)j * g.ldw + g.wc0 + 4 * kc);
        dst[u] = idx < total ? (CM::lds_row(j) * S + kc) * 4 : -1;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (dst[u] >= 0) *reinterpret_cast<f32x4*>(Wl + dst[u]) = v[u];
    }
  }
  // bias and score weights behind W (read with ds_read in the epilogue: global loads there would have to wait for
  // the fragment refills issued just before them)
  float* Bl = Wl + (size_t)LR * S * 4;
  float* Sl = Bl + Nout;
  for (int j = tid; j < Nout; j += 512) {
    Bl[j] = g.bias ? g.bias[j] : 0.f;
    Sl[j] = (EPI == EPI_UPDATE) ? g.w_s[j] : 0.f;
  }

  // this wave's 16-row tiles: a contiguous run (a workgroup-local ticket that hands tiles to whichever wave is free
  // was measured and is slower: 126 vs 116 us, the last tiles then start late and run alone)
  const long long U = ((long long)g.M + 15) >> 4;
  const long long GW = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
  int t = (int)(U * gw / GW);
  const int tend = (int)(U * (gw + 1) / GW);
  // per k group c: this lane's float4 offset inside a row of A / of W (zero beyond K: the A value is zeroed, the W
  // address is clamped to a valid chunk so that 0 * finite = 0)
  // (raw load: the k >= K zeroing happens where the fragment is USED, so that a refill does not wait for its data)
  // Without KGUARD only the last k group can reach beyond K (K > 16 * (NC - 1)).
  auto a_frag = [&](int tile, int c) -> f32x4 {
    const int row = min(tile * 16 + fr, g.M - 1);                 // rows beyond M: valid memory, results discarded
    const float* rowp = g.A0 + (size_t)row * g.K0 + 4 * fg;
    if (!KGUARD && c < NC - 1) return *reinterpret_cast<const f32x4*>(rowp + 16 * c);
    return *reinterpret_cast<const f32x4*>(rowp + (min(16 * c + 4 * fg, K - 4) - 4 * fg));
  };
  f32x4 ra[NC];
  if (t < tend) {
#pragma unroll
    for (int c = 0; c < NC; ++c) ra[c] = a_frag(t, c);
  }
  // LDS fragment addresses: row (nt*16 + fr) * S + chunk; rows beyond Nout are clamped (their columns are never stored)
  int wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int a = nt >> 2, b = nt & 3;
    const int col = a < CM::NFULL ? 64 * a + 4 * fr + b : 64 * CM::NFULL + (nt - 4 * CM::NFULL) * 16 + fr;
    wrow[nt] = (col < Nout ? nt * 16 + fr : 0) * S;
  }
  auto col_of = [&](int a) { return a < CM::NFULL ? 64 * a + 4 * fr : 64 * CM::NFULL + fr; };
  const float bs = (EPI == EPI_UPDATE) ? g.b_s[0] : 0.f;
  int tnext_c = 0;
  // AFL: the four row gates of a lane's `add` rows (one dword), requested a tile ahead like the A fragments
  unsigned fl_next = 0x01010101u;
  if (AFL && t < tend) fl_next = *reinterpret_cast<const unsigned*>(g.add_flag + (size_t)t * 16 + 4 * fg);
  __syncthreads();

  for (; t < tend; t = tnext_c) {
    const int rbase = t * 16 + 4 * fg;                             // C layout: this lane's rows rbase + q
    const unsigned fl = fl_next;
    if (AFL) fl_next = *reinterpret_cast<const unsigned*>(g.add_flag + (size_t)(t + 1 < tend ? t + 1 : t) * 16 + 4 * fg);
    // the epilogue's addend rows, in the MFMA / ColMap layout
    f32x4 addv[CM::NGRP][4];
#pragma unroll
    for (int a = 0; a < CM::NGRP; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) addv[a][q] = zero4;
    if constexpr (HAS_ADD) {       // compile time: a branch here would make the compiler's vmcnt waits conservative
#pragma unroll
      for (int a = 0; a < CM::NGRP; ++a) {
        const int col = col_of(a);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int row = min(rbase + q, g.M - 1);
          if (AFL && ((fl >> (8 * q)) & 0xffu) == 0u) row = g.M;      // not a frontier row: the zero row
          const float* arow = g.add + (size_t)row * Nout;
          if (a < CM::NFULL) addv[a][q] = *reinterpret_cast<const f32x4*>(arow + min(col, Nout - 4));
          else addv[a][q][0] = arow[min(col, Nout - 1)];
        }
      }
    }
    float mrow = 0.f;                          // mask of the row whose score this lane will write
    if constexpr (EPI == EPI_UPDATE) mrow = g.mask[min(rbase + row16_sum4_index(fr), g.M - 1)];
    __builtin_amdgcn_sched_barrier(0);        // keep the requests up here: the scheduler would sink them to their use
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero4;
    const int tnext = t + 1;
    tnext_c = tnext;
    const int tload = tnext < tend ? tnext : t;      // (no next tile: the current one is simply requested again -
                                                     // no branch, so that the compiler's vmcnt bookkeeping stays exact)
    // W never changes after the first barrier, so the compiler would hoist all NT*NC fragment reads out of the
    // tile loop (676 registers): the lane's chunk offset is made opaque once per tile
    const int fg_t = opaque(fg);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 a = ra[c];
      int kc = 4 * c + fg_t;
      if (KGUARD || c == NC - 1) {
        if (16 * c + 4 * fg_t >= K) a = zero4;
        kc = min(kc, KC - 1);
      }
#pragma unroll
      for (int nt = 0; nt < NT; nt += 2) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Wl + (size_t)(wrow[nt] + kc) * 4);
        f32x4 b1 = zero4;
        if (nt + 1 < NT) b1 = *reinterpret_cast<const f32x4*>(Wl + (size_t)(wrow[nt + 1 < NT ? nt + 1 : nt] + kc) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc[nt], 0, 0, 0);
          if (nt + 1 < NT) acc[nt + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc[nt + 1], 0, 0, 0);
        }
      }
      ra[c] = a_frag(tload, c);                                    // refill: the next tile's k group c
    }
    // epilogue from the registers: 4 consecutive columns per lane and column group
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 ws4[CM::NGRP], bias4[CM::NGRP];        // this lane's columns of bias / score weights (from LDS)
#pragma unroll
    for (int a = 0; a < CM::NGRP; ++a) {
      const int col = col_of(a);
      ws4[a] = zero4;
      bias4[a] = zero4;
      if (a < CM::NFULL) {
        if (col < Nout) {                           // Nout % 4 == 0: the four columns are valid together
          ws4[a] = *reinterpret_cast<const f32x4*>(Sl + col);
          bias4[a] = *reinterpret_cast<const f32x4*>(Bl + col);
        }
      } else if (col < Nout) {
        ws4[a][0] = Sl[col];
        bias4[a][0] = Bl[col];
      }
    }
#pragma unroll
    for (int a = 0; a < CM::NGRP; ++a) {
      const int col = col_of(a);
      const int nb = a < CM::NFULL ? 4 : 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rbase + q;
        f32x4 v = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * a + e < NT) v[e] = acc[4 * a + e][q];
        v = (v + bias4[a]) + addv[a][q];
        if (EPI == EPI_UPDATE || g.relu) v = __builtin_elementwise_max(v, zero4);
        if (row < g.M) {
          float* crow = g.C + (size_t)row * Nout + col;
          if (nb == 4) {
            if (col < Nout) *reinterpret_cast<f32x4*>(crow) = v;
          } else if (col < Nout) {
            crow[0] = v[0];
          }
        }
        if (EPI == EPI_UPDATE)
          part[q] += v[0] * ws4[a][0] + v[1] * ws4[a][1] + v[2] * ws4[a][2] + v[3] * ws4[a][3];
      }
    }
    if constexpr (EPI == EPI_UPDATE) {
      const float tot = row16_sum4(part, lane);
      const int srow = rbase + row16_sum4_index(fr);
      // fp32 on purpose: score - 1e11 rounds to exactly -1e11, as in the reference
      if (fr < 4 && srow < g.M) g.score[srow] = (tot + bs) + (1.0f - mrow) * kVeryNeg;
    }
  }
}

// applicability of k_gemm_wres and its LDS row stride (float4 chunks): S >= K/4 with S % 4 == 2 keeps the
// ds_read_b128 fragment reads bank-conflict free; the whole block must fit one CU's LDS
// LDS rows the W block of an Nout-column problem occupies in ColMap order (largest row in use + 1)
template <int NT> static int wres_rows_nt(int Nout) {
  int rows = 0;
  for (int j = Nout > 64 ? Nout - 64 : 0; j < Nout; ++j) {
    const int r = ColMap<NT>::lds_row(j) + 1;
    rows = r > rows ? r : rows;
  }
  return rows;
}
static int wres_rows(int Nout, int K) {      // same (NT, NC) classes as launch_wres
  const int nt = (Nout + 15) / 16, nc = (K + 15) / 16, m = nt > nc ? nt : nc;
  return m <= 4 ? wres_rows_nt<4>(Nout) : m <= 8 ? wres_rows_nt<8>(Nout) : wres_rows_nt<13>(Nout);
}
static int wres_stride(const GemmArgs& g) {
  if (g.K % 4 || g.Nout % 4 || g.K0 != g.K || g.A1 || g.n0 || g.K < 16 || g.Nout > 208 || g.K > 208) return 0;
  int S = g.K / 4;
  while (S % 4 != 2) ++S;
  if ((size_t)wres_rows(g.Nout, g.K) * S * 16 + (size_t)g.Nout * 8 > 160 * 1024) return 0;
  return S;
}

template <int EPI>
static int launch_wres(const GemmArgs& g, int S, hipStream_t stream) {
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  const int tiles = (g.M + 15) / 16;
  int grid = cus > 0 ? cus : 1;
  if (grid * 8 > tiles) grid = (tiles + 7) / 8;
  const int LR = wres_rows(g.Nout, g.K);
  const size_t lds = (size_t)LR * S * 16 + (size_t)g.Nout * 8;           // W (ColMap rows), bias, score weights
  const int nc = (g.K + 15) / 16, nt = (g.Nout + 15) / 16;
#define GNNRAG_WRES1(NTT, NCC, HA, KG, FL)                                                                      \
  do {                                                                                                          \
    static DeviceMask cap;                                                                                      \
    const int rc_ = raise_lds_cap(k_gemm_wres<NTT, NCC, EPI, HA, KG, FL>, cap);                                 \
    if (rc_) return rc_;                                                                                        \
    hipLaunchKernelGGL((k_gemm_wres<NTT, NCC, EPI, HA, KG, FL>), dim3(grid), dim3(512), lds, stream, g, S, LR); \
  } while (0)
#define GNNRAG_WRES(NTT, NCC)                                                                                   \
  do {                                                                                                          \
    const bool kg = g.K <= 16 * (NCC - 1);                                                                      \
    if (g.add && g.add_flag && kg) GNNRAG_WRES1(NTT, NCC, true, true, true);                                    \
    else if (g.add && g.add_flag) GNNRAG_WRES1(NTT, NCC, true, false, true);                                    \
    else if (g.add && kg) GNNRAG_WRES1(NTT, NCC, true, true, false);                                            \
    else if (g.add) GNNRAG_WRES1(NTT, NCC, true, false, false);                                                 \
    else if (kg) GNNRAG_WRES1(NTT, NCC, false, true, false);                                                    \
    else GNNRAG_WRES1(NTT, NCC, false, false, false);                                                           \
  } while (0)
  if (nt <= 4 && nc <= 4) GNNRAG_WRES(4, 4);
  else if (nt <= 8 && nc <= 8) GNNRAG_WRES(8, 8);
  else if (nt <= 13 && nc <= 13) GNNRAG_WRES(13, 13);
  else return GNNRAG_E_UNSUPPORTED;
#undef GNNRAG_WRES
#undef GNNRAG_WRES1
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

// score for Nout > 208 (column blocks): one wave per row, dot(h', w_s)
__global__ __launch_bounds__(256) void k_score_rows(const float* __restrict__ h, const float* __restrict__ w_s,
                                                    const float* __restrict__ b_s,
                                                    const float* __restrict__ mask, float* __restrict__ score,
                                                    int M, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += h[(size_t)row * D + c] * w_s[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) score[row] = (s + b_s[0]) + (1.0f - mask[row]) * kVeryNeg;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }


// ---- skinny problems (M up to a few thousand rows, e.g. the [R1,D] relation transforms) ------
// The tiled kernel would run them on M/128 workgroups (5 for R1 = 602) and be latency bound.
// Here one wave owns a 16 x 64 output tile and reads its MFMA fragments straight from global
// memory (the operands are L2 resident), so a 602 x 200 problem spreads over ~150 waves.
template <bool V4>
__global__ __launch_bounds__(256) void k_gemm_skinny(GemmArgs g) {
  if (blockIdx.z) {          // second problem of a pair (same W / bias / shapes): one launch for both
    g.A0 = g.A0b;
    g.add = g.add_b;
    g.C = g.C_b;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int c0 = (blockIdx.y * 4 + wave) * 64;
  if (c0 >= g.Nout) return;
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    f32x4 a[2], b[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = k0 + c * 16 + fg * 4;
      a[c] = load_a4<V4, AMODE_PLAIN>(g, m0 + fr, k);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) b[c][nt] = load_w4<V4, AMODE_PLAIN>(g, c0 + nt * 16 + fr, k);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s], b[c][nt][s], acc[nt], 0, 0, 0);
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = c0 + nt * 16 + fr;
    if (col >= g.Nout) continue;
    const float bia = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + fg * 4 + r;
      if (row >= g.M) continue;
      float v = acc[nt][r] + bia;
      if (g.add && row < g.add_rows) v += g.add[(size_t)row * g.Nout + col];
      if (g.relu) v = fmaxf(v, 0.f);
      g.C[(size_t)row * g.Nout + col] = v;
    }
  }
}

// ---- the self-block update of SMALL batches (B*N < 4096: one WebQSP question, BASELINE config 1) ------------------------
// The k-tiled kernel runs such a problem on M/64 workgroups and spends ~13 us on a 12 MFLOP product (2000 x 56 x 56):
// staging, two barriers per 32 k, an LDS-transposed epilogue.  Here one WAVE owns a 16-row tile and ALL output columns
// (NT <= 13 accumulator tiles), reads its A and W fragments straight from global memory (L2 resident), every load of a
// 32-wide k step issued before the step's MFMAs, exact fp32 (v_mfma_f32_16x16x4_f32); the epilogue works in the MFMA C
// layout (a 16-lane row holds 64 contiguous bytes of an output row) and reduces the score with DPP row operations.
// Row gates (add_flag) as in k_gemm_wres: an unflagged row adds no `add`.
template <int NT>
__global__ __launch_bounds__(256) void k_update_skinny(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  const int m0 = tile * 16;
  if (m0 >= g.M) return;
  const int K = g.K, Nout = g.Nout;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = zero4;
  const float* arow = g.A0 + (size_t)min(m0 + fr, g.M - 1) * g.K0;
  const float* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = g.W + (size_t)min(nt * 16 + fr, Nout - 1) * g.ldw + g.wc0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    f32x4 a[2], b[2][NT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = k0 + c * 16 + fg * 4;
      const bool ok = k < K;                              // K % 4 == 0: a float4 is inside or outside as a whole
      a[c] = ok ? *reinterpret_cast<const f32x4*>(arow + k) : zero4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[c][nt] = ok ? *reinterpret_cast<const f32x4*>(wrow[nt] + k) : zero4;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][e], b[c][nt][e], acc[nt], 0, 0, 0);
  }
  // epilogue in the C layout: lane (fr, fg) holds rows m0 + 4 fg + q, column nt * 16 + fr
  const int rbase = m0 + 4 * fg;
  unsigned fl = 0x01010101u;
  if (g.add_flag) fl = *reinterpret_cast<const unsigned*>(g.add_flag + rbase);      // (M + 4 bytes, rbase % 4 == 0)
  float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 16 + fr;
    const bool cok = col < Nout;
    const float bia = (cok && g.bias) ? g.bias[col] : 0.f;
    const float wsc = cok ? g.w_s[col] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rbase + q;
      float v = 0.f;
      if (cok && row < g.M) {
        float ad = 0.f;
        if (g.add && ((fl >> (8 * q)) & 0xffu)) ad = g.add[(size_t)row * Nout + col];
        v = fmaxf((acc[nt][q] + bia) + ad, 0.f);
        g.C[(size_t)row * Nout + col] = v;
      }
      part[q] += v * wsc;
    }
  }
  const float tot = row16_sum4(part, lane);
  const int srow = rbase + row16_sum4_index(fr);
  // fp32 on purpose: score - 1e11 rounds to exactly -1e11, as in the reference
  if (fr < 4 && srow < g.M) g.score[srow] = (tot + g.b_s[0]) + (1.0f - g.mask[srow]) * kVeryNeg;
}

static bool update_skinny_ok(const GemmArgs& g) {
  return g.M < 4096 && !g.A1 && g.K == g.K0 && g.K % 4 == 0 && g.Nout <= 208 && g.ldw % 4 == 0 && g.wc0 % 4 == 0 &&
         aligned16(g.A0) && aligned16(g.W) && (!g.add || g.add_rows >= g.M) &&
         (!g.add_flag || ((uintptr_t)g.add_flag & 3) == 0);
}

static int launch_update_skinny(const GemmArgs& g, hipStream_t stream) {
  const int nt = (g.Nout + 15) / 16;
  const dim3 grid((unsigned)(((g.M + 15) / 16 + 3) / 4));
#define GNNRAG_USK(N) case N: hipLaunchKernelGGL((k_update_skinny<N>), grid, dim3(256), 0, stream, g); break;
  switch (nt) {
    GNNRAG_USK(1) GNNRAG_USK(2) GNNRAG_USK(3) GNNRAG_USK(4) GNNRAG_USK(5) GNNRAG_USK(6) GNNRAG_USK(7) GNNRAG_USK(8)
    GNNRAG_USK(9) GNNRAG_USK(10) GNNRAG_USK(11) GNNRAG_USK(12) GNNRAG_USK(13)
    default: return GNNRAG_E_UNSUPPORTED;
  }
#undef GNNRAG_USK
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

static bool math_ok(int math) { return math == GNNRAG_MATH_FP32 || math == GNNRAG_MATH_BF16X3 || math == GNNRAG_MATH_MIXED; }

template <int EPI, int AMODE>
static int launch_gemm(GemmArgs g, hipStream_t stream, int math) {
  if (g.M <= 0) return 0;
  const bool v4 = (g.K % 4 == 0) && (g.K0 % 4 == 0) && (g.ldw % 4 == 0) && (g.wc0 % 4 == 0) &&
                  aligned16(g.A0) && aligned16(g.W) && (g.A1 == nullptr || aligned16(g.A1)) &&
                  (g.C_b == nullptr || aligned16(g.A0b)) &&
                  (AMODE != AMODE_GEN || g.gen_D % 4 == 0);
  if (EPI == EPI_LINEAR && AMODE == AMODE_PLAIN && g.M <= kSkinnyMaxM && g.n0 == 0) {
    const dim3 grid((g.M + 15) / 16, (g.Nout + 255) / 256, g.C_b ? 2 : 1);
    if (v4) hipLaunchKernelGGL((k_gemm_skinny<true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((k_gemm_skinny<false>), grid, dim3(256), 0, stream, g);
    GNNRAG_LAUNCH_CHECK();
    return 1 << 30;   // "all columns done" marker for the column-block loop of the caller
  }
  const int ny = AMODE == AMODE_GEN ? 2 : 1;
  // 128-row tiles unless they would leave the chip badly quantised (2 workgroups per CU = 512 slots):
  // mid-size problems (a few hundred tiles) run as twice as many 64-row tiles
  const int tiles128 = ((g.M + 127) / 128) * ny;
  const bool small_tiles = tiles128 < 1024;
  // one 16-row tile per wave: 8-wave workgroups (128 rows, 4 waves/SIMD resident) when that still
  // fills the chip evenly, else 4-wave workgroups (64 rows)
  const int slots8 = 512;
  const int t8 = tiles128;
  const bool nw8 = small_tiles && (t8 >= 3 * slots8 / 2 || (t8 % slots8 == 0) || (t8 % slots8) > slots8 / 2) &&
                   GNNRAG_GEMM_MT1_NW == 8;
  const int bm = small_tiles ? (nw8 ? 128 : 64) : 128;
  const dim3 grid((g.M + bm - 1) / bm, ny);
  const int ncol = g.Nout - g.n0;
  const bool b3 = math != GNNRAG_MATH_FP32;       // MIXED: the k-tiled kernel runs its bf16x3 form
  g.v4out = (g.Nout % 4 == 0) && (g.n0 % 4 == 0) && aligned16(g.C) && (g.add == nullptr || aligned16(g.add));
#define GNNRAG_GEMM_LAUNCH(NT, MT, V, MATH, NW) \
  hipLaunchKernelGGL((k_gemm_f32<NT, MT, V, EPI, AMODE, MATH, NW>), grid, dim3(64 * NW), 0, stream, g)
#define GNNRAG_GEMM_CASE(NT)                                                        \
  do {                                                                              \
    if (small_tiles && nw8) {                                                       \
      if (v4 && b3) GNNRAG_GEMM_LAUNCH(NT, 1, true, 1, 8);                          \
      else if (v4) GNNRAG_GEMM_LAUNCH(NT, 1, true, 0, 8);                           \
      else GNNRAG_GEMM_LAUNCH(NT, 1, false, 0, 8);                                  \
    } else if (small_tiles) {                                                       \
      if (v4 && b3) GNNRAG_GEMM_LAUNCH(NT, 1, true, 1, 4);                          \
      else if (v4) GNNRAG_GEMM_LAUNCH(NT, 1, true, 0, 4);                           \
      else GNNRAG_GEMM_LAUNCH(NT, 1, false, 0, 4);                                  \
    } else {                                                                        \
      if (v4 && b3) GNNRAG_GEMM_LAUNCH(NT, 2, true, 1, 4);                          \
      else if (v4) GNNRAG_GEMM_LAUNCH(NT, 2, true, 0, 4);                           \
      else GNNRAG_GEMM_LAUNCH(NT, 2, false, 0, 4);                                  \
    }                                                                               \
  } while (0)
  if (ncol <= 64) GNNRAG_GEMM_CASE(4);
  else if (ncol <= 128) GNNRAG_GEMM_CASE(8);
  else GNNRAG_GEMM_CASE(13);
#undef GNNRAG_GEMM_CASE
#undef GNNRAG_GEMM_LAUNCH
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag

using namespace gnnrag;


extern "C" int gnnrag_linear(const float* A, int64_t M, int32_t K, const float* W, const float* bias,
                             const float* add, int64_t add_rows, int relu, float* C, int32_t Nout,
                             int32_t math, gnnrag_stream_t stream) {
  if (!A || !W || !C || M < 0 || K <= 0 || Nout <= 0 || !math_ok(math)) return GNNRAG_E_BADARG;
  if (M >= ((int64_t)1 << 31)) return GNNRAG_E_UNSUPPORTED;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A0 = A; g.W = W; g.bias = bias; g.add = add; g.C = C;
  g.M = (int32_t)M; g.K = K; g.K0 = K; g.Nout = Nout; g.ldw = K;
  g.add_rows = add ? (int32_t)(add_rows < M ? add_rows : M) : 0;
  g.relu = relu;
  for (int n0 = 0; n0 < Nout; n0 += 208) {
    g.n0 = n0;
    const int rc = launch_gemm<EPI_LINEAR, AMODE_PLAIN>(g, (hipStream_t)stream, math);
    if (rc == (1 << 30)) break;
    if (rc) return rc;
  }
  return 0;
}

// shared by the two update entry points: h' = relu(A.W^T + b (+ add)), score = score_func(h') + mask term
static int update_common(GemmArgs g, int64_t BN, int32_t D, hipStream_t stream, int math, bool score_zeroed = false) {
  if (D <= 208) {
    g.n0 = 0;
    // short K, exact fp32, aligned operands: the W-resident kernel (whole weight block in LDS)
    const bool al = aligned16(g.A0) && aligned16(g.W) && aligned16(g.C) && (!g.add || aligned16(g.add)) &&
                    g.ldw % 4 == 0 && g.wc0 % 4 == 0 && (!g.add || g.add_rows >= g.M);
    if (GNNRAG_UPDATE_B3 && math != GNNRAG_MATH_FP32 && al && g.add && !g.A1 && g.K == D &&
        (!g.add_flag || ((uintptr_t)g.add_flag & 3) == 0)) {
      const int rc = update_b3_launch_f(g.A0, g.add, g.add_flag, g.W, g.bias, g.w_s, g.b_s, g.mask, g.C, g.score, BN, D,
                                        g.ldw, stream, score_zeroed);
      if (rc != GNNRAG_E_UNSUPPORTED) return rc;
    }
    const int S = (GNNRAG_GEMM_WRES && math != GNNRAG_MATH_BF16X3 && al && g.M >= 4096 &&
                   (!g.add_flag || ((uintptr_t)g.add_flag & 3) == 0)) ? wres_stride(g) : 0;
    if (S) return launch_wres<EPI_UPDATE>(g, S, stream);
    if (GNNRAG_UPDATE_SKINNY && update_skinny_ok(g)) return launch_update_skinny(g, stream);    // small batches: exact fp32
    if (g.add_flag) return GNNRAG_E_UNSUPPORTED;      // the k-tiled kernel has no row-gated form: nothing launched
    return launch_gemm<EPI_UPDATE, AMODE_PLAIN>(g, stream, math);
  }
  if (g.add_flag) return GNNRAG_E_UNSUPPORTED;
  // wide hidden sizes: column blocks of 208 with bias(+add)+ReLU epilogue, then a row-dot for the score
  g.relu = 1;
  for (int n0 = 0; n0 < D; n0 += 208) {
    g.n0 = n0;
    const int rc = launch_gemm<EPI_LINEAR, AMODE_PLAIN>(g, stream, math);
    if (rc == (1 << 30)) break;
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_score_rows, dim3((int)((BN + 3) / 4)), dim3(256), 0, stream, g.C, g.w_s, g.b_s, g.mask,
                     g.score, (int)BN, D);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_linear_pair(const float* A0, const float* A1, int64_t M, int32_t K, const float* W,
                                  const float* bias, const float* add0, const float* add1, int64_t add_rows,
                                  float* C0, float* C1, int32_t Nout, int32_t math, gnnrag_stream_t stream) {
  if (!A0 || !A1 || !W || !C0 || !C1 || M < 0 || K <= 0 || Nout <= 0 || !math_ok(math)) return GNNRAG_E_BADARG;
  if ((add0 == nullptr) != (add1 == nullptr)) return GNNRAG_E_BADARG;
  if (M > kSkinnyMaxM) {     // large problems: two ordinary launches
    int rc = gnnrag_linear(A0, M, K, W, bias, add0, add_rows, 0, C0, Nout, math, stream);
    if (rc) return rc;
    return gnnrag_linear(A1, M, K, W, bias, add1, add_rows, 0, C1, Nout, math, stream);
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A0 = A0; g.A0b = A1; g.W = W; g.bias = bias; g.add = add0; g.add_b = add1; g.C = C0; g.C_b = C1;
  g.M = (int32_t)M; g.K = K; g.K0 = K; g.Nout = Nout; g.ldw = K;
  g.add_rows = add0 ? (int32_t)(add_rows < M ? add_rows : M) : 0;
  const int rc = launch_gemm<EPI_LINEAR, AMODE_PLAIN>(g, (hipStream_t)stream, math);
  return rc == (1 << 30) ? 0 : rc;
}

extern "C" int gnnrag_update_score(const float* h, const float* agg, const float* W, const float* b,
                                   const float* w_s, const float* b_s, const float* mask, float* h_out,
                                   float* score, int64_t BN, int32_t D, int32_t I, int32_t math,
                                   gnnrag_stream_t stream) {
  if (!h || !agg || !W || !b || !w_s || !b_s || !mask || !h_out || !score || BN < 0 || D <= 0 || I <= 0 ||
      !math_ok(math))
    return GNNRAG_E_BADARG;
  if (BN >= ((int64_t)1 << 31)) return GNNRAG_E_UNSUPPORTED;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A0 = h; g.A1 = agg; g.W = W; g.bias = b; g.C = h_out;
  g.w_s = w_s; g.b_s = b_s; g.mask = mask; g.score = score;
  g.M = (int32_t)BN; g.K = (2 * I + 1) * D; g.K0 = D; g.Nout = D; g.ldw = g.K;
  g.relu = 1;
  return update_common(g, BN, D, (hipStream_t)stream, math);
}

extern "C" int gnnrag_update_score_fused(const float* h, const float* nbr, const float* W, const float* b,
                                         const float* w_s, const float* b_s, const float* mask, float* h_out,
                                         float* score, int64_t BN, int32_t D, int32_t I, int32_t math,
                                         gnnrag_stream_t stream) {
  return gnnrag::update_score_fused_z(h, nbr, W, b, w_s, b_s, mask, h_out, score, BN, D, I, math, (hipStream_t)stream,
                                      false);
}

bool gnnrag::update_rows_supported(const float* h, const float* nbr, const float* W, const float* h_out, int64_t BN,
                                   int32_t D, int32_t I, int32_t math) {
  if (D > 208 || BN >= ((int64_t)1 << 31)) return false;
  const int ldw = (2 * I + 1) * D;
  const bool al = aligned16(h) && aligned16(W) && aligned16(h_out) && aligned16(nbr) && ldw % 4 == 0;
  if (!al) return false;
  if (GNNRAG_UPDATE_B3 && math != GNNRAG_MATH_FP32 && update_b3_shape_ok(BN, D, ldw)) return true;
  if (GNNRAG_UPDATE_SKINNY && BN < 4096 && D % 4 == 0) return true;                 // k_update_skinny
  if (!(GNNRAG_GEMM_WRES && math != GNNRAG_MATH_BF16X3 && BN >= 4096)) return false;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.M = (int32_t)BN; g.K = D; g.K0 = D; g.Nout = D; g.ldw = ldw;
  return wres_stride(g) != 0;
}

int gnnrag::update_score_fused_z(const float* h, const float* nbr, const float* W, const float* b, const float* w_s,
                                 const float* b_s, const float* mask, float* h_out, float* score, int64_t BN, int32_t D,
                                 int32_t I, int32_t math, hipStream_t stream, bool score_zeroed) {
  return update_score_fused_rows(h, nbr, nullptr, W, b, w_s, b_s, mask, h_out, score, BN, D, I, math, stream, score_zeroed);
}

int gnnrag::update_score_fused_rows(const float* h, const float* nbr, const uint8_t* add_flag, const float* W,
                                    const float* b, const float* w_s, const float* b_s, const float* mask, float* h_out,
                                    float* score, int64_t BN, int32_t D, int32_t I, int32_t math, hipStream_t stream,
                                    bool score_zeroed) {
  if (!h || !nbr || !W || !b || !w_s || !b_s || !mask || !h_out || !score || BN < 0 || D <= 0 || I <= 0 ||
      !math_ok(math))
    return GNNRAG_E_BADARG;
  if (BN >= ((int64_t)1 << 31)) return GNNRAG_E_UNSUPPORTED;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  // only the self block W[:, 0:D] of e2e_linear is multiplied here; the neighbour blocks were
  // pushed into the relation tables and arrive already reduced in `nbr`
  g.A0 = h; g.W = W; g.bias = b; g.add = nbr; g.add_rows = (int32_t)BN; g.C = h_out;
  g.w_s = w_s; g.b_s = b_s; g.mask = mask; g.score = score;
  g.M = (int32_t)BN; g.K = D; g.K0 = D; g.Nout = D; g.ldw = (2 * I + 1) * D; g.wc0 = 0;
  g.relu = 1;
  g.add_flag = add_flag;
  return update_common(g, BN, D, stream, math, score_zeroed);
}

extern "C" int gnnrag_relation_tables(const gnnrag_csr* csr, const float* T_fwd, const float* T_inv,
                                      const float* ins, const float* W, float* P, int32_t D, int32_t I,
                                      int32_t math, gnnrag_stream_t stream) {
  if (!csr || !T_fwd || !T_inv || !ins || !W || !P || D <= 0 || I <= 0 || csr->rel_total < 0 || !math_ok(math))
    return GNNRAG_E_BADARG;
  if (csr->rel_total == 0) return 0;      // no facts, no tables
  {   // small batches (one question): one workgroup per 16-row tile, split k, exact fp32 - in every math mode
    const int rc = tables_small_launch(csr, T_fwd, T_inv, ins, W, P, D, I, (hipStream_t)stream);
    if (rc != GNNRAG_E_UNSUPPORTED) return rc;
  }
  if (math == GNNRAG_MATH_MIXED) math = GNNRAG_MATH_BF16X3;
  if (math == GNNRAG_MATH_BF16X3 && GNNRAG_TABLES_WRES) {     // W-resident kernel (tables_b3.hip) where its shapes allow
    const int rc = tables_b3_launch(csr, T_fwd, T_inv, ins, W, P, D, I, (hipStream_t)stream);
    if (rc != GNNRAG_E_UNSUPPORTED) return rc;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A0 = T_fwd;
  g.A0b = T_inv;
  g.A1 = ins;
  g.W = W;
  g.C = P;                       // direction d (blockIdx.y) writes P + d*rel_total*D
  g.M = csr->rel_total; g.K = I * D; g.K0 = g.K; g.Nout = D; g.ldw = (2 * I + 1) * D;
  g.gen_rows = (const int2*)csr->rel_rows; g.gen_D = D; g.gen_I = I;
  for (int n0 = 0; n0 < D; n0 += 208) {
    g.n0 = n0;
    const int rc = launch_gemm<EPI_LINEAR, AMODE_GEN>(g, (hipStream_t)stream, math);
    if (rc) return rc;
  }
  return 0;
}
