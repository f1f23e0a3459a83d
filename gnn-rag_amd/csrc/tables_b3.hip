// The bf16x3 W-resident kernels of the fused path (gfx950):
//   k_tables_b3   relation tables, operand relu(T * ins) generated in registers (a lone gnnrag_relation_tables call)
//   k_tables_vq   relation tables in "V form" from pre-split relation planes (what a layer / stack call runs)
//   k_update_b3   the self-block update h' = relu(h W^T + b + nbr) with the fused score
// They share the LDS weight-plane layout (three bf16 planes of an exact 3-way split, 26-slot row stride), the six
// plane products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, and the register epilogue.
//
// ---- k_tables_b3: per-question relation tables of the fused path, operand generated on the fly ----------------------
//
//   P[d, m, :] = sum_i  W_e2e[:, (1+2i+d)D : (2+2i+d)D] . relu( T_d[r_m, :] * ins[b_m, i, :] )      m = compact row (b_m, r_m)
//
// (reference: the e2e_linear column blocks applied to the per-fact messages of reasongnn.py:71-79 / :98-105, once
// per (question, relation in use) instead of once per fact - see gemm_f32.hip / DESIGN.md section 3.3).
//
// Why a second kernel: on the k-tiled kernel (gemm_f32.hip) this product ran at ~30 % of the bf16 matrix rate at C2
// (127 us): 150 rows per workgroup leave a 22-row remainder tile that costs a full tile, and every 32 k the
// workgroup restages operands and synchronises twice.  The operand here is GENERATED from two small L2-resident
// tables, so nothing has to stream from HBM and the weights can stay put:
//   * workgroup = (direction d, column part h, row chunk): 8 waves, one per CU;
//   * per instruction i the workgroup splits its weight block W_{i,d}[columns of h, :] into three bf16 planes ONCE
//     (exact 3-way split, gnnrag_common.h) and keeps them in LDS (3 x 112 x 416 B = 140 KB at D = 200; row stride
//     26 x 16 B keeps the ds_read_b128 fragment reads bank-conflict free);
//   * every wave owns up to 5 of the chunk's 16-row tiles and keeps ALL their accumulators in registers across the
//     instructions (5 x 7 column tiles x 4 = 140 VGPRs), so P is written once and nothing is re-read;
//   * the A fragment of (tile, i, 32 k) - relu(T * ins), split into planes in registers - is built by the lane
//     that feeds it to the MFMA, from loads issued one k block ahead; six v_mfma_f32_16x16x32_bf16 per (column
//     tile, k block): hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid, smallest terms first, fp32 accumulation.
// Two barriers per instruction and workgroup instead of two per 32 k.
#include "gnnrag_common.h"
#include "dense_internal.h"

namespace gnnrag {

#ifndef GNNRAG_TAB_TPW
#define GNNRAG_TAB_TPW 3
#endif
#ifndef GNNRAG_TAB_MINBLK
#define GNNRAG_TAB_MINBLK 2       // waves per SIMD: one 8-wave workgroup per CU (the planes fill its LDS) = 2, 256 registers each
#endif
constexpr int kTabTPW = GNNRAG_TAB_TPW;        // row tiles per wave and pass (accumulators in registers; 5 spill: 140 + 40 + 36 VGPRs)
constexpr int kTabNTH = 7;        // column tiles per column part (two parts cover up to 13 tiles = 208 columns)
constexpr int kTabNKB = 7;        // k blocks of 32: 192 < D <= 208 (the hidden size of the BASELINE configs is 200)
constexpr int kTabSlots = 26;     // LDS row stride of a weight plane in 16-byte slots (26 % 4 == 2: conflict free)
constexpr int kTabQBytes = 160 * 1024 - 3 * kTabNTH * 16 * kTabSlots * 16 - 64;   // LDS left for instruction rows (24 000 B)

struct TabArgs {
  const float* T[2];       // [R1, D] relation tables of the two directions
  const float* ins;        // [B, I, D]
  const float* W;          // e2e_linear.weight [D, (2I+1) D]
  float* P;                // [2, M, D]
  const int2* rows;        // [M] (question, relation id) of every compact row
  int32_t M, D, I, ldw;
  int32_t ct0;             // column tiles of part 0 (part 1 takes the rest)
  int32_t npass;           // passes of kTabTPW tiles per wave
};

__device__ __forceinline__ int opaque_i(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// sum of 4 values per lane over the 16 lanes of a DPP row (see gemm_f32.hip: row16_sum4); returns the total of value
// index 2*(lane&1) + ((lane>>1)&1)
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_b3(float old, float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ float row16_sum4_b3(const float (&v)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float w2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float send = b0 ? v[j] : v[j + 2];
    w2[j] = (b0 ? v[j + 2] : v[j]) + dpp_b3<0xB1, 0xf>(send, send);
  }
  const float send = b1 ? w2[0] : w2[1];
  float w = (b1 ? w2[1] : w2[0]) + dpp_b3<0x4E, 0xf>(send, send);
  const float t4 = dpp_b3<0x104, 0x5>(w, w);
  w += dpp_b3<0x114, 0xA>(t4, w);
  w += dpp_b3<0x128, 0xf>(w, w);
  return w;
}

// LDS row of column slot: the first four column tiles of a part are interleaved so that a lane holds four
// consecutive columns (float4 stores in the epilogue), the others keep the plain order
__device__ __forceinline__ int tab_lds_row(int j) {
  if (j < 64) return ((j & 3) << 4) + (j >> 2);
  return j;
}

// Weight rows a column part stages per plane: its CTN x 16 columns (columns past the part's last real one as zeros)
// PLUS ONE zero row when the part does not fill the plane - the last k block of a row reads up to 32 bytes past the
// row's end (k >= D; the A side is zero there), i.e. the head of the NEXT row, and 0 x (whatever an unwritten LDS row
// holds) is NaN when that happens to be an Inf / NaN pattern.  With D = 200 the next row is one of the part's own zero
// rows; with D = 208 the 6-tile part ends exactly at its last staged row (found by the D = 208 shape-sweep test:
// column 207 came out as relu(NaN) = 0 for some rows).  A full plane (7 tiles) is followed by the next plane's row 0
// or the zeroed slack.
__device__ __forceinline__ constexpr int tab_stage_rows(int ctn) { return ctn < kTabNTH ? ctn * 16 + 1 : ctn * 16; }

// One column part with CTN column tiles (compile time: the MFMA loop has no branches).  QLDS: the instruction rows
// of the chunk's questions are in LDS (the usual case) - also compile time: a run-time branch around the global-memory
// variant's loads would make every vmcnt wait of the loop conservative, i.e. wait for the prefetch just issued.
template <int CTN, bool QLDS>
__device__ __forceinline__ void tables_b3_part(const TabArgs& a, unsigned char* lds, int col0, int bmin, int nq) {
  constexpr int RB = kTabSlots * 16;                        // bytes per weight row of one plane
  constexpr int PL = kTabNTH * 16 * RB;                     // bytes per plane
  constexpr int ctn = CTN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int d = blockIdx.y;
  const int D = a.D, I = a.I;
  const int ncol = min(ctn * 16, D - col0);                 // valid columns of this part
  const float* T = a.T[d];
  float* P = a.P + (size_t)d * a.M * D;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int NKB = kTabNKB;                              // 32-wide k blocks
  const int KC = D >> 2;                                    // float4 chunks of a weight row

  // this wave's 16-row tiles: the chunk's tiles are dealt out contiguously to the 8 waves
  const long long U = ((long long)a.M + 15) >> 4;
  const int c0 = (int)(U * blockIdx.x / gridDim.x), c1 = (int)(U * (blockIdx.x + 1) / gridDim.x);
  const int nchunk = c1 - c0;
  const int w0 = c0 + (int)((long long)nchunk * wave / 8), w1 = c0 + (int)((long long)nchunk * (wave + 1) / 8);

  if (tid < 16) reinterpret_cast<unsigned*>(lds + 3 * PL)[tid] = 0u;   // slack behind the last plane: finite (k >= D reads)
  // the instruction rows ins[b, :, :] of the questions this chunk's rows belong to (rows are sorted by question):
  // LDS behind the planes when they fit, else they are read from global memory (cache resident)
  float* qarea = reinterpret_cast<float*>(lds + 3 * PL + 64);
  constexpr bool q_in_lds = QLDS;
  if (q_in_lds) {
    const float* src = a.ins + (size_t)bmin * I * D;
    for (int x = tid * 4; x < nq * I * D; x += 512 * 4)
      *reinterpret_cast<f32x4*>(qarea + x) = *reinterpret_cast<const f32x4*>(src + x);
  }

  for (int pass = 0; pass < a.npass; ++pass) {
    const int tbase = w0 + pass * kTabTPW;
    const int ntile = max(0, min(kTabTPW, w1 - tbase));     // wave-uniform
    f32x4 acc[kTabTPW][CTN];
#pragma unroll
    for (int j = 0; j < kTabTPW; ++j)
#pragma unroll
      for (int nt = 0; nt < CTN; ++nt) acc[j][nt] = zero4;

    for (int i = 0; i < I; ++i) {
      __syncthreads();                                      // the previous block's fragment reads are done
      // ---- this instruction's weight block -> three bf16 planes in LDS ----
      {
        const int wcol = (1 + 2 * i + d) * D;               // first k column of the block inside e2e_linear.weight
        const int total = tab_stage_rows(ctn) * kTabSlots * 2;      // 8-byte pieces (4 k) per plane: rows x 52
        constexpr int UN = 6;                               // requests in flight per thread before the first split
        for (int base = 0; base < total; base += 512 * UN) {
          f32x4 v[UN];
          int off[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int idx = base + u * 512 + tid;
            const int j = idx / (kTabSlots * 2), kc = idx - j * (kTabSlots * 2);
            v[u] = zero4;
            off[u] = idx < total ? tab_lds_row(j) * RB + kc * 8 : -1;
            if (idx < total && j < ncol && kc < KC)
              v[u] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(col0 + j) * a.ldw + wcol + 4 * kc);
          }
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            if (off[u] >= 0) {
              const Split3 sp = split3(v[u]);
              unsigned char* dst = lds + off[u];
              *reinterpret_cast<uint2*>(dst) = sp.hi;
              *reinterpret_cast<uint2*>(dst + PL) = sp.mid;
              *reinterpret_cast<uint2*>(dst + 2 * PL) = sp.lo;
            }
          }
        }
      }
      __syncthreads();

      // ---- MFMA phase: k blocks outermost, the wave's tiles inside: a tile's T row piece for k block kb + 1 is
      // requested right after the piece for kb has been consumed, i.e. a whole round of tiles (kTabTPW x 42 MFMAs)
      // before its use; the instruction rows come from LDS ----
      int toff[kTabTPW], qoff[kTabTPW];                     // per tile: this lane's row offsets (floats) into T / the q area
#pragma unroll
      for (int j = 0; j < kTabTPW; ++j) {
        const int m = min((tbase + (j < ntile ? j : 0)) * 16 + fr, a.M - 1);
        const int2 br = a.rows[m];
        toff[j] = br.y * D + 8 * fg;
        qoff[j] = ((br.x - bmin) * I + i) * D + 8 * fg;
      }
      const int kmax = D - 8;                               // last valid 8-element start of a row
      f32x4 traw[kTabTPW][2];
#pragma unroll
      for (int j = 0; j < kTabTPW; ++j) {
        const float* tp = T + toff[j] + (min(8 * fg, kmax) - 8 * fg);
        traw[j][0] = *reinterpret_cast<const f32x4*>(tp);
        traw[j][1] = *reinterpret_cast<const f32x4*>(tp + 4);
      }
      for (int kb = 0; kb < NKB; ++kb) {      // (not unrolled: the unrolled loop spills 200 registers)
        const bool kok = 32 * kb + 8 * fg <= kmax;
        const int kbn = min(kb + 1, NKB - 1);               // (the last block requests itself again: no branch)
        const int kon = min(32 * kbn + 8 * fg, kmax) - 8 * fg;
        const int koq = min(32 * kb + 8 * fg, kmax) - 8 * fg;
        const unsigned char* wb = lds + fr * RB + kb * 64 + fg * 16;
#pragma unroll
        for (int j = 0; j < kTabTPW; ++j) {
          if (j < ntile) {                                  // wave-uniform
            const f32x4 t0 = traw[j][0], t1 = traw[j][1];
            {
              const float* tp = T + toff[j] + kon;
              traw[j][0] = *reinterpret_cast<const f32x4*>(tp);
              traw[j][1] = *reinterpret_cast<const f32x4*>(tp + 4);
            }
            __builtin_amdgcn_sched_barrier(0);              // the requests stay ABOVE the MFMAs
            f32x4 q0, q1;
            if constexpr (QLDS) {
              const float* qp = qarea + qoff[j] + koq;
              q0 = *reinterpret_cast<const f32x4*>(qp);
              q1 = *reinterpret_cast<const f32x4*>(qp + 4);
            } else {
              const float* qp = a.ins + (size_t)bmin * I * D + qoff[j] + koq;
              q0 = *reinterpret_cast<const f32x4*>(qp);
              q1 = *reinterpret_cast<const f32x4*>(qp + 4);
            }
            // A planes of this k block: relu(T * ins), exact 3-way bf16 split
            const Split3 s0 = split3(kok ? __builtin_elementwise_max(t0 * q0, zero4) : zero4);
            const Split3 s1 = split3(kok ? __builtin_elementwise_max(t1 * q1, zero4) : zero4);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            bf16x8 ap[3];
            ap[0] = __builtin_bit_cast(bf16x8, (u32x4){s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y});
            ap[1] = __builtin_bit_cast(bf16x8, (u32x4){s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y});
            ap[2] = __builtin_bit_cast(bf16x8, (u32x4){s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y});
            // (A plane, B plane) pairs, smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
            constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
            constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int nt = 0; nt < CTN; nt += 2) {
              bf16x8 b0[3], b1[3];
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) {
                b0[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb + pl * PL + nt * 16 * RB));
                b1[pl] = b0[pl];
                if (nt + 1 < CTN)
                  b1[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb + pl * PL + (nt + 1) * 16 * RB));
              }
#pragma unroll
              for (int p = 0; p < 6; ++p) {
                acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PA[p]], b0[PB[p]], acc[j][nt], 0, 0, 0);
                if (nt + 1 < CTN)
                  acc[j][nt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PA[p]], b1[PB[p]], acc[j][nt + 1], 0, 0, 0);
              }
            }
          }
        }
      }
    }

    // ---- epilogue: the pass's tiles leave the registers (C layout: rows 4 fg + q, column slot fr) ----
#pragma unroll
    for (int j = 0; j < kTabTPW; ++j) {
      if (j < ntile) {
        const int rbase = (tbase + j) * 16 + 4 * fg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = rbase + q;
          if (row < a.M) {
            float* prow = P + (size_t)row * D + col0;
            // interleaved group (column tiles 0..3): columns 4 fr .. 4 fr + 3
            static_assert(CTN >= 4, "a column part holds at least the interleaved group");
            {
              const f32x4 v = {acc[j][0][q], acc[j][1][q], acc[j][2][q], acc[j][3][q]};
              const int c = 4 * fr;
              if (c + 4 <= ncol) *reinterpret_cast<f32x4*>(prow + c) = v;
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (c + e < ncol) prow[c + e] = v[e];
              }
            }
#pragma unroll
            for (int nt = 4; nt < CTN; ++nt) {
              const int c = nt * 16 + fr;
              if (c < ncol) prow[c] = acc[j][nt][q];
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(512, GNNRAG_TAB_MINBLK) void k_tables_b3(TabArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int NT = (a.D + 15) >> 4;
  const int h = blockIdx.z;
  const int ctn = h == 0 ? a.ct0 : NT - a.ct0;              // column tiles of this part
  const int col0 = h == 0 ? 0 : a.ct0 * 16;                 // first column of this part
  // questions this chunk's rows belong to (rows are sorted by question)
  const long long U = ((long long)a.M + 15) >> 4;
  const int c0 = (int)(U * blockIdx.x / gridDim.x), c1 = (int)(U * (blockIdx.x + 1) / gridDim.x);
  const int rlo = min(c0 * 16, a.M - 1), rhi = min(c1 * 16, a.M) - 1;
  const int bmin = a.rows[rlo].x, bmax = a.rows[max(rhi, rlo)].x;
  const int nq = bmax - bmin + 1;
  const bool q_in_lds = (size_t)nq * a.I * a.D * sizeof(float) <= kTabQBytes;
#define GNNRAG_TAB_CASE(N)                                                  \
  case N:                                                                   \
    if (q_in_lds) tables_b3_part<N, true>(a, lds, col0, bmin, nq);          \
    else tables_b3_part<N, false>(a, lds, col0, bmin, nq);                  \
    break;
  switch (ctn) {
    GNNRAG_TAB_CASE(6) GNNRAG_TAB_CASE(7)
    default: break;
  }
#undef GNNRAG_TAB_CASE
}

// ---- relation tables from PRE-SPLIT relation planes and per-question weights ("V form") -------------------------------
// relu(t * q) = max(q, 0) * relu(t) + max(-q, 0) * relu(-t) for every real t, q - so with Tp = relu(T_d), Tn = relu(-T_d)
//
//   P[d, (b, r), :] = sum_i W_{i,d} . relu(T_d[r, :] * ins[b, i, :])  =  [Tp[r, :], Tn[r, :]] . V_{b,d}
//   V_{b,d}[k, :]     = sum_i W_{i,d}[:, k] * max( ins[b,i,k], 0)          (rows 0 .. D-1,   "half" 0)
//   V_{b,d}[D + k, :] = sum_i W_{i,d}[:, k] * max(-ins[b,i,k], 0)          (rows D .. 2D-1,  "half" 1)
//
// The left operand no longer depends on the question: its three bf16 planes are written ONCE per layer by the relation
// projection kernel (rel_transform.hip: a few hundred rows, L2 resident) and arrive in the MFMA loop as ready 16-byte
// fragments - no multiply / relu / 3-way split beside the MFMAs (the VALU work that held k_tables_b3 at ~1/3 of the
// matrix rate; a packed-fp32 VALU instruction beside MFMAs costs far more than its issue slot on this chip), and the
// k extent is 2 D for every number of instructions I.  The question moves into the RIGHT operand: workgroup =
// (question, row chunk, direction, column part); it builds V's three planes for its column part and one half at a
// time in LDS (I weight blocks x the question's instruction rows; same layout as above), all row tiles of the
// question (<= 5 per wave: one pass at C2) multiply against them with the accumulators held across both halves.
// Same 6 plane products and fp32 accumulation as above; V is rounded to fp32 once per element before its exact split.
constexpr int kVqHalf = 32 * kTabNKB;          // bf16 elements per half row of the relation planes (224: k >= D are zero)
constexpr int kVqRowB = 2 * kVqHalf * 2;       // bytes per plane row: [relu(T) | relu(-T)]
#ifndef GNNRAG_VQ_TPW
#define GNNRAG_VQ_TPW 5
#endif
#ifndef GNNRAG_VQ_UN
#define GNNRAG_VQ_UN 3
#endif
constexpr int kVqTPW = GNNRAG_VQ_TPW;          // row tiles per wave and pass

struct VqArgs {
  const unsigned char* planes;   // [2 directions][3 planes][R1][kVqRowB]
  const float* ins;              // [B, I, D]
  const float* W;                // e2e_linear.weight [D, (2I+1) D]
  float* P;                      // [2, M, D]
  const int2* rows;              // [M] (question, relation id) of every compact row
  const int* rel_off;            // [B+1] first compact row of each question
  int32_t M, D, I, ldw, R1;
  int32_t ct0;                   // column tiles of part 0
  int32_t nchunk;                // row chunks per question (> 1 only when the batch has few questions)
  int32_t dir0;                  // first direction of the launch (grid y counts on from it)
  float* zero;                   // null or a buffer the launch also zeroes (the layer's score, see dense_internal.h)
  int64_t zero_n;
};

// V planes of one half and one column part -> LDS:  V[n, k] = sum_i W[col0 + n, (1 + 2 i + d) D + k] * max(+-ins[g, i, k], 0)
// UN pieces (4 k each) per thread and round; indices are recomputed rather than kept (register pressure)
template <int CTN, int UN>
__device__ __forceinline__ void vq_stage(const VqArgs& a, unsigned char* lds, const float* qarea, int col0, int ncol,
                                         int d, int s) {
  constexpr int RB = kTabSlots * 16;
  constexpr int PL = kTabNTH * 16 * RB;
  const int tid = threadIdx.x;
  const int D = a.D, I = a.I, KC = D >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int total = tab_stage_rows(CTN) * kTabSlots * 2;        // 8-byte pieces (4 k) per plane
  for (int base = 0; base < total; base += 512 * UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = zero4;
    for (int i = 0; i < I; ++i) {
      f32x4 w[UN];
      const float* wsrc = a.W + (size_t)col0 * a.ldw + (1 + 2 * i + d) * D;
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * 512 + tid;
        const int j = idx / (kTabSlots * 2), kc = idx - j * (kTabSlots * 2);
        // dead pieces (k >= D, columns past the part) read a valid address and are zeroed below
        w[u] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)min(j, ncol - 1) * a.ldw + 4 * min(kc, KC - 1));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * 512 + tid;
        const int kc = idx % (kTabSlots * 2);
        f32x4 q = *reinterpret_cast<const f32x4*>(qarea + i * D + 4 * min(kc, KC - 1));
        q = __builtin_elementwise_max(s ? -q : q, zero4);
        v[u] += w[u] * q;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = base + u * 512 + tid;
      const int j = idx / (kTabSlots * 2), kc = idx - j * (kTabSlots * 2);
      if (idx < total) {
        const Split3 sp = split3(j < ncol && kc < KC ? v[u] : zero4);
        unsigned char* dst = lds + tab_lds_row(j) * RB + kc * 8;
        *reinterpret_cast<uint2*>(dst) = sp.hi;
        *reinterpret_cast<uint2*>(dst + PL) = sp.mid;
        *reinterpret_cast<uint2*>(dst + 2 * PL) = sp.lo;
      }
    }
  }
}

template <int CTN>
__device__ __forceinline__ void tables_vq_part(const VqArgs& a, unsigned char* lds, int col0) {
  constexpr int RB = kTabSlots * 16;
  constexpr int PL = kTabNTH * 16 * RB;
  constexpr int NKB = kTabNKB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int d = a.dir0 + blockIdx.y;
  const int g = blockIdx.x / a.nchunk, ch = blockIdx.x - g * a.nchunk;
  const int D = a.D, I = a.I;
  const int r0 = a.rel_off[g], r1 = a.rel_off[g + 1];
  if (r1 <= r0) return;                                       // a question without facts has no rows
  const int ncol = min(CTN * 16, D - col0);
  const int KC = D >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* P = a.P + (size_t)d * a.M * D;
  const unsigned char* planes = a.planes + (size_t)d * 3 * a.R1 * kVqRowB;
  const size_t plane_stride = (size_t)a.R1 * kVqRowB;

  // 16-row tiles of the question start at its first row; the chunk's tiles are dealt out contiguously to the 8 waves
  const int U = (r1 - r0 + 15) >> 4;
  const int c0 = U * ch / a.nchunk, c1 = U * (ch + 1) / a.nchunk;
  const int wv = __builtin_amdgcn_readfirstlane(wave);        // scalar: the tile-count branches below stay uniform
  const int w0 = c0 + (c1 - c0) * wv / 8, w1 = c0 + (c1 - c0) * (wv + 1) / 8;
  const int npass = ((c1 - c0 + 7) / 8 + kVqTPW - 1) / kVqTPW;     // workgroup-uniform (the barriers below)

  if (tid < 16) reinterpret_cast<unsigned*>(lds + 3 * PL)[tid] = 0u;   // slack behind the last plane stays finite
  float* qarea = reinterpret_cast<float*>(lds + 3 * PL + 64);          // ins[g, :, :]
  for (int x = tid * 4; x < I * D; x += 512 * 4)
    *reinterpret_cast<f32x4*>(qarea + x) = *reinterpret_cast<const f32x4*>(a.ins + (size_t)g * I * D + x);

  for (int pass = 0; pass < npass; ++pass) {
    const int tbase = w0 + pass * kVqTPW;
    const int ntile = max(0, min(kVqTPW, w1 - tbase));        // wave-uniform
    f32x4 acc[kVqTPW][CTN];
#pragma unroll
    for (int j = 0; j < kVqTPW; ++j)
#pragma unroll
      for (int nt = 0; nt < CTN; ++nt) acc[j][nt] = zero4;
    // this lane's plane row of every tile (rows past the question / the wave's run read a valid row, never stored)
    unsigned aoff[kVqTPW];     // 32-bit byte offsets from the planes' (uniform) bases: scalar base + lane offset + immediate
#pragma unroll
    for (int j = 0; j < kVqTPW; ++j) {
      const int m = min(r0 + (tbase + (j < ntile ? j : 0)) * 16 + fr, r1 - 1);
      aoff[j] = (unsigned)a.rows[m].y * (unsigned)kVqRowB + (unsigned)fg * 16u;
    }
    const unsigned char* const plane_base[3] = {planes, planes + plane_stride, planes + 2 * plane_stride};

    for (int s = 0; s < 2; ++s) {           // (rolled: the peeled form spills ~300 registers)
      __syncthreads();                                        // q rows staged / the previous half's fragment reads done
      // ---- V planes of this half ----
      vq_stage<CTN, GNNRAG_VQ_UN>(a, lds, qarea, col0, ncol, d, s);
      __syncthreads();

      // ---- MFMA phase: a tile's A fragments of k block kb + 1 are requested as soon as its MFMAs of kb are issued,
      // i.e. the other tiles' MFMAs (kVqTPW - 1 times 6 CTN) before their use ----
      bf16x8 ap[kVqTPW][3];
#pragma unroll
      for (int j = 0; j < kVqTPW; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          ap[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base[pl] + (aoff[j] + (unsigned)(s * (kVqHalf * 2)))));
      for (int kb = 0; kb < NKB; ++kb) {        // (not unrolled: register pressure)
        const int kbn = min(kb + 1, NKB - 1);   // (the last block requests itself again: no branch around a load)
        const unsigned char* wb = lds + fr * RB + kb * 64 + fg * 16;
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
        constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int j = 0; j < kVqTPW; ++j) {
          if (j < ntile) {                                    // wave-uniform
#pragma unroll
            for (int nt = 0; nt < CTN; nt += 2) {
              bf16x8 b0[3], b1[3];
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) {
                b0[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb + pl * PL + nt * 16 * RB));
                b1[pl] = b0[pl];
                if (nt + 1 < CTN)
                  b1[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb + pl * PL + (nt + 1) * 16 * RB));
              }
#pragma unroll
              for (int p = 0; p < 6; ++p) {
                acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[j][PA[p]], b0[PB[p]], acc[j][nt], 0, 0, 0);
                if (nt + 1 < CTN)
                  acc[j][nt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[j][PA[p]], b1[PB[p]], acc[j][nt + 1], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)                      // refill: this tile's fragments of the next k block
            ap[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base[pl] + (aoff[j] + (unsigned)(s * (kVqHalf * 2) + kbn * 64))));
        }
      }
    }

    // ---- epilogue: the pass's tiles leave the registers (C layout: rows 4 fg + q, column slot fr) ----
#pragma unroll
    for (int j = 0; j < kVqTPW; ++j) {
      if (j < ntile) {
        const int rbase = r0 + (tbase + j) * 16 + 4 * fg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = rbase + q;
          if (row < r1) {
            float* prow = P + (size_t)row * D + col0;
            static_assert(CTN >= 4, "a column part holds at least the interleaved group");
            {
              const f32x4 v = {acc[j][0][q], acc[j][1][q], acc[j][2][q], acc[j][3][q]};
              const int c = 4 * fr;
              if (c + 4 <= ncol) *reinterpret_cast<f32x4*>(prow + c) = v;
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (c + e < ncol) prow[c + e] = v[e];
              }
            }
#pragma unroll
            for (int nt = 4; nt < CTN; ++nt) {
              const int c = nt * 16 + fr;
              if (c < ncol) prow[c] = acc[j][nt][q];
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(512, 2) void k_tables_vq(VqArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (a.zero) {               // every workgroup zeroes its share (a few floats per thread)
    const int64_t nb = (int64_t)gridDim.x * gridDim.y * gridDim.z;
    const int64_t lin = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
    for (int64_t i = lin * 512 + threadIdx.x; i < a.zero_n; i += nb * 512) a.zero[i] = 0.f;
  }
  const int NT = (a.D + 15) >> 4;
  const int h = blockIdx.z;
  const int ctn = h == 0 ? a.ct0 : NT - a.ct0;
  const int col0 = h == 0 ? 0 : a.ct0 * 16;
  switch (ctn) {
    case 6: tables_vq_part<6>(a, lds, col0); break;
    case 7: tables_vq_part<7>(a, lds, col0); break;
    default: break;
  }
}

// ---- k_tables_vq_lite: the V form for the stack driver's SIDE STREAM (round 6) ------------------------------------------
// Same product, same six plane products per (column tile, k block) in the same order on the same operands - every
// element of P gets the same fp32 sum as from k_tables_vq, bit for bit - but cut so that ONE of these workgroups fits
// a CU BESIDE a workgroup of the LDS walk (k_walk_slice: 16 waves x 64 registers, ~80 KB of LDS at 602 relations) or
// beside the small frontier / softmax launches, instead of owning the CU:
//   * 256 threads = ONE wave per SIMD with a 256-register budget (the walk's four waves per SIMD hold the other 256);
//   * column parts of 3 / 3 / 3 / 2 / 2 tiles: the V planes of one half of K for 48 columns are 3 x 48 x 416 B = 58.5 KB
//     (61.5 KB with the instruction rows; the walk's 79.7 KB beside it leave room);
//   * a wave owns up to 10 row tiles (38 tiles of a 602-relation question over 4 waves: one pass) - 10 x 3 x 4 = 120
//     accumulator registers -, reads a k block's B fragments of all column tiles ONCE for its ten row tiles (36
//     registers; no LDS read inside the MFMA stream) and keeps only TWO tiles' A fragments in flight (the current one and
//     the next one's, requested a whole tile = 18 MFMAs ahead) instead of all tiles' (12 registers per tile).
// The matrix pipe is then fed by one wave per SIMD whose stalls nothing of its own kernel hides - that is the point: the
// co-resident walk's waves issue VALU / LDS / SALU work into exactly those slots, and the walk's own s_waitcnt idle
// (64 % of its wave-cycles) is where the MFMAs run.
constexpr int kVlThreads = 256;
constexpr int kVlWaves = kVlThreads / 64;
#ifndef GNNRAG_VL_TPW
#define GNNRAG_VL_TPW 10
#endif
constexpr int kVlTPW = GNNRAG_VL_TPW;          // row tiles per wave and pass (even: the two fragment sets alternate)
constexpr int kVlCT = 3;                       // column tiles of the widest column part = rows of an LDS plane / 16
#ifndef GNNRAG_VL_NB
#define GNNRAG_VL_NB 5
#endif
constexpr int kVlNB = GNNRAG_VL_NB;            // A-fragment sets in flight per wave (ring; kVlTPW is a multiple of it)
static_assert(kVlTPW % kVlNB == 0, "the A-fragment ring keeps its phase from one k block to the next");

__device__ __forceinline__ int vl_lds_row(int j, int ctn) { return ctn >= 4 ? tab_lds_row(j) : j; }   // (parts of < 4 tiles: plain order)

template <int CTN, int UN>
__device__ __forceinline__ void vl_stage(const VqArgs& a, unsigned char* lds, const float* qarea, int col0, int ncol,
                                         int d, int s) {
  constexpr int RB = kTabSlots * 16;
  constexpr int PL = kVlCT * 16 * RB;
  const int tid = threadIdx.x;
  const int D = a.D, I = a.I, KC = D >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int rows = CTN < kVlCT ? CTN * 16 + 1 : CTN * 16;    // + one zero row behind a part that does not fill the plane
  const int total = rows * kTabSlots * 2;                        // 8-byte pieces (4 k) per plane
  for (int base = 0; base < total; base += kVlThreads * UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = zero4;
    for (int i = 0; i < I; ++i) {
      f32x4 w[UN];
      const float* wsrc = a.W + (size_t)col0 * a.ldw + (1 + 2 * i + d) * D;
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * kVlThreads + tid;
        const int j = idx / (kTabSlots * 2), kc = idx - j * (kTabSlots * 2);
        w[u] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)min(j, ncol - 1) * a.ldw + 4 * min(kc, KC - 1));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * kVlThreads + tid;
        const int kc = idx % (kTabSlots * 2);
        f32x4 q = *reinterpret_cast<const f32x4*>(qarea + i * D + 4 * min(kc, KC - 1));
        q = __builtin_elementwise_max(s ? -q : q, zero4);
        v[u] += w[u] * q;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = base + u * kVlThreads + tid;
      const int j = idx / (kTabSlots * 2), kc = idx - j * (kTabSlots * 2);
      if (idx < total) {
        const Split3 sp = split3(j < ncol && kc < KC ? v[u] : zero4);
        unsigned char* dst = lds + vl_lds_row(j, CTN) * RB + kc * 8;
        *reinterpret_cast<uint2*>(dst) = sp.hi;
        *reinterpret_cast<uint2*>(dst + PL) = sp.mid;
        *reinterpret_cast<uint2*>(dst + 2 * PL) = sp.lo;
      }
    }
  }
}

template <int CTN>
__device__ __forceinline__ void tables_vl_part(const VqArgs& a, unsigned char* lds, int col0) {
  constexpr int RB = kTabSlots * 16;
  constexpr int PL = kVlCT * 16 * RB;
  constexpr int NKB = kTabNKB;
  constexpr int TPW = kVlTPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int d = a.dir0 + blockIdx.y;
  const int g = blockIdx.x / a.nchunk, ch = blockIdx.x - g * a.nchunk;
  const int D = a.D, I = a.I;
  const int r0 = a.rel_off[g], r1 = a.rel_off[g + 1];
  if (r1 <= r0) return;
  const int ncol = min(CTN * 16, D - col0);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* P = a.P + (size_t)d * a.M * D;
  const unsigned char* planes = a.planes + (size_t)d * 3 * a.R1 * kVqRowB;
  const size_t plane_stride = (size_t)a.R1 * kVqRowB;

  const int U = (r1 - r0 + 15) >> 4;
  const int c0 = U * ch / a.nchunk, c1 = U * (ch + 1) / a.nchunk;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int w0 = c0 + (c1 - c0) * wv / kVlWaves, w1 = c0 + (c1 - c0) * (wv + 1) / kVlWaves;
  const int npass = ((c1 - c0 + kVlWaves - 1) / kVlWaves + TPW - 1) / TPW;     // workgroup-uniform (the barriers below)

  if (tid < 16) reinterpret_cast<unsigned*>(lds + 3 * PL)[tid] = 0u;   // slack behind the last plane stays finite
  float* qarea = reinterpret_cast<float*>(lds + 3 * PL + 64);          // ins[g, :, :]
  for (int x = tid * 4; x < I * D; x += kVlThreads * 4)
    *reinterpret_cast<f32x4*>(qarea + x) = *reinterpret_cast<const f32x4*>(a.ins + (size_t)g * I * D + x);

  for (int pass = 0; pass < npass; ++pass) {
    const int tbase = w0 + pass * TPW;
    const int ntile = max(0, min(TPW, w1 - tbase));           // wave-uniform
    f32x4 acc[TPW][CTN];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int nt = 0; nt < CTN; ++nt) acc[j][nt] = zero4;
    unsigned aoff[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int m = min(r0 + (tbase + (j < ntile ? j : 0)) * 16 + fr, r1 - 1);
      aoff[j] = (unsigned)a.rows[m].y * (unsigned)kVqRowB + (unsigned)fg * 16u;
    }
    const unsigned char* const plane_base[3] = {planes, planes + plane_stride, planes + 2 * plane_stride};

    for (int s = 0; s < 2; ++s) {
      __syncthreads();
      vl_stage<CTN, 3>(a, lds, qarea, col0, ncol, d, s);
      __syncthreads();

      // a ring of NB A-fragment sets: tile j computes from set j % NB while the set of the tile NB - 1 slots ahead (behind
      // the last tile: the first tiles of the next k block) is requested - an L2 hit takes ~3 tiles' worth of MFMAs
      constexpr int NB = kVlNB;
      bf16x8 ap[NB][3];
#pragma unroll
      for (int j = 0; j < NB - 1; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          ap[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base[pl] + (aoff[j] + (unsigned)(s * (kVqHalf * 2)))));
      for (int kb = 0; kb < NKB; ++kb) {
        const int kbn = min(kb + 1, NKB - 1);
        const unsigned char* wb = lds + fr * RB + kb * 64 + fg * 16;
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
        constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
        // the k block's B fragments of ALL column tiles, once: every row tile of the wave multiplies against the same
        // V columns, so the MFMA stream below runs without an LDS read inside it
        bf16x8 bfr[CTN][3];
#pragma unroll
        for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            bfr[nt][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb + pl * PL + nt * 16 * RB));
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          {
            constexpr int AH = NB - 1;
            const unsigned nxt = j + AH < TPW ? aoff[j + AH < TPW ? j + AH : 0] + (unsigned)(s * (kVqHalf * 2) + kb * 64)
                                              : aoff[j + AH < TPW ? 0 : j + AH - TPW] + (unsigned)(s * (kVqHalf * 2) + kbn * 64);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              ap[(j + AH) % NB][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base[pl] + nxt));
          }
          if (j < ntile) {                                    // wave-uniform
            // p outermost: CTN independent accumulators between two MFMAs on the same one
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
              for (int nt = 0; nt < CTN; ++nt)
                acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[j % NB][PA[p]], bfr[nt][PB[p]], acc[j][nt], 0, 0, 0);
          }
          // the scheduler may not move the later tiles' fragment loads up across this point (it would hold all ten tiles'
          // fragments at once: 120 registers)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // ---- epilogue (C layout: rows 4 fg + q, column slot fr) ----
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (j < ntile) {
        const int rbase = r0 + (tbase + j) * 16 + 4 * fg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = rbase + q;
          if (row < r1) {
            float* prow = P + (size_t)row * D + col0;
            if constexpr (CTN >= 4) {       // interleaved rows: a lane holds four consecutive columns
              const f32x4 v = {acc[j][0][q], acc[j][1][q], acc[j][2][q], acc[j][3][q]};
              const int c = 4 * fr;
              if (c + 4 <= ncol) *reinterpret_cast<f32x4*>(prow + c) = v;
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (c + e < ncol) prow[c + e] = v[e];
              }
            } else {
#pragma unroll
              for (int nt = 0; nt < CTN; ++nt) {
                const int c = nt * 16 + fr;
                if (c < ncol) prow[c] = acc[j][nt][q];
              }
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(kVlThreads, 2) void k_tables_vq_lite(VqArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (a.zero) {
    const int64_t nb = (int64_t)gridDim.x * gridDim.y * gridDim.z;
    const int64_t lin = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
    for (int64_t i = lin * kVlThreads + threadIdx.x; i < a.zero_n; i += nb * kVlThreads) a.zero[i] = 0.f;
  }
  // 13 column tiles as parts of 3 / 3 / 3 / 2 / 2
  const int h = blockIdx.z;
  if (h < 3) tables_vl_part<3>(a, lds, h * 48);
  else tables_vl_part<2>(a, lds, 144 + (h - 3) * 32);
}

// ---- the self-block update in bf16x3 on the same weight-plane layout -------------------------------------------------
//   h'[m, :] = relu( h[m, :] . W_e2e[:, 0:D]^T + b + nbr[m, :] ),   score[m] = w_s . h'[m, :] + b_s + (1 - mask[m]) * -1e11
// (reasongnn.py:161-168 with the neighbour blocks already reduced into nbr).  Workgroup = (row chunk, column part of
// 7 / 6 column tiles): the part's three weight planes are split and staged ONCE and stay in LDS for all row tiles of
// the chunk; every wave owns a contiguous run of 16-row tiles and, as in k_gemm_wres, reads a tile's A fragments in
// the MFMA layout straight from global memory (two float4 = 8 consecutive k per lane and k block, 128-byte lines) a
// whole tile ahead and splits them into planes in registers; the epilogue works from the registers.  A is read by
// both parts of a chunk - they sit next to each other in the grid and on one XCD, so the second read hits L2.  The
// two parts' score dots are two commutative atomic adds onto a zeroed score (x + y == y + x: deterministic); the part
// that owns column 0 adds bias and mask term to its share first, so a masked slot still ends exactly at -1e11.
struct UpdB3Args {
  const float* A;        // h [M, D]
  const float* W;        // e2e_linear.weight [D, ldw]; columns 0..D-1 are the self block
  const float* bias;     // [D]
  const float* add;      // nbr [M, D]
  const float* w_s;      // [D]
  const float* b_s;      // [1]
  const float* mask;     // [M]
  float* C;              // [M, D]
  float* score;          // [M], zeroed before the launch
  const uint8_t* add_flag;   // FL: [M + 4] row gates of `add`; a row whose byte is 0 reads the zero row `add + M * D`
  int32_t M, D, ldw, ct0;
};

template <int CTN, bool FL>
__device__ __forceinline__ void update_b3_part(const UpdB3Args& a, unsigned char* lds, int col0, bool first_part,
                                               int chunk, int nchunks) {
  constexpr int RB = kTabSlots * 16;
  constexpr int PL = kTabNTH * 16 * RB;
  constexpr int NKB = kTabNKB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int D = a.D;
  const int ncol = min(CTN * 16, D - col0);
  const int KC = D >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* Bl = reinterpret_cast<float*>(lds + 3 * PL + 64);      // bias / score weights of this part's columns
  float* Sl = Bl + kTabNTH * 16;

  if (tid < 16) reinterpret_cast<unsigned*>(lds + 3 * PL)[tid] = 0u;
  for (int j = tid; j < kTabNTH * 16; j += 512) {
    Bl[j] = (j < ncol && a.bias) ? a.bias[col0 + j] : 0.f;
    Sl[j] = j < ncol ? a.w_s[col0 + j] : 0.f;
  }
  {   // weight planes of this column part (self block: columns 0..D-1 of e2e_linear.weight)
    const int total = tab_stage_rows(CTN) * kTabSlots * 2;
    constexpr int UN = 6;
    for (int base = 0; base < total; base += 512 * UN) {
      f32x4 v[UN];
      int off[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * 512 + tid;
        const int j = idx / (kTabSlots * 2), kc = idx - j * (kTabSlots * 2);
        v[u] = zero4;
        off[u] = idx < total ? tab_lds_row(j) * RB + kc * 8 : -1;
        if (idx < total && j < ncol && kc < KC)
          v[u] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(col0 + j) * a.ldw + 4 * kc);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (off[u] >= 0) {
          const Split3 sp = split3(v[u]);
          unsigned char* dst = lds + off[u];
          *reinterpret_cast<uint2*>(dst) = sp.hi;
          *reinterpret_cast<uint2*>(dst + PL) = sp.mid;
          *reinterpret_cast<uint2*>(dst + 2 * PL) = sp.lo;
        }
      }
    }
  }
  // this wave's 16-row tiles
  const long long U = ((long long)a.M + 15) >> 4;
  const int c0 = (int)(U * chunk / nchunks), c1 = (int)(U * (chunk + 1) / nchunks);
  const int nch = c1 - c0;
  int t = c0 + (int)((long long)nch * wave / 8);
  const int tend = c0 + (int)((long long)nch * (wave + 1) / 8);
  const int kmax = D - 8;
  // Addresses (round 5).  PMC showed the VALU pipe as busy as the matrix pipe (14.9 M VALU against 4.4 M MFMA instructions
  // per launch, profiles/r05c_pmc_update_b3.txt), and the ISA showed why: ~40 % of the loop's VALU instructions were
  // 64-bit address arithmetic the compiler rematerialised per access (v_lshl_add_u64, v_mad_i64_i32, clamps) and v_add_u32
  // for LDS offsets beyond the 16-bit immediate.  Now every global access is SCALAR BASE + 32-BIT LANE OFFSET + immediate
  // (the launcher admits (M + 1) * D * 4 < 2^32): one row offset per tile, lane constants for the column pieces; the
  // three LDS planes have one base register each, so that every fragment offset fits the immediate.
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* addb = reinterpret_cast<const unsigned char*>(a.add);
  unsigned char* Cb = reinterpret_cast<unsigned char*>(a.C);
  const unsigned rowB = (unsigned)D * 4u;
  const unsigned k6B = (unsigned)min(32 * (NKB - 1) + 8 * fg, kmax) * 4u;      // the last k block's (clamped) piece
  auto a_rowoff = [&](int tile) -> unsigned { return (unsigned)min(tile * 16 + fr, a.M - 1) * rowB; };
  // raw A pieces of a tile: per k block two float4 (k = 32 kb + 8 fg .. + 7), unconditional (clamped addresses)
  auto a_piece = [&](unsigned rowoff, int kb, int half) -> f32x4 {
    const unsigned off = kb < NKB - 1 ? rowoff + 32u * (unsigned)fg + (unsigned)(128 * kb + 16 * half)
                                      : rowoff + k6B + (unsigned)(16 * half);
    return *reinterpret_cast<const f32x4*>(Ab + off);
  };
  f32x4 ra[NKB][2];
  if (t < tend) {
    const unsigned ro = a_rowoff(t);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      ra[kb][0] = a_piece(ro, kb, 0);
      ra[kb][1] = a_piece(ro, kb, 1);
    }
  }
  const float bs = a.b_s[0];
  // lane constants of the epilogue's column pieces (bytes inside a row of nbr / h')
  const unsigned cgB = (unsigned)(col0 + min(4 * fr, ncol - 4)) * 4u;
  unsigned ctB[CTN > 4 ? CTN - 4 : 1];
#pragma unroll
  for (int nt = 4; nt < CTN; ++nt) ctB[nt - 4] = (unsigned)(col0 + min(nt * 16 + fr, ncol - 1)) * 4u;
  // FL: the four row gates of a lane's rows (one dword), requested a tile ahead like the A pieces
  unsigned fl_next = 0x01010101u;
  if (FL && t < tend) fl_next = *reinterpret_cast<const unsigned*>(a.add_flag + (size_t)t * 16 + 4 * fg);
  __syncthreads();

  for (; t < tend; ++t) {
    const int rbase = t * 16 + 4 * fg;                       // C layout: rows rbase + q, column slot fr
    const unsigned fl = fl_next;
    if (FL) fl_next = *reinterpret_cast<const unsigned*>(a.add_flag + (size_t)(t + 1 < tend ? t + 1 : t) * 16 + 4 * fg);
    // the epilogue's operands: nbr in the register layout (interleaved group: 4 consecutive columns per lane)
    f32x4 addg[4];                                           // group 0 (column tiles 0..3), per row q
    float addt[CTN > 4 ? CTN - 4 : 1][4];                    // plain tiles 4.., per row q
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int row = min(rbase + q, a.M - 1);
      if (FL && ((fl >> (8 * q)) & 0xffu) == 0u) row = a.M;      // not a frontier row: the zero row behind the buffer
      const unsigned ro = (unsigned)row * rowB;
      addg[q] = *reinterpret_cast<const f32x4*>(addb + (ro + cgB));
#pragma unroll
      for (int nt = 4; nt < CTN; ++nt) addt[nt - 4][q] = *reinterpret_cast<const float*>(addb + (ro + ctB[nt - 4]));
    }
    float mrow = 0.f;
    {
      const int srow = min(rbase + (2 * (fr & 1) + ((fr >> 1) & 1)), a.M - 1);
      mrow = a.mask[srow];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[CTN];
#pragma unroll
    for (int nt = 0; nt < CTN; ++nt) acc[nt] = zero4;
    const unsigned ro_next = a_rowoff(t + 1 < tend ? t + 1 : t);
    // one base register per LDS plane (made opaque once per tile: keeps the loop-invariant plane reads inside the loop)
    const unsigned char* wbp[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      unsigned o = (unsigned)(fr * RB + fg * 16 + pl * PL);
      asm volatile("" : "+v"(o));
      wbp[pl] = lds + o;
    }
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
    constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      bf16x8 ap[3];
      // only the LAST k block can reach past D (the launcher admits 192 < D <= 208): no select in the others
      const bool kok = kb < NKB - 1 || 32 * kb + 8 * fg <= kmax;
      const Split3 s0 = split3(kok ? ra[kb][0] : zero4);
      const Split3 s1 = split3(kok ? ra[kb][1] : zero4);
      ap[0] = __builtin_bit_cast(bf16x8, (u32x4){s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y});
      ap[1] = __builtin_bit_cast(bf16x8, (u32x4){s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y});
      ap[2] = __builtin_bit_cast(bf16x8, (u32x4){s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y});
      ra[kb][0] = a_piece(ro_next, kb, 0);                   // refill: the next tile's k block kb
      ra[kb][1] = a_piece(ro_next, kb, 1);
#pragma unroll
      for (int nt = 0; nt < CTN; nt += 2) {
        bf16x8 b0[3], b1[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          b0[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wbp[pl] + (nt * 16 * RB + kb * 64)));
          b1[pl] = b0[pl];
          if (nt + 1 < CTN)
            b1[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wbp[pl] + ((nt + 1) * 16 * RB + kb * 64)));
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PA[p]], b0[PB[p]], acc[nt], 0, 0, 0);
          if (nt + 1 < CTN)
            acc[nt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PA[p]], b1[PB[p]], acc[nt + 1], 0, 0, 0);
        }
      }
    }
    // epilogue from the registers
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias_g = *reinterpret_cast<const f32x4*>(Bl + min(4 * fr, kTabNTH * 16 - 4));
    const f32x4 ws_g = *reinterpret_cast<const f32x4*>(Sl + min(4 * fr, kTabNTH * 16 - 4));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rbase + q;
      const unsigned ro = (unsigned)min(row, a.M - 1) * rowB;
      f32x4 v = {acc[0][q], acc[1][q], acc[2][q], acc[3][q]};
      v = __builtin_elementwise_max((v + bias_g) + addg[q], zero4);
      const int c = 4 * fr;
      if (c + 4 > ncol) {                                    // (ncol % 4 == 0: a lane's group is all in or all out)
        v = zero4;
      } else if (row < a.M) {
        *reinterpret_cast<f32x4*>(Cb + (ro + cgB)) = v;
      }
      part[q] += v[0] * ws_g[0] + v[1] * ws_g[1] + v[2] * ws_g[2] + v[3] * ws_g[3];
#pragma unroll
      for (int nt = 4; nt < CTN; ++nt) {
        const int cc = nt * 16 + fr;
        float x = fmaxf((acc[nt][q] + Bl[cc]) + addt[nt - 4][q], 0.f);
        if (cc >= ncol) x = 0.f;
        else if (row < a.M) *reinterpret_cast<float*>(Cb + (ro + ctB[nt - 4])) = x;
        part[q] += x * Sl[cc];
      }
    }
    {
      const float tot = row16_sum4_b3(part, lane);
      const int srow = rbase + (2 * (fr & 1) + ((fr >> 1) & 1));
      if (fr < 4 && srow < a.M) {
        // this part's share of the score; the first part carries bias and mask term (fp32 adds: a masked slot's
        // share is exactly -1e11 and stays there when the other part's share is added)
        const float share = first_part ? (tot + bs) + (1.0f - mrow) * kVeryNeg : tot;
        atomicAdd(a.score + srow, share);
      }
    }
  }
}

template <bool FL>
__global__ __launch_bounds__(512, 2) void k_update_b3(UpdB3Args a, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int blk = blockIdx.x;
  const int NT = (a.D + 15) >> 4;
  // the two parts of a row chunk are neighbours in the grid AND on one XCD (blocks b and b + 8):
  // block = 16*(c/8) + 8*h + c%8
  const int h = (blk >> 3) & 1;
  const int chunk = (blk >> 4) * 8 + (blk & 7);
  if (chunk >= nchunks) return;
  if (h == 0) update_b3_part<kTabNTH, FL>(a, lds, 0, true, chunk, nchunks);
  else if (NT - a.ct0 == 6) update_b3_part<6, FL>(a, lds, a.ct0 * 16, false, chunk, nchunks);
}

int update_b3_launch(const float* h, const float* nbr, const float* W, const float* b, const float* w_s, const float* b_s,
                     const float* mask, float* h_out, float* score, int64_t BN, int32_t D, int32_t ldw,
                     hipStream_t stream) {
  return update_b3_launch_z(h, nbr, W, b, w_s, b_s, mask, h_out, score, BN, D, ldw, stream, false);
}

int update_b3_launch_z(const float* h, const float* nbr, const float* W, const float* b, const float* w_s,
                       const float* b_s, const float* mask, float* h_out, float* score, int64_t BN, int32_t D,
                       int32_t ldw, hipStream_t stream, bool score_zeroed) {
  return update_b3_launch_f(h, nbr, nullptr, W, b, w_s, b_s, mask, h_out, score, BN, D, ldw, stream, score_zeroed);
}

bool update_b3_shape_ok(int64_t BN, int32_t D, int32_t ldw) {
  // (32-bit byte offsets from the kernel's base pointers: the zero row behind nbr included)
  return !(D % 8 || (D + 31) / 32 != kTabNKB || (D + 15) / 16 != 13 || BN < 8192 || (BN + 1) * D * 4 >= ((int64_t)1 << 32) || ldw % 4);
}

int update_b3_launch_f(const float* h, const float* nbr, const uint8_t* add_flag, const float* W, const float* b,
                       const float* w_s, const float* b_s, const float* mask, float* h_out, float* score, int64_t BN,
                       int32_t D, int32_t ldw, hipStream_t stream, bool score_zeroed) {
  if (!update_b3_shape_ok(BN, D, ldw)) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)h | (uintptr_t)nbr | (uintptr_t)W | (uintptr_t)h_out) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  UpdB3Args a;
  memset(&a, 0, sizeof(a));
  a.A = h; a.W = W; a.bias = b; a.add = nbr; a.w_s = w_s; a.b_s = b_s; a.mask = mask; a.C = h_out; a.score = score;
  a.M = (int32_t)BN; a.D = D; a.ldw = ldw; a.ct0 = kTabNTH;
  a.add_flag = add_flag;
  if (add_flag && ((uintptr_t)add_flag & 3)) return GNNRAG_E_UNSUPPORTED;
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  const long long U = (BN + 15) / 16;
  int chunks = cus / 2;
  if (chunks < 1) chunks = 1;
  if ((long long)chunks * 8 > U) chunks = (int)((U + 7) / 8);
  if (!score_zeroed) GNNRAG_HIP(hipMemsetAsync(score, 0, (size_t)BN * sizeof(float), stream));
  static DeviceMask cap, cap_f;
  {
    const int rc = add_flag ? raise_lds_cap(k_update_b3<true>, cap_f) : raise_lds_cap(k_update_b3<false>, cap);
    if (rc) return rc;
  }
  const int nblk = ((chunks + 7) / 8) * 16;
  if (add_flag) hipLaunchKernelGGL(k_update_b3<true>, dim3(nblk), dim3(512), 160 * 1024, stream, a, chunks);
  else hipLaunchKernelGGL(k_update_b3<false>, dim3(nblk), dim3(512), 160 * 1024, stream, a, chunks);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

size_t tables_vq_planes_bytes(int64_t R1) { return (size_t)2 * 3 * R1 * kVqRowB; }

bool tables_vq_shape_ok(int32_t D, int32_t I) {
  return D % 8 == 0 && (D + 31) / 32 == kTabNKB && (D + 15) / 16 == 13 && I >= 1 &&
         (size_t)I * D * sizeof(float) <= (size_t)kTabQBytes;
}

int tables_vq_launch(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                     int32_t I, int32_t only_dir, hipStream_t stream) {
  return tables_vq_launch_z(csr, planes, ins, W, P, D, I, only_dir, nullptr, 0, stream);
}

int tables_vq_launch_z(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                       int32_t I, int32_t only_dir, float* zero, int64_t zero_n, hipStream_t stream) {
  if (!tables_vq_shape_ok(D, I) || csr->rel_total < 1024 || only_dir > 1) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)planes | (uintptr_t)ins | (uintptr_t)W | (uintptr_t)P) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  VqArgs a;
  memset(&a, 0, sizeof(a));
  a.planes = (const unsigned char*)planes; a.ins = ins; a.W = W; a.P = P;
  a.rows = (const int2*)csr->rel_rows; a.rel_off = csr->rel_off;
  a.M = csr->rel_total; a.D = D; a.I = I; a.ldw = (2 * I + 1) * D; a.R1 = csr->R1;
  const int NT = (D + 15) / 16;
  a.ct0 = (NT + 1) / 2;
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  // one workgroup per (question, direction, column part) fills the chip from 64 questions on; smaller batches cut a
  // question's rows into chunks (each builds the question's V again), as long as a chunk keeps >= 8 tiles
  const int ndir = only_dir < 0 ? 2 : 1;
  a.dir0 = only_dir < 0 ? 0 : only_dir;
  int nchunk = cus / (csr->B * 2 * ndir);
  const int tiles_max = (csr->rel_max + 15) / 16;
  if (nchunk > tiles_max / 8) nchunk = tiles_max / 8;
  if (nchunk < 1) nchunk = 1;
  a.nchunk = nchunk;
  a.zero = zero;
  a.zero_n = zero ? zero_n : 0;
  static DeviceMask cap;
  {
    const int rc = raise_lds_cap(k_tables_vq, cap);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_tables_vq, dim3(csr->B * nchunk, ndir, 2), dim3(512), 160 * 1024, stream, a);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

// the side-stream form (k_tables_vq_lite): same results as tables_vq_launch_z, bit for bit
int tables_vq_lite_launch_z(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                            int32_t I, int32_t only_dir, float* zero, int64_t zero_n, hipStream_t stream) {
  if (!tables_vq_shape_ok(D, I) || csr->rel_total < 1024 || only_dir > 1) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)planes | (uintptr_t)ins | (uintptr_t)W | (uintptr_t)P) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  VqArgs a;
  memset(&a, 0, sizeof(a));
  a.planes = (const unsigned char*)planes; a.ins = ins; a.W = W; a.P = P;
  a.rows = (const int2*)csr->rel_rows; a.rel_off = csr->rel_off;
  a.M = csr->rel_total; a.D = D; a.I = I; a.ldw = (2 * I + 1) * D; a.R1 = csr->R1;
  a.ct0 = kVlCT;
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  const int ndir = only_dir < 0 ? 2 : 1;
  a.dir0 = only_dir < 0 ? 0 : only_dir;
  int nchunk = cus / (csr->B * 5 * ndir);
  const int tiles_max = (csr->rel_max + 15) / 16;
  if (nchunk > tiles_max / 8) nchunk = tiles_max / 8;
  if (nchunk < 1) nchunk = 1;
  a.nchunk = nchunk;
  a.zero = zero;
  a.zero_n = zero ? zero_n : 0;
  static DeviceMask cap;
  {
    const int rc = raise_lds_cap(k_tables_vq_lite, cap);
    if (rc) return rc;
  }
  size_t lds = (size_t)3 * kVlCT * 16 * kTabSlots * 16 + 64 + (size_t)I * D * sizeof(float);
  // GNNRAG_VL_LDS_KB (experiment knob): a larger LDS request than the kernel needs limits how many of its workgroups share
  // a CU (61.5 KB: two; above 80 KB: one - the other half of the CU stays free for a workgroup of the walk)
  if (const char* e = getenv("GNNRAG_VL_LDS_KB")) {
    const size_t want = (size_t)atoi(e) * 1024;
    if (want > lds && want <= 160 * 1024) lds = want;
  }
  hipLaunchKernelGGL(k_tables_vq_lite, dim3(csr->B * nchunk, ndir, 5), dim3(kVlThreads), lds, stream, a);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

int tables_b3_launch(const gnnrag_csr* csr, const float* T_fwd, const float* T_inv, const float* ins, const float* W,
                     float* P, int32_t D, int32_t I, hipStream_t stream) {
  // shapes of the kernel: D a multiple of 8 (a lane's 8 consecutive k), at most 2 x 7 column tiles, 4-byte offsets
  // that keep float4 accesses aligned, enough rows to fill the chip
  if (D % 8 || (D + 31) / 32 != kTabNKB || (D + 15) / 16 != 13 || csr->rel_total < 1024) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)T_fwd | (uintptr_t)T_inv | (uintptr_t)ins | (uintptr_t)W | (uintptr_t)P) & 15) != 0)
    return GNNRAG_E_UNSUPPORTED;
  TabArgs a;
  memset(&a, 0, sizeof(a));
  a.T[0] = T_fwd; a.T[1] = T_inv; a.ins = ins; a.W = W; a.P = P;
  a.rows = (const int2*)csr->rel_rows;
  a.M = csr->rel_total; a.D = D; a.I = I; a.ldw = (2 * I + 1) * D;
  const int NT = (D + 15) / 16;
  a.ct0 = NT <= kTabNTH ? NT : (NT + 1) / 2;
  const int parts = NT <= kTabNTH ? 1 : 2;
  int cus = 0;
  {
    const int rc = device_cu_count(&cus);
    if (rc) return rc;
  }
  const long long U = ((long long)a.M + 15) / 16;
  int chunks = cus / (2 * parts);
  if (chunks < 1) chunks = 1;
  if ((long long)chunks * 8 > U) chunks = (int)((U + 7) / 8);
  const int tiles_per_wave = (int)((((U + chunks - 1) / chunks) + 7) / 8);         // largest w1 - w0 of the floor split
  a.npass = (tiles_per_wave + kTabTPW - 1) / kTabTPW;
  const size_t lds = 160 * 1024;          // three weight planes, 64 B of slack, instruction rows
  static DeviceMask cap;
  {
    const int rc = raise_lds_cap(k_tables_b3, cap);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_tables_b3, dim3(chunks, 2, parts), dim3(512), lds, stream, a);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag

extern "C" int gnnrag_relation_tables_planes(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W,
                                             float* P, int32_t D, int32_t I, gnnrag_stream_t stream) {
  if (!csr || !planes || !ins || !W || !P || D <= 0 || I <= 0 || csr->rel_total < 0) return GNNRAG_E_BADARG;
  if (csr->rel_total == 0) return 0;
  return gnnrag::tables_vq_launch(csr, planes, ins, W, P, D, I, -1, (hipStream_t)stream);
}
