// The self-block update of the fused path on 32x32x16 matrix instructions (round 5; gfx950, hidden size 200):
//
//   h'[m, :] = relu( h[m, :] . W_e2e[:, 0:D]^T + b + nbr[m, :] ),   score[m] = w_s . h'[m, :] + b_s + (1 - mask[m]) * -1e11
//
// (reasongnn.py:161-168 with the neighbour blocks already reduced into nbr; the bf16x3 arithmetic of k_update_b3 in
// tables_b3.hip: exact 3-way split of both operands, six plane products, fp32 accumulation).
//
// Why another shape.  k_update_b3 (v_mfma_f32_16x16x32_bf16, 16-row tiles) keeps the matrix pipe ~47 % busy: per MFMA it
// issues 2.4 VALU, 0.54 ds_read_b128 and ~0.4 other instructions, and a 16x16x32 MFMA occupies the pipe for 16 cycles -
// four issue slots - so the stream cannot hide its own fillers (profiles/r04a_pmc_dense_layer_C2.txt; MI355X_MICROARCH.md
// "single-issue instructions hidden per MFMA gap").  v_mfma_f32_32x32x16_bf16 does twice the flops per instruction in a
// 32-cycle slot: the same split work per element of h (it is per element), but HALF the MFMA instructions, HALF the
// weight-fragment reads from LDS (a fragment now serves 32 rows) and one epilogue per 32 rows.  Per MFMA: ~1.6 VALU,
// 0.5 ds_read - inside the ~5 fillers a 32-cycle slot hides.
//
//   * workgroup = (row chunk, column part); part 0 = column tiles 0..3 (columns 0..127), part 1 = tiles 4..6 (128..199;
//     the 7th tile holds 8 real columns).  The part's three weight planes stay in LDS for the whole chunk
//     (3 x 128 x 400 B = 150 KB; row stride 25 slots of 16 B - odd, so the 16 lanes of a ds_read_b128 group hit 16
//     different 4-bank groups: conflict free without padding);
//   * operands swapped: MFMA A = weight fragment (32 output columns x 16 k), B = node fragment (32 rows of h x 16 k) -
//     a lane's 16 accumulators of a tile are 4 x 4 CONSECUTIVE columns of ONE row (columns 8 g + 4 (lane / 32) + e), so
//     nbr is read and h' written in 16-byte pieces, two lanes completing 32 bytes, and a row's score share is a
//     per-lane dot plus one exchange with lane + 32;
//   * every wave owns a run of 32-row tiles; raw rows of h come straight from global memory in the fragment layout
//     (two float4 per lane and k step) through a 7-slot register ring filled 6-13 k steps (~4-9 us) ahead, and are
//     split in registers; nbr is requested at k step 2 of its tile, the row gates / mask a tile ahead;
//   * the two parts' score shares are two commutative atomic adds onto a zeroed score (as k_update_b3).
#include "gnnrag_common.h"
#include "dense_internal.h"

namespace gnnrag {

constexpr int kXD = 200;            // the hidden size this kernel is written for
constexpr int kXKS = 13;            // k steps of 16 (k >= 200 is zero on the node side)
constexpr int kXRowB = 400;         // LDS bytes per weight row of one plane
constexpr int kXRing = 7;           // register ring of raw node pieces (k steps)
constexpr int kXCT0 = 3;            // column tiles (of 32) of part 0 (columns 0..95); part 1 takes the other 4 (96..199: its
                                    // last tile holds 8 real columns, so its epilogue operands are 52 registers, not 64)
// what may cross the scheduling fence behind a k step's global requests: VALU, SALU and LDS instructions (the next step's
// split may rise into this step's MFMAs), but neither MFMAs nor vector-memory instructions - without it the scheduler
// sinks every early request down to its first use and the wait becomes vmcnt(0)
#ifndef GNNRAG_X32_SCHED_MASK
#define GNNRAG_X32_SCHED_MASK 0x386
#endif
constexpr int kXSchedMask = GNNRAG_X32_SCHED_MASK;
#ifndef GNNRAG_X1_VALU
#define GNNRAG_X1_VALU 3            // form 2: VALU instructions pinned behind every MFMA of a region
#endif
constexpr int kX1Valu = GNNRAG_X1_VALU;

struct UpdXArgs {
  const float* A;        // h [M, D]
  const float* W;        // e2e_linear.weight [D, ldw]; columns 0..D-1 are the self block
  const float* bias;     // [D] or null
  const float* add;      // nbr [M (+1 zero row when gated), D]
  const float* w_s;      // [D]
  const float* b_s;      // [1]
  const float* mask;     // [M]
  float* C;              // [M, D]
  float* score;          // [M], zeroed before the launch
  const uint8_t* add_flag;   // FL: [M + padding] row gates of `add`; a row whose byte is 0 reads the zero row add + M * D
  int32_t M, ldw;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_x __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int opaque_x(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int CTN, bool FL>
__device__ __forceinline__ void update_x32_part(const UpdXArgs& a, unsigned char* lds, const int col0, const bool first_part,
                                                const int chunk, const int nchunks) {
  constexpr int D = kXD;
  constexpr int PL = CTN * 32 * kXRowB;                          // bytes per plane
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ri = lane & 31, hi = lane >> 5;
  const int M = a.M;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* Bl = reinterpret_cast<float*>(lds + 3 * PL + 64);       // bias / score weights of this part's column slots
  float* Sl = Bl + CTN * 32;

  if (tid < 16) reinterpret_cast<unsigned*>(lds + 3 * PL)[tid] = 0u;       // slack behind the last plane stays finite
  for (int j = tid; j < CTN * 32; j += 512) {
    const int c = col0 + j;
    Bl[j] = (c < D && a.bias) ? a.bias[c] : 0.f;
    Sl[j] = c < D ? a.w_s[c] : 0.f;
  }
  {   // weight planes of this column part (self block: columns 0..D-1 of e2e_linear.weight), rows past D as zeros
    constexpr int KC = D / 4;                                    // float4 pieces per row
    constexpr int total = CTN * 32 * KC;
    constexpr int UN = 6;
    for (int base = 0; base < total; base += 512 * UN) {
      f32x4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * 512 + tid;
        const int j = idx / KC, kc = idx - j * KC;
        v[u] = zero4;
        if (idx < total && col0 + j < D) v[u] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(col0 + j) * a.ldw + 4 * kc);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * 512 + tid;
        if (idx < total) {
          const int j = idx / KC, kc = idx - j * KC;
          const Split3 sp = split3(v[u]);
          unsigned char* dst = lds + j * kXRowB + kc * 8;
          *reinterpret_cast<uint2*>(dst) = sp.hi;
          *reinterpret_cast<uint2*>(dst + PL) = sp.mid;
          *reinterpret_cast<uint2*>(dst + 2 * PL) = sp.lo;
        }
      }
    }
  }
  // this wave's 32-row tiles
  const int U = (M + 31) >> 5;
  const int c0 = (int)((long long)U * chunk / nchunks), c1 = (int)((long long)U * (chunk + 1) / nchunks);
  const int nch = c1 - c0;
  int t = c0 + nch * wave / 8;
  const int tend = c0 + nch * (wave + 1) / 8;
  // raw node pieces: k step s, two float4 = k 16 s + 8 hi .. + 7 of row 32 tile + ri (clamped addresses, always loaded)
  auto a_piece = [&](int tile, int s, int half) -> f32x4 {
    const int row = min(tile * 32 + ri, M - 1);
    const int k = min(16 * s + 8 * hi, D - 8) + 4 * half;
    return *reinterpret_cast<const f32x4*>(a.A + (size_t)row * D + k);
  };
  f32x4 ra[kXRing][2];
  if (t < tend) {
#pragma unroll
    for (int s = 0; s < kXRing; ++s) {
      ra[s][0] = a_piece(t, s, 0);
      ra[s][1] = a_piece(t, s, 1);
    }
  }
  const float bs = a.b_s[0];
  // a tile ahead: the lane's row gate (FL) and mask value
  unsigned fl_next = 1u;
  float mk_next = 0.f;
  if (t < tend) {
    const int row = min(t * 32 + ri, M - 1);
    if (FL) fl_next = a.add_flag[row];
    mk_next = a.mask[row];
  }
  __syncthreads();

  constexpr int PA[6] = {1, 2, 0, 1, 0, 0};        // node plane of product p (smallest terms first: mid*mid, lo*hi, hi*lo, ...)
  constexpr int PB[6] = {1, 0, 2, 0, 1, 0};        // weight plane of product p
  for (; t < tend; ++t) {
    const int row = t * 32 + ri;                              // this lane's row of the tile
    const int tload = t + 1 < tend ? t + 1 : t;
    const unsigned fl = fl_next;
    const float mrow = mk_next;
    {
      const int rown = min(tload * 32 + ri, M - 1);
      if (FL) fl_next = a.add_flag[rown];
      mk_next = a.mask[rown];
    }
    const int hi_t = opaque_x(hi);                            // keeps the (loop invariant) plane reads inside the tile loop
    f32x16 acc[CTN];
#pragma unroll
    for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    f32x4 addv[CTN][4];
#pragma unroll
    for (int s = 0; s < kXKS; ++s) {
      constexpr int dummy = 0;
      (void)dummy;
      const int slot = s % kXRing;
      // (a full fence in front of the last step: its zero select below is VALU, which may cross the other fences, and the
      // scheduler hoisted it up to the request - 7 steps early, behind a vmcnt(0))
      if (s == kXKS - 1) __builtin_amdgcn_sched_barrier(0);
      f32x4 x0 = ra[slot][0], x1 = ra[slot][1];
      if (s == kXKS - 1) {                                    // k 200..207 does not exist: the upper lanes multiply zeros
        x0 = hi_t ? zero4 : x0;
        x1 = hi_t ? zero4 : x1;
      }
      const Split3 s0 = split3(x0), s1 = split3(x1);
      bf16x8 ap[3];
      ap[0] = __builtin_bit_cast(bf16x8, (u32x4_x){s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y});
      ap[1] = __builtin_bit_cast(bf16x8, (u32x4_x){s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y});
      ap[2] = __builtin_bit_cast(bf16x8, (u32x4_x){s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y});
      {   // refill the slot just consumed: static schedule (steps 0..5 -> same tile's s + 7, step 6 -> next tile's 6,
          // steps 7..12 -> next tile's s - 7)
        const int rt = s < kXRing - 1 ? t : tload;
        const int rs = s < kXRing - 1 ? s + kXRing : (s == kXRing - 1 ? kXRing - 1 : s - kXRing);
        ra[slot][0] = a_piece(rt, rs, 0);
        ra[slot][1] = a_piece(rt, rs, 1);
      }
      if (s != 2) __builtin_amdgcn_sched_barrier(kXSchedMask);   // the requests stay ABOVE this step's MFMAs
      if (s == 2) {                                           // the epilogue's operand: nbr in the accumulator layout
        int arow = min(row, M - 1);
        if (FL && fl == 0u) arow = M;                         // not a frontier row: the zero row behind the buffer
        const float* ap_ = a.add + (size_t)arow * D + col0 + 4 * hi;
#pragma unroll
        for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            addv[nt][g] = zero4;
            if (col0 + 32 * nt + 8 * g < D) addv[nt][g] = *reinterpret_cast<const f32x4*>(ap_ + 32 * nt + 8 * g);
          }
        __builtin_amdgcn_sched_barrier(kXSchedMask);
      }
      const unsigned char* wb = lds + ri * kXRowB + s * 32 + hi_t * 16;
      constexpr int G = CTN == 4 ? 2 : 3;                     // column tiles whose chains are interleaved
#pragma unroll
      for (int n0 = 0; n0 < CTN; n0 += G) {
        bf16x8 b[G][3];
#pragma unroll
        for (int n = 0; n < G; ++n)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[n][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb + pl * PL + (n0 + n) * 32 * kXRowB));
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int n = 0; n < G; ++n)
            acc[n0 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[n][PB[p]], ap[PA[p]], acc[n0 + n], 0, 0, 0);
      }
    }
    // epilogue from the registers: lane (ri, hi) holds row `row`, columns col0 + 32 nt + 8 g + 4 hi + e
    // (computing everything first and storing under ONE branch costs 9 spilled registers and vmcnt(0) waits: the
    // per-piece form below is what fits the 256 registers)
    float part = 0.f;
#pragma unroll
    for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (col0 + 32 * nt + 8 * g < D) {                     // (D % 8 == 0: a lane's group is all in or all out)
          const int cs = 32 * nt + 8 * g + 4 * hi;
          const f32x4 bias4 = *reinterpret_cast<const f32x4*>(Bl + cs);
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(Sl + cs);
          f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
          v = __builtin_elementwise_max((v + bias4) + addv[nt][g], zero4);
          if (row < M) *reinterpret_cast<f32x4*>(a.C + (size_t)row * D + col0 + cs) = v;
          part += v[0] * ws4[0] + v[1] * ws4[1] + v[2] * ws4[2] + v[3] * ws4[3];
        }
      }
    const float tot = part + __shfl_xor(part, 32);
    if (hi == 0 && row < M) {
      // this part's share of the score; the first part carries bias and mask term (fp32 adds: a masked slot's share is
      // exactly -1e11 and stays there when the other part's share is added)
      const float share = first_part ? (tot + bs) + (1.0f - mrow) * kVeryNeg : tot;
      atomicAdd(a.score + row, share);
    }
  }
}

// ---- form 2: ONE wave per SIMD (256 threads, up to 512 registers), software pipelined in the source ---------------------
// Form 1 above (two 256-register waves per SIMD) measured 105-118 us against k_update_b3's 94-107 (profiles/
// r05a_update_x32_v1_tune.txt): at 254 registers there is no room to request a group's weight fragments before the
// previous group's MFMAs, so every group starts with ds_read -> s_waitcnt lgkmcnt -> MFMA, and both waves of a SIMD stall
// the same way.  With one wave per SIMD the register file holds two fragment sets, two node-plane sets and a WHOLE tile
// of raw node pieces:
//   * the fragments of group g + 1 (or of the next k step's group 0) are requested before the MFMAs of group g;
//   * the node planes of step s + 1 are split from the ring while step s multiplies (first half beside group 0, second
//     half beside the last group), the ring slot is refilled with the NEXT tile's piece at once (distance: one tile);
//   * every region between two fences is pinned by sched_group_barrier to the pattern MFMA, <= V VALU, <= 1 ds_read:
//     the fillers sit in the 32-cycle shadows of the MFMAs instead of in front of them.
// Stores the compiler's s_waitcnt bookkeeping does not see.  On gfx9 loads and stores share vmcnt, and with both kinds
// outstanding the compiler treats the counter as unordered: the first use of ANY earlier load after the epilogue's stores
// becomes s_waitcnt vmcnt(0) - once per tile every wave drained its whole prefetch ring AND its stores (with one wave
// per SIMD nothing covers that).  Loads complete in order among themselves, so the compiler's counted waits stay valid
// with extra stores in the counter (a count of N or fewer outstanding operations still implies the (N+1)-th youngest
// load has landed); nothing in the kernel reads what these instructions write.
__device__ __forceinline__ void store16_nowait(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void atomic_add_nowait(float* p, float v) {
  asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

template <int CTN, bool FL>
__device__ __forceinline__ void update_x1_part(const UpdXArgs& a, unsigned char* lds, const int col0, const bool first_part,
                                               const int chunk, const int nchunks) {
  constexpr int D = kXD;
  constexpr int NTH = 256;
  constexpr int PL = CTN * 32 * kXRowB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ri = lane & 31, hi = lane >> 5;
  const int M = a.M;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* Bl = reinterpret_cast<float*>(lds + 3 * PL + 64);
  float* Sl = Bl + CTN * 32;

  if (tid < 16) reinterpret_cast<unsigned*>(lds + 3 * PL)[tid] = 0u;
  for (int j = tid; j < CTN * 32; j += NTH) {
    const int c = col0 + j;
    Bl[j] = (c < D && a.bias) ? a.bias[c] : 0.f;
    Sl[j] = c < D ? a.w_s[c] : 0.f;
  }
  {
    constexpr int KC = D / 4;
    constexpr int total = CTN * 32 * KC;
    constexpr int UN = 8;
    for (int base = 0; base < total; base += NTH * UN) {
      f32x4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * NTH + tid;
        const int j = idx / KC, kc = idx - j * KC;
        v[u] = zero4;
        if (idx < total && col0 + j < D) v[u] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(col0 + j) * a.ldw + 4 * kc);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * NTH + tid;
        if (idx < total) {
          const int j = idx / KC, kc = idx - j * KC;
          const Split3 sp = split3(v[u]);
          unsigned char* dst = lds + j * kXRowB + kc * 8;
          *reinterpret_cast<uint2*>(dst) = sp.hi;
          *reinterpret_cast<uint2*>(dst + PL) = sp.mid;
          *reinterpret_cast<uint2*>(dst + 2 * PL) = sp.lo;
        }
      }
    }
  }
  const int U = (M + 31) >> 5;
  const int c0 = (int)((long long)U * chunk / nchunks), c1 = (int)((long long)U * (chunk + 1) / nchunks);
  const int nch = c1 - c0;
  int t = c0 + nch * wave / 4;
  const int tend = c0 + nch * (wave + 1) / 4;
  auto a_piece = [&](int tile, int s, int half) -> f32x4 {
    const int row = min(tile * 32 + ri, M - 1);
    const int k = min(16 * s + 8 * hi, D - 8) + 4 * half;
    return *reinterpret_cast<const f32x4*>(a.A + (size_t)row * D + k);
  };
  f32x4 ra[kXKS][2];                                            // a whole tile of raw node pieces
  if (t < tend) {
#pragma unroll
    for (int s = 0; s < kXKS; ++s) {
      ra[s][0] = a_piece(t, s, 0);
      ra[s][1] = a_piece(t, s, 1);
    }
  }
  const float bs = a.b_s[0];
  unsigned fl_next = 1u;
  float mk_next = 0.f;
  if (t < tend) {
    const int row = min(t * 32 + ri, M - 1);
    if (FL) fl_next = a.add_flag[row];
    mk_next = a.mask[row];
  }
  __syncthreads();
  if (t >= tend) return;

  constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
  constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
  constexpr int NG = 2;                                         // MFMA groups per k step: column tiles {0, 1} and the rest
  constexpr int G0 = 2, G1 = CTN - 2;                           // tiles per group
  const unsigned char* wb0 = lds + ri * kXRowB + hi * 16;
  auto frag = [&](int s, int nt, int pl) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wb0 + s * 32 + pl * PL + nt * 32 * kXRowB));
  };
  // pipeline state carried from step to step (and over the tile boundary)
  bf16x8 bq[2][2][3];                                           // [buffer][tile of the group][plane]
  Split3 sn0, sn1;                                              // planes of the NEXT step's node pieces
  {
    sn0 = split3(ra[0][0]);
    sn1 = split3(ra[0][1]);
    const int tl = t + 1 < tend ? t + 1 : t;
    ra[0][0] = a_piece(tl, 0, 0);
    ra[0][1] = a_piece(tl, 0, 1);
#pragma unroll
    for (int n = 0; n < G0; ++n)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bq[0][n][pl] = frag(0, n, pl);
  }
  for (; t < tend; ++t) {
    const int row = t * 32 + ri;
    const int tload = t + 1 < tend ? t + 1 : t;                 // tile whose step s + 1 pieces replace the ones split now
    const int tload2 = tload + 1 < tend ? tload + 1 : tload;    // ... and the tile AFTER it for step 0 (split at step 12)
    const unsigned fl = fl_next;
    const float mrow = mk_next;
    {
      const int rown = min(tload * 32 + ri, M - 1);
      if (FL) fl_next = a.add_flag[rown];
      mk_next = a.mask[rown];
    }
    f32x16 acc[CTN];
#pragma unroll
    for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    f32x4 addv[CTN][4];
#pragma unroll
    for (int s = 0; s < kXKS; ++s) {
      bf16x8 ap[3];
      ap[0] = __builtin_bit_cast(bf16x8, (u32x4_x){sn0.hi.x, sn0.hi.y, sn1.hi.x, sn1.hi.y});
      ap[1] = __builtin_bit_cast(bf16x8, (u32x4_x){sn0.mid.x, sn0.mid.y, sn1.mid.x, sn1.mid.y});
      ap[2] = __builtin_bit_cast(bf16x8, (u32x4_x){sn0.lo.x, sn0.lo.y, sn1.lo.x, sn1.lo.y});
      const int sn = s + 1 < kXKS ? s + 1 : 0;                  // the step whose planes are prepared beside this one
      // ---- group 0: tiles 0, 1.  Beside it: fragments of group 1, first half of the next step's split ----
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < G1; ++n)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[1][n][pl] = frag(s, G0 + n, pl);
      {
        f32x4 x0 = ra[sn][0];
        if (sn == kXKS - 1) x0 = hi ? zero4 : x0;               // k 200..207 does not exist: the upper lanes multiply zeros
        sn0 = split3(x0);
      }
      if (s == 2) {                                             // the epilogue's operand: nbr in the accumulator layout
        int arow = min(row, M - 1);
        if (FL && fl == 0u) arow = M;
        const float* ap_ = a.add + (size_t)arow * D + col0 + 4 * hi;
#pragma unroll
        for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            addv[nt][g] = zero4;
            if (col0 + 32 * nt + 8 * g < D) addv[nt][g] = *reinterpret_cast<const f32x4*>(ap_ + 32 * nt + 8 * g);
          }
      }
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int n = 0; n < G0; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0][n][PB[p]], ap[PA[p]], acc[n], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 6 * G0; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kX1Valu, 0);
        if (i < 3 * G1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (s == 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
      // ---- group 1: the other tiles.  Beside it: fragments of the next step's group 0, second half of the split, refills ----
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < G0; ++n)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[0][n][pl] = frag(sn, n, pl);
      {
        f32x4 x1 = ra[sn][1];
        if (sn == kXKS - 1) x1 = hi ? zero4 : x1;
        sn1 = split3(x1);
        const int rt = sn == 0 ? tload2 : tload;                // (step 0's pieces were split for the NEXT tile just now)
        ra[sn][0] = a_piece(rt, sn, 0);
        ra[sn][1] = a_piece(rt, sn, 1);
      }
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int n = 0; n < G1; ++n)
          acc[G0 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[1][n][PB[p]], ap[PA[p]], acc[G0 + n], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 6 * G1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kX1Valu, 0);
        if (i < 3 * G0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the planes and fragments prepared for the NEXT tile's first step are pinned here: left alone, the compiler sinks
    // their computation behind the epilogue (they are only used after it), i.e. out of the MFMA shadow and behind a wait
    asm volatile("" : "+v"(sn0.hi.x), "+v"(sn0.hi.y), "+v"(sn0.mid.x), "+v"(sn0.mid.y), "+v"(sn0.lo.x), "+v"(sn0.lo.y));
    asm volatile("" : "+v"(sn1.hi.x), "+v"(sn1.hi.y), "+v"(sn1.mid.x), "+v"(sn1.mid.y), "+v"(sn1.lo.x), "+v"(sn1.lo.y));
#pragma unroll
    for (int n = 0; n < G0; ++n)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(bq[0][n][pl]));
    // epilogue from the registers (as form 1)
    float part = 0.f;
#pragma unroll
    for (int nt = 0; nt < CTN; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (col0 + 32 * nt + 8 * g < D) {
          const int cs = 32 * nt + 8 * g + 4 * hi;
          const f32x4 bias4 = *reinterpret_cast<const f32x4*>(Bl + cs);
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(Sl + cs);
          f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
          v = __builtin_elementwise_max((v + bias4) + addv[nt][g], zero4);
          if (row < M) store16_nowait(a.C + (size_t)row * D + col0 + cs, v);
          part += v[0] * ws4[0] + v[1] * ws4[1] + v[2] * ws4[2] + v[3] * ws4[3];
        }
      }
    const float tot = part + __shfl_xor(part, 32);
    if (hi == 0 && row < M) {
      const float share = first_part ? (tot + bs) + (1.0f - mrow) * kVeryNeg : tot;
      atomic_add_nowait(a.score + row, share);
    }
  }
}

template <bool FL>
__global__ __launch_bounds__(256, 1) void k_update_x1(const UpdXArgs a, const int nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int blk = blockIdx.x;
  const int h = (blk >> 3) & 1;
  const int chunk = (blk >> 4) * 8 + (blk & 7);
  if (chunk >= nchunks) return;
  if (h == 0) update_x1_part<kXCT0, FL>(a, lds, 0, true, chunk, nchunks);
  else update_x1_part<7 - kXCT0, FL>(a, lds, kXCT0 * 32, false, chunk, nchunks);
}

template <bool FL>
__global__ __launch_bounds__(512, 2) void k_update_x32(const UpdXArgs a, const int nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // the two parts of a row chunk are neighbours in the grid AND on one XCD (blocks b and b + 8): the second read of the
  // chunk's rows of h hits that XCD's L2.  block = 16 * (c / 8) + 8 * part + c % 8
  const int blk = blockIdx.x;
  const int h = (blk >> 3) & 1;
  const int chunk = (blk >> 4) * 8 + (blk & 7);
  if (chunk >= nchunks) return;
  if (h == 0) update_x32_part<kXCT0, FL>(a, lds, 0, true, chunk, nchunks);
  else update_x32_part<7 - kXCT0, FL>(a, lds, kXCT0 * 32, false, chunk, nchunks);
}

bool update_x32_shape_ok(int64_t BN, int32_t D, int32_t ldw) {
  return D == kXD && ldw % 4 == 0 && BN >= 8192 && (BN + 1) * (int64_t)D < ((int64_t)1 << 31);
}

// GNNRAG_UPDATE_X32 in the environment (read when the library is first used: A/B runs): 0 off, 1 form 1 (two waves per
// SIMD), 2 form 2 (one wave per SIMD, pipelined).  Default: GNNRAG_X32_DEFAULT.
#ifndef GNNRAG_X32_DEFAULT
#define GNNRAG_X32_DEFAULT 0
#endif
static int update_x32_form() {
  static const int form = [] {
    const char* e = getenv("GNNRAG_UPDATE_X32");
    return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : GNNRAG_X32_DEFAULT;
  }();
  return form;
}

int update_x32_launch_f(const float* h, const float* nbr, const uint8_t* add_flag, const float* W, const float* b,
                        const float* w_s, const float* b_s, const float* mask, float* h_out, float* score, int64_t BN,
                        int32_t D, int32_t ldw, hipStream_t stream, bool score_zeroed) {
  const int form = update_x32_form();
  if (form == 0 || !update_x32_shape_ok(BN, D, ldw)) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)h | (uintptr_t)nbr | (uintptr_t)W | (uintptr_t)h_out) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  UpdXArgs a;
  memset(&a, 0, sizeof(a));
  a.A = h; a.W = W; a.bias = b; a.add = nbr; a.w_s = w_s; a.b_s = b_s; a.mask = mask; a.C = h_out; a.score = score;
  a.add_flag = add_flag;
  a.M = (int32_t)BN; a.ldw = ldw;
  int cus = 0;
  GNNRAG_RC(device_cu_count(&cus));
  const long long U = (BN + 31) / 32;
  int chunks = cus / 2;
  if (chunks < 1) chunks = 1;
  if ((long long)chunks * 8 > U) chunks = (int)((U + 7) / 8);
  if (!score_zeroed) GNNRAG_HIP(hipMemsetAsync(score, 0, (size_t)BN * sizeof(float), stream));
  const int nblk = ((chunks + 7) / 8) * 16;
  if (form == 2) {
    static DeviceMask cap1, cap1_f;
    GNNRAG_RC(add_flag ? raise_lds_cap(k_update_x1<true>, cap1_f) : raise_lds_cap(k_update_x1<false>, cap1));
    if (add_flag) hipLaunchKernelGGL(k_update_x1<true>, dim3(nblk), dim3(256), 160 * 1024, stream, a, chunks);
    else hipLaunchKernelGGL(k_update_x1<false>, dim3(nblk), dim3(256), 160 * 1024, stream, a, chunks);
    GNNRAG_LAUNCH_CHECK();
    return 0;
  }
  static DeviceMask cap, cap_f;
  GNNRAG_RC(add_flag ? raise_lds_cap(k_update_x32<true>, cap_f) : raise_lds_cap(k_update_x32<false>, cap));
  if (add_flag) hipLaunchKernelGGL(k_update_x32<true>, dim3(nblk), dim3(512), 160 * 1024, stream, a, chunks);
  else hipLaunchKernelGGL(k_update_x32<false>, dim3(nblk), dim3(512), 160 * 1024, stream, a, chunks);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag
