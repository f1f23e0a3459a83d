// The self-block update of the fused path with the WEIGHTS IN REGISTERS (round 4; gfx950, hidden size 200):
//
//   h'[m, :] = relu( h[m, :] . W_e2e[:, 0:D]^T + b + nbr[m, :] ),   score[m] = w_s . h'[m, :] + b_s + (1 - mask[m]) * -1e11
//
// (reasongnn.py:161-168 with the neighbour blocks already reduced into nbr; same bf16x3 arithmetic as k_update_b3 in
// tables_b3.hip: exact 3-way split of both operands, six plane products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
//
// Why another kernel.  PMC of k_update_b3 at C2 (profiles/r04a_pmc_dense_layer_C2.txt): the matrix pipe is busy 47 % of
// the waves' lifetime.  Its waves are W-stationary through LDS: every wave reads 147 weight fragments from LDS and splits
// the 3200 elements of its A tile itself (2.4 VALU + 0.5 ds_read per MFMA), and a row of h is split TWICE, once per column
// part, because the three planes of all 200 columns (240 KB) do not fit a CU's LDS.  More waves per SIMD made it slower
// (k_update_b3w: 99 / 105 us with 12 / 16 waves against 91).  What is left is fewer instructions per MFMA:
//   * a CU's register file (512 KB) is larger than its LDS: the hi and mid planes of W live in REGISTERS, distributed over
//     the 8 waves of the one workgroup per CU by column tile (a wave holds 2 column tiles' fragments: 112 VGPRs); only the
//     lo plane (one of the six products) is read from LDS (91 KB, fragment-ordered blocks: conflict free, no padding);
//   * a 16-row tile of h is split ONCE per workgroup, by all threads together (800 float4 pieces over 512 threads), into a
//     3-slot ring of fragment-ordered planes in LDS (21 KB per tile); every wave reads the tile's 21 plane fragments from
//     there and multiplies them against its own column tiles: 0.25 ds_read and ~0.3 VALU per MFMA;
//   * 13 column tiles over 4 SIMDs: each SIMD's two waves own 3 tiles (2 + 1) and take the 13th (columns 192..199) for
//     every 4th row tile - 136.5 MFMAs per SIMD and row tile on all four;
//   * the operands are swapped (A operand = W, B operand = h), so a lane's accumulator holds FOUR CONSECUTIVE COLUMNS of
//     one row: nbr is read and h' written as one 16-byte piece per lane and column tile;
//   * every row's score is complete inside the workgroup (all 200 columns): the waves' partial dots meet in LDS and are
//     summed in wave order after the tile's barrier - no atomics, no zeroed score buffer, one fixed order.
// One barrier per row tile; raw rows of h are requested four tiles ahead, nbr two.
#include "gnnrag_common.h"
#include "dense_internal.h"
#include <type_traits>

namespace gnnrag {

#ifndef GNNRAG_UPDATE_WR
#define GNNRAG_UPDATE_WR 1
#endif

constexpr int kWrD = 200;                 // the hidden size this kernel is written for (12 full column tiles + 8 columns)
constexpr int kWrKB = 7;                  // k blocks of 32 (k >= 200 is zero padding)
constexpr int kWrTiles = 13;
constexpr int kWrBlock = 1024;            // one (plane, k block) fragment block: 64 lanes x 16 B
constexpr int kWrPlane = kWrKB * kWrBlock;
constexpr int kWrSlot = 3 * kWrPlane;     // the three planes of one 16-row tile of h
constexpr int kWrSlots = 3;
constexpr int kWrOffWlo = kWrSlots * kWrSlot;                         // lo plane of W: [13 tiles][7][1024]
constexpr int kWrOffScore = kWrOffWlo + kWrTiles * kWrPlane;          // [2][8 waves][16 rows] floats
constexpr int kWrOffBias = kWrOffScore + 2 * 8 * 16 * 4;              // [208] floats
constexpr int kWrOffWs = kWrOffBias + 208 * 4;                        // [208] floats
constexpr int kWrLds = kWrOffWs + 208 * 4;
static_assert(kWrLds <= 160 * 1024, "LDS map of k_update_wr");
constexpr int kWrPieces = 16 * (kWrD / 4);                            // float4 pieces of one tile of h: 800

struct UpdWrArgs {
  const float* A;        // h [M, D]
  const float* W;        // e2e_linear.weight [D, ldw]; columns 0..D-1 are the self block
  const float* bias;     // [D] or null
  const float* add;      // nbr [M (+1 zero row when gated), D]
  const float* w_s;      // [D]
  const float* b_s;      // [1]
  const float* mask;     // [M]
  float* C;              // [M, D]
  float* score;          // [M]
  const uint8_t* add_flag;   // FL: row gates of `add`
  int32_t M, ldw;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 consecutive k of one weight / node row as the three 16-byte bf16 fragments of the exact 3-way split
struct Frag3 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Frag3 split8(f32x4 x0, f32x4 x1) {
  const Split3 s0 = split3(x0), s1 = split3(x1);
  Frag3 f;
  f.hi = __builtin_bit_cast(bf16x8, (u32x4){s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y});
  f.mid = __builtin_bit_cast(bf16x8, (u32x4){s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y});
  f.lo = __builtin_bit_cast(bf16x8, (u32x4){s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y});
  return f;
}

template <bool FL>
__global__ __launch_bounds__(512, 1) void k_update_wr(const UpdWrArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int D = kWrD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int sp = wave & 3;                       // waves sp and sp + 4 share a SIMD (cyclic wave placement)
  const bool wb = wave >= 4;
  // column tiles of this wave's two slots: A waves (0..3) own tiles 3 sp, 3 sp + 1; B waves own 3 sp + 2 and take the
  // 13th tile (columns 192..199) for the row tiles t with t % 4 == sp
  const int ct[2] = {wb ? 3 * sp + 2 : 3 * sp, wb ? 12 : 3 * sp + 1};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* addb = reinterpret_cast<const unsigned char*>(a.add);
  unsigned char* Cb = reinterpret_cast<unsigned char*>(a.C);
  const int M = a.M;

  // ---- prologue ---------------------------------------------------------------------------------------------------
  for (int x = tid * 16; x < kWrOffWlo; x += 512 * 16) *reinterpret_cast<f32x4*>(lds + x) = zero4;   // ring incl. k padding
  {
    float* Bl = reinterpret_cast<float*>(lds + kWrOffBias);
    float* Sl = reinterpret_cast<float*>(lds + kWrOffWs);
    for (int j = tid; j < 208; j += 512) {
      Bl[j] = (j < D && a.bias) ? a.bias[j] : 0.f;
      Sl[j] = j < D ? a.w_s[j] : 0.f;
    }
  }
  // the lo plane of W for all 13 column tiles, fragment ordered: block (tile c, k block kb), entry l = fg * 16 + fr holds
  // lo(W[16 c + fr, 32 kb + 8 fg .. + 7])
  for (int f = tid; f < kWrTiles * kWrKB * 64; f += 512) {
    const int l = f & 63, blk = f >> 6;
    const int c = blk / kWrKB, kb = blk - c * kWrKB;
    const int n = c * 16 + (l & 15), k0 = kb * 32 + (l >> 4) * 8;
    f32x4 x0 = zero4, x1 = zero4;
    if (n < D && k0 < D) {
      const float* wp = a.W + (size_t)n * a.ldw + k0;
      x0 = *reinterpret_cast<const f32x4*>(wp);
      x1 = *reinterpret_cast<const f32x4*>(wp + 4);
    }
    const Frag3 fw = split8(x0, x1);
    *reinterpret_cast<bf16x8*>(lds + kWrOffWlo + f * 16) = fw.lo;
  }
  __syncthreads();     // the ring is zeroed before anybody writes a piece into it
  // hi / mid planes of this wave's two column tiles -> registers (A operand: row = output column 16 c + fr, k = 8 fg ..)
  bf16x8 wh[2][kWrKB], wm[2][kWrKB];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int n = ct[s] * 16 + fr;
#pragma unroll
    for (int kb = 0; kb < kWrKB; ++kb) {
      const int k0 = kb * 32 + fg * 8;
      f32x4 x0 = zero4, x1 = zero4;
      if (n < D && k0 < D) {
        const float* wp = a.W + (size_t)n * a.ldw + k0;
        x0 = *reinterpret_cast<const f32x4*>(wp);
        x1 = *reinterpret_cast<const f32x4*>(wp + 4);
      }
      const Frag3 fw = split8(x0, x1);
      wh[s][kb] = fw.hi;
      wm[s][kb] = fw.mid;
    }
  }

  // this workgroup's row tiles
  const long long U = ((long long)M + 15) >> 4;
  const int t0 = (int)(U * blockIdx.x / gridDim.x), t1 = (int)(U * (blockIdx.x + 1) / gridDim.x);

  // producer side: a thread's (up to) two float4 pieces of a tile: piece q = (row m = q / 50, float4 c4 = q % 50)
  int pm[2], pc[2], plds[2];
  bool pv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = tid + 512 * j;
    pv[j] = q < kWrPieces;
    pm[j] = pv[j] ? q / (D / 4) : 0;
    pc[j] = pv[j] ? q - pm[j] * (D / 4) : 0;
    const int k = 4 * pc[j];
    // entry of the piece inside a plane: block kb = k / 32, lane (k % 32 / 8) * 16 + m, half (k / 4) & 1
    plds[j] = (k >> 5) * kWrBlock + ((((k & 31) >> 3) * 16 + pm[j]) << 4) + ((k >> 2) & 1) * 8;
  }
  auto raw_load = [&](int t, int j) -> f32x4 {
    const unsigned row = (unsigned)min(t * 16 + pm[j], M - 1);
    return *reinterpret_cast<const f32x4*>(Ab + (size_t)(row * (unsigned)(D * 4) + (unsigned)pc[j] * 16u));
  };
  auto put_piece = [&](int slot, int j, f32x4 v) {
    const Split3 s3 = split3(v);
    unsigned char* dst = lds + slot * kWrSlot + plds[j];
    *reinterpret_cast<uint2*>(dst) = s3.hi;
    *reinterpret_cast<uint2*>(dst + kWrPlane) = s3.mid;
    *reinterpret_cast<uint2*>(dst + 2 * kWrPlane) = s3.lo;
  };
  // consumer side: nbr piece of (tile t, slot s): row t * 16 + fr, columns 16 ct[s] + 4 fg .. + 3
  const int colb[2] = {ct[0] * 16 + 4 * fg, ct[1] * 16 + 4 * fg};
  const bool colok[2] = {colb[0] < D, colb[1] < D};
  auto slot1_on = [&](int t) -> bool { return !wb || (t & 3) == sp; };       // wave-uniform
  auto gate_load = [&](int t) -> unsigned {
    if (!FL) return 1u;
    return a.add_flag[min(t * 16 + fr, M - 1)];
  };
  auto nbr_load = [&](int t, int s, unsigned gate) -> f32x4 {
    unsigned row = (unsigned)min(t * 16 + fr, M - 1);
    if (FL && gate == 0u) row = (unsigned)M;                 // not a frontier row: the zero row behind the buffer
    const unsigned col = (unsigned)(colok[s] ? colb[s] : 0);
    return *reinterpret_cast<const f32x4*>(addb + (size_t)(row * (unsigned)(D * 4) + col * 4u));
  };

  // fill: tiles t0, t0 + 1 into slots 0, 1; raw rows of t0 + 2, t0 + 3 and nbr of t0, t0 + 1 on their way.
  // Everything that is loaded ahead lives in TWO STATIC STAGES (stage = tile parity relative to t0) and the tile loop is
  // unrolled by two: a register that still waits for its load is never moved (a v_mov of an in-flight register makes the
  // compiler wait for ALL outstanding loads - the first version of this kernel rotated its stages with moves and spent a
  // full HBM round trip per tile at an s_waitcnt vmcnt(0): 107 us instead of k_update_b3's 90).
  f32x4 raw[2][2];                 // [stage][piece]: raw pieces of tile t + 2 (stage of t) - written to the ring at step t
  f32x4 nb[2][2];                  // [stage][slot]: nbr pieces of tile t
  unsigned gt[2] = {1u, 1u};       // [stage]: row gates of tile t + 2 (FL)
  float mk[2];                     // [stage]: mask of row 16 t + (lane & 15)
  {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 v0 = raw_load(t0, j), v1 = raw_load(t0 + 1, j);
      raw[0][j] = raw_load(t0 + 2, j);
      raw[1][j] = raw_load(t0 + 3, j);
      if (pv[j]) {
        put_piece(0, j, v0);
        put_piece(1, j, v1);
      }
    }
    const unsigned g0 = gate_load(t0), g1 = gate_load(t0 + 1);
    gt[0] = gate_load(t0 + 2);
    gt[1] = gate_load(t0 + 3);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      nb[0][s] = nbr_load(t0, s, g0);
      nb[1][s] = nbr_load(t0 + 1, s, g1);
    }
    mk[0] = a.mask[min(t0 * 16 + fr, M - 1)];
    mk[1] = a.mask[min((t0 + 1) * 16 + fr, M - 1)];
  }
  const float bs = a.b_s[0];
  __syncthreads();

  int slot = 0;                                              // ring slot of tile t
  auto step = [&](int t, auto stage) {
    constexpr int S = decltype(stage)::value;
    // ---- producer: tile t + 2 -> slot (t + 2) % 3 (read last as tile t - 1, behind the previous barrier) ----------
    {
      const int sw = slot == 0 ? 2 : slot - 1;               // (slot + 2) % 3
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (pv[j]) put_piece(sw, j, raw[S][j]);              // (a tile past the end: written, never read)
        raw[S][j] = raw_load(t + 4, j);                      // (clamped rows past the end: loaded, never used)
      }
    }
    // ---- the tile's plane products -----------------------------------------------------------------------------------
    f32x4 acc[2] = {zero4, zero4};
    const bool s1 = slot1_on(t);
    {
      const unsigned char* ab = lds + slot * kWrSlot + lane * 16;
      const unsigned char* l0 = lds + kWrOffWlo + ct[0] * kWrPlane + lane * 16;
      const unsigned char* l1 = lds + kWrOffWlo + ct[1] * kWrPlane + lane * 16;
      // one straight-line block per case (a branch per k block would end the scheduling region: the next k block's
      // fragment reads could not move above this one's MFMAs); fragments of k block kb + 1 are requested before the
      // products of kb
      auto products = [&](auto two) {
        constexpr bool TWO = decltype(two)::value;
        bf16x8 hh = *reinterpret_cast<const bf16x8*>(ab);
        bf16x8 hm = *reinterpret_cast<const bf16x8*>(ab + kWrPlane);
        bf16x8 hl = *reinterpret_cast<const bf16x8*>(ab + 2 * kWrPlane);
        bf16x8 wl0 = *reinterpret_cast<const bf16x8*>(l0);
        bf16x8 wl1 = wl0;
        if (TWO) wl1 = *reinterpret_cast<const bf16x8*>(l1);
#pragma unroll
        for (int kb = 0; kb < kWrKB; ++kb) {
          const bf16x8 chh = hh, chm = hm, chl = hl, cw0 = wl0, cw1 = wl1;
          if (kb + 1 < kWrKB) {
            hh = *reinterpret_cast<const bf16x8*>(ab + (kb + 1) * kWrBlock);
            hm = *reinterpret_cast<const bf16x8*>(ab + kWrPlane + (kb + 1) * kWrBlock);
            hl = *reinterpret_cast<const bf16x8*>(ab + 2 * kWrPlane + (kb + 1) * kWrBlock);
            wl0 = *reinterpret_cast<const bf16x8*>(l0 + (kb + 1) * kWrBlock);
            if (TWO) wl1 = *reinterpret_cast<const bf16x8*>(l1 + (kb + 1) * kWrBlock);
          }
          // (W plane, h plane), smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi; the two slots'
          // chains alternate
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[0][kb], chm, acc[0], 0, 0, 0);
          if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[1][kb], chm, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw0, chh, acc[0], 0, 0, 0);
          if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw1, chh, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[0][kb], chl, acc[0], 0, 0, 0);
          if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[1][kb], chl, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[0][kb], chh, acc[0], 0, 0, 0);
          if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[1][kb], chh, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[0][kb], chm, acc[0], 0, 0, 0);
          if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[1][kb], chm, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[0][kb], chh, acc[0], 0, 0, 0);
          if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[1][kb], chh, acc[1], 0, 0, 0);
        }
      };
      if (s1) products(std::true_type{});
      else products(std::false_type{});
    }
    // ---- epilogue: a lane holds row t * 16 + fr, columns colb[s] .. + 3 of each of its slots --------------------------
    {
      const int row = t * 16 + fr;
      float part = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + kWrOffBias + colb[s] * 4);
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(lds + kWrOffWs + colb[s] * 4);
        f32x4 v = __builtin_elementwise_max((acc[s] + b4) + nb[S][s], zero4);
        const bool live = colok[s] && (s == 0 || s1);
        if (!live) v = zero4;
        else if (row < M && t < t1) *reinterpret_cast<f32x4*>(Cb + (size_t)((unsigned)row * (unsigned)(D * 4) + (unsigned)colb[s] * 4u)) = v;
        part += v[0] * w4[0] + v[1] * w4[1] + v[2] * w4[2] + v[3] * w4[3];
      }
      part += __shfl_xor(part, 16, 64);                      // the four column groups of the wave's tiles
      part += __shfl_xor(part, 32, 64);
      if (fg == 0) reinterpret_cast<float*>(lds + kWrOffScore)[((t & 1) * 8 + wave) * 16 + fr] = part;
      // nbr of tile t + 2 into the stage just used, its gates were requested two tiles ago (unconditional loads: a
      // slot that is not multiplied at t + 2 loads its piece and ignores it)
#pragma unroll
      for (int s = 0; s < 2; ++s) nb[S][s] = nbr_load(t + 2, s, gt[S]);
      gt[S] = gate_load(t + 4);
    }
    __syncthreads();
    // ---- the tile's scores: wave t % 8, partial dots in wave order ---------------------------------------------------
    if ((t & 7) == wave && lane < 16) {
      const float* sp_ = reinterpret_cast<const float*>(lds + kWrOffScore) + (t & 1) * 8 * 16 + lane;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += sp_[w * 16];
      const int r = t * 16 + lane;
      if (r < M && t < t1) a.score[r] = (tot + bs) + (1.0f - mk[S]) * kVeryNeg;
    }
    // (requested only after its last use: the stage register then carries over the back edge without a copy - a copy of
    // an in-flight register would cost an s_waitcnt vmcnt(0) per trip)
    mk[S] = a.mask[min((t + 2) * 16 + fr, M - 1)];
    slot = slot == 2 ? 0 : slot + 1;
  };
  // two tiles per trip, both steps unconditional (an odd tile count runs one step on the tile behind the range: its
  // loads are clamped, its stores switched off) - a conditional second step would put the stages behind phi nodes again
  for (int t = t0; t < t1; t += 2) {
    step(t, std::integral_constant<int, 0>{});
    step(t + 1, std::integral_constant<int, 1>{});
  }
}

bool update_wr_shape_ok(int64_t BN, int32_t D, int32_t ldw) {
  // hidden size 200, 4-byte offsets of (BN + 1) rows, enough row tiles to give every CU a few
  return GNNRAG_UPDATE_WR && D == kWrD && ldw % 4 == 0 && BN >= 8192 && (BN + 1) * (int64_t)D * 4 < ((int64_t)1 << 32);
}

// MEASURED AND NOT THE DEFAULT (C2, same box, alternating): 103-107 us against k_update_b3's 90-92 (profiles/r04b_*;
// DESIGN.md Appendix A.7).  Opt-in with GNNRAG_UPDATE_WR=1 in the environment (read when the library is first used);
// parity-tested in a process of its own (tests/test_gpu_round3_shapes.py).
static bool update_wr_enabled() {
  static const bool on = [] {
    const char* e = getenv("GNNRAG_UPDATE_WR");
    return e && e[0] == '1';
  }();
  return on;
}

// same contract as update_b3_launch_f (tables_b3.hip); `score` needs no zeroing for this kernel
int update_wr_launch_f(const float* h, const float* nbr, const uint8_t* add_flag, const float* W, const float* b,
                       const float* w_s, const float* b_s, const float* mask, float* h_out, float* score, int64_t BN,
                       int32_t D, int32_t ldw, hipStream_t stream) {
  if (!update_wr_enabled() || !update_wr_shape_ok(BN, D, ldw)) return GNNRAG_E_UNSUPPORTED;
  if ((((uintptr_t)h | (uintptr_t)nbr | (uintptr_t)W | (uintptr_t)h_out) & 15) != 0) return GNNRAG_E_UNSUPPORTED;
  UpdWrArgs a;
  memset(&a, 0, sizeof(a));
  a.A = h; a.W = W; a.bias = b; a.add = nbr; a.w_s = w_s; a.b_s = b_s; a.mask = mask; a.C = h_out; a.score = score;
  a.add_flag = add_flag;
  a.M = (int32_t)BN; a.ldw = ldw;
  int cus = 0;
  GNNRAG_RC(device_cu_count(&cus));
  const long long U = (BN + 15) / 16;
  int grid = cus;
  if (grid > U / 4) grid = (int)(U / 4 > 0 ? U / 4 : 1);
  static DeviceMask cap, cap_f;
  if (add_flag) {
    GNNRAG_RC(raise_lds_cap(k_update_wr<true>, cap_f));
    hipLaunchKernelGGL(k_update_wr<true>, dim3(grid), dim3(512), kWrLds, stream, a);
  } else {
    GNNRAG_RC(raise_lds_cap(k_update_wr<false>, cap));
    hipLaunchKernelGGL(k_update_wr<false>, dim3(grid), dim3(512), kWrLds, stream, a);
  }
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnnrag
