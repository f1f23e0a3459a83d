// The question encoder's recurrence: a one-layer, one-direction LSTM over a handful of tokens
// (reference: gnn/modules/question_encoding/lstm_encoder.py:27-36 builds nn.LSTM(word_dim, entity_dim, batch_first=True)
// and calls it on [B, max_query_word, word_dim] with zero initial states, twice per forward - base_encoder.py:74-80 via
// rearev.py's init_reason and again via the instruction module's forward; relation texts go through the same call when
// --relation_word_emb is set).  MIOpen's RNN call costs ~12 ms at these shapes on the MI355X - two thirds of an
// evaluation batch's wall time at BASELINE config 2's hidden size - because the work is a chain of T tiny steps.
//
// Semantics = torch.nn.LSTM (gate order i, f, g, o):
//   g_t = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh;  c_t = sig(f) c_{t-1} + sig(i) tanh(g);  h_t = sig(o) tanh(c_t)
//
// One workgroup walks R sequences from the first token to the last: thread j owns gate row j (4 H threads), the
// weights are read transposed ([k][4H]: coalesced 16 H-byte rows, L2-resident - 1.6 MB at H = 200, E = 300), a weight
// element is used for the R sequences and (input part) for TC consecutive tokens at once; h lives in LDS, c in the
// registers of the first H threads.  Sums run over k in ascending order: one fixed order, fp32 throughout.
#include "gnnrag_common.h"

namespace gnnrag {

struct LstmArgs {
  const float* x;        // [B, T, E]
  const float* wih_t;    // [E, 4H]   (transposed copies in the workspace)
  const float* whh_t;    // [H, 4H]
  const float* b_ih;     // [4H] or null
  const float* b_hh;     // [4H] or null
  const float* h0;       // [B, H] or null (zeros)
  const float* c0;       // [B, H] or null
  float* out;            // [B, T, H]
  float* hn;             // [B, H]
  float* cn;             // [B, H]
  int32_t B, T, E, H;
};

// dst[k][j] = src[j][k]  (src [rows][cols])
__global__ __launch_bounds__(256) void k_lstm_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                        int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[tx][i];
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

constexpr int kLstmTC = 4;      // tokens whose input projections share one pass over W_ih

template <int R>
__global__ __launch_bounds__(1024) void k_lstm(const LstmArgs a) {
  extern __shared__ float smem[];
  const int H = a.H, E = a.E, T = a.T, G = 4 * H;
  float* s_h = smem;                 // [R][H]
  float* s_g = smem + R * H;         // [R][4H]
  const int j = threadIdx.x;
  const int b0 = blockIdx.x * R;
  const bool gate = j < G;
  const int jj = gate ? j : 0;
  float c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int b = min(b0 + r, a.B - 1);
    c[r] = (j < H && a.c0) ? a.c0[(size_t)b * H + j] : 0.f;
    if (j < H) s_h[r * H + j] = a.h0 ? a.h0[(size_t)b * H + j] : 0.f;
  }
  const float bias = gate ? (a.b_ih ? a.b_ih[jj] : 0.f) + (a.b_hh ? a.b_hh[jj] : 0.f) : 0.f;
  const float* __restrict__ wih = a.wih_t + jj;
  const float* __restrict__ whh = a.whh_t + jj;
  const float* xr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) xr[r] = a.x + (size_t)min(b0 + r, a.B - 1) * T * E;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += kLstmTC) {
    // input part of the gates of tokens t0 .. t0 + TC - 1 (clamped to the last token: computed, not used)
    float gin[kLstmTC][R];
#pragma unroll
    for (int tt = 0; tt < kLstmTC; ++tt)
#pragma unroll
      for (int r = 0; r < R; ++r) gin[tt][r] = bias;
    int toff[kLstmTC];
#pragma unroll
    for (int tt = 0; tt < kLstmTC; ++tt) toff[tt] = min(t0 + tt, T - 1) * E;
#pragma unroll 4
    for (int k = 0; k < E; ++k) {
      const float w = wih[(size_t)k * G];
#pragma unroll
      for (int tt = 0; tt < kLstmTC; ++tt)
#pragma unroll
        for (int r = 0; r < R; ++r) gin[tt][r] = fmaf(xr[r][toff[tt] + k], w, gin[tt][r]);     // (uniform address: scalar load)
    }
#pragma unroll
    for (int tt = 0; tt < kLstmTC; ++tt) {
      const int t = t0 + tt;
      if (t >= T) break;
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = gin[tt][r];
#pragma unroll 8
      for (int k = 0; k < H; ++k) {
        const float w = whh[(size_t)k * G];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fmaf(s_h[r * H + k], w, acc[r]);
      }
      if (gate) {
#pragma unroll
        for (int r = 0; r < R; ++r) s_g[r * G + j] = acc[r];
      }
      __syncthreads();
      if (j < H) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float* g = s_g + r * G;
          const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), og = sigmoidf_(g[3 * H + j]);
          c[r] = fmaf(fg, c[r], ig * gg);
          const float h = og * tanhf(c[r]);
          s_h[r * H + j] = h;
          if (b0 + r < a.B) a.out[((size_t)(b0 + r) * T + t) * H + j] = h;
        }
      }
      __syncthreads();
    }
  }
  if (j < H) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (b0 + r < a.B) {
        a.hn[(size_t)(b0 + r) * H + j] = s_h[r * H + j];
        a.cn[(size_t)(b0 + r) * H + j] = c[r];
      }
  }
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" size_t gnnrag_lstm_workspace_bytes(int32_t E, int32_t H) {
  if (E <= 0 || H <= 0) return 0;
  return align_up((size_t)(E + H) * 4 * (size_t)H * sizeof(float), 256);
}

extern "C" int gnnrag_lstm_forward(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                                   const float* b_hh, const float* h0, const float* c0, float* out, float* h_n,
                                   float* c_n, int32_t B, int32_t T, int32_t E, int32_t H, void* workspace,
                                   size_t workspace_bytes, gnnrag_stream_t stream_) {
  if (!x || !w_ih || !w_hh || !out || !h_n || !c_n || B <= 0 || T <= 0 || E <= 0 || H <= 0) return GNNRAG_E_BADARG;
  if (4 * H > 1024) return GNNRAG_E_UNSUPPORTED;            // one thread per gate row
  if (!workspace || workspace_bytes < gnnrag_lstm_workspace_bytes(E, H)) return GNNRAG_E_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int G = 4 * H;
  float* wih_t = (float*)workspace;
  float* whh_t = wih_t + (size_t)E * G;
  hipLaunchKernelGGL(k_lstm_transpose, dim3((E + 31) / 32, (G + 31) / 32), dim3(256), 0, stream, w_ih, wih_t, G, E);
  GNNRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_lstm_transpose, dim3((H + 31) / 32, (G + 31) / 32), dim3(256), 0, stream, w_hh, whh_t, G, H);
  GNNRAG_LAUNCH_CHECK();
  LstmArgs a;
  a.x = x; a.wih_t = wih_t; a.whh_t = whh_t; a.b_ih = b_ih; a.b_hh = b_hh; a.h0 = h0; a.c0 = c0;
  a.out = out; a.hn = h_n; a.cn = c_n; a.B = B; a.T = T; a.E = E; a.H = H;
  const int threads = (G + 63) / 64 * 64;
  if (B > 512) {            // many sequences (a relation vocabulary): four per workgroup share every weight element
    constexpr int R = 4;
    hipLaunchKernelGGL((k_lstm<R>), dim3((B + R - 1) / R), dim3(threads), (size_t)R * 5 * H * sizeof(float), stream, a);
  } else {
    hipLaunchKernelGGL((k_lstm<1>), dim3(B), dim3(threads), (size_t)5 * H * sizeof(float), stream, a);
  }
  GNNRAG_LAUNCH_CHECK();
  return 0;
}
