// Row-stride probe for the gather walk (tools only; VERDICT round 5, item 4: "measure the stride fix instead of estimating it").
//
// BASELINE config 5's dense-prior walk gathers one 800-byte relation-table row per fact (D = 200 floats) from per-question
// tables of 6001 rows x 2 directions (k_walk_light_q: a wave's four 16-lane groups take one fact each, 16 lanes x float4 x
// 4 column groups).  PMC traffic of that launch is 1.46 x the algorithmic bytes, and the question was whether storing the
// rows on a 128-byte aligned stride (224 floats = 896 B) instead of 800 B would bring it down.  This probe runs exactly
// that access pattern - same lane mapping, same loads in flight, same table sizes, same number of facts per question, a
// question's workgroups on one XCD - on synthetic row ids, once per stride, so the two layouts are compared without
// touching the six kernels and the public P argument that a layout change crosses.
//
//   hipcc --offload-arch=gfx950 -O3 -o row_stride_probe row_stride_probe.hip
//   ./row_stride_probe                       (device time per launch, both strides, three row-id distributions)
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o fetch -- ./row_stride_probe     (HBM bytes per launch)
//
// Arithmetic: an 800-byte row starting at 32 k bytes into a 128-byte line (k = 0..3: 800 = 6 x 128 + 32) covers
// ceil((32 k + 800) / 128) = 7 lines for EVERY k - the unaligned layout never touches an eighth line - and the rows of a
// table are contiguous, so every fetched line is made of rows of that table only.  The aligned layout also touches 7
// lines per row, and makes the table 12 % larger.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one wave per node run of `per_node` facts; 4 waves per workgroup; workgroup -> question by blockIdx % Q (XCD-interleaved
// like the library's launch)
template <int U>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, const int2* __restrict__ facts, float* __restrict__ out,
                                                int stride, int rows_per_q, int nodes_per_q, int per_node, int Q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, l16 = lane & 15;
  const int q = blockIdx.x % Q;
  const int node0 = (blockIdx.x / Q) * 4 + wave;
  const int nblk = gridDim.x / Q;
  const float* T = tab + (size_t)q * 2 * rows_per_q * stride;
  for (int node = node0; node < nodes_per_q; node += nblk * 4) {
    const int2* f = facts + ((size_t)q * nodes_per_q + node) * per_node;
    f32x4 acc[4] = {};
    for (int j0 = 0; j0 < per_node; j0 += 4 * U) {
      int2 e[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + 4 * u + grp;
        e[u] = j < per_node ? f[j] : make_int2(-1, 0);
      }
      f32x4 t[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = l16 + 16 * m;                      // float4 index in the row: 50 of them at D = 200
          t[u][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (e[u].x >= 0 && c < 50) t[u][m] = *reinterpret_cast<const f32x4*>(T + (size_t)e[u].x * stride + 4 * c);
        }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float p = __int_as_float(e[u].y);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] += p * t[u][m];
      }
    }
    // the four groups' sums meet like the library's (xor tree over the group index), one 800-byte row is written
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[m][e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[m][e] = v;
      }
    if (grp == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = l16 + 16 * m;
        if (c < 50) *reinterpret_cast<f32x4*>(out + ((size_t)q * nodes_per_q + node) * 200 + 4 * c) = acc[m];
      }
    }
  }
}

int main() {
  const int Q = 32, R = 6001, NODES = 20000, PER = 8;      // 32 questions, 160 000 light facts per question and direction pair
  const int D = 200;
  std::mt19937 rng(7);
  struct Dist { const char* name; int kind; };
  const Dist dists[] = {{"uniform over the question's 2 x 6001 rows", 0}, {"zipf(1.3) over the rows (a few hot relations)", 1},
                        {"300 rows per question (WebQSP-like compaction)", 2}};
  float* out;
  CHECK(hipMalloc(&out, (size_t)Q * NODES * D * sizeof(float)));
  int2* facts;
  CHECK(hipMalloc(&facts, (size_t)Q * NODES * PER * sizeof(int2)));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (const Dist& d : dists) {
    std::vector<int2> h((size_t)Q * NODES * PER);
    std::uniform_int_distribution<int> uni(0, 2 * R - 1);
    std::vector<double> cdf;
    if (d.kind == 1) {
      double s = 0;
      for (int r = 1; r <= 2 * R; ++r) { s += 1.0 / std::pow((double)r, 1.3); cdf.push_back(s); }
      for (double& x : cdf) x /= s;
    }
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    for (size_t i = 0; i < h.size(); ++i) {
      int r;
      if (d.kind == 0) r = uni(rng);
      else if (d.kind == 1) r = (int)(std::lower_bound(cdf.begin(), cdf.end(), u01(rng)) - cdf.begin());
      else r = uni(rng) % 300 * 40 % (2 * R);
      if (r >= 2 * R) r = 2 * R - 1;
      const float p = 0.001f;
      h[i] = make_int2(r, *reinterpret_cast<const int*>(&p));
    }
    CHECK(hipMemcpy(facts, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
    for (int stride : {200, 224}) {
      float* tab;
      const size_t n = (size_t)Q * 2 * R * stride;
      CHECK(hipMalloc(&tab, n * sizeof(float) + 4096));
      CHECK(hipMemset(tab, 0, n * sizeof(float) + 4096));
      const int nblk = Q * 64;                              // 2048 workgroups of 4 waves: 8 per CU
      float best = 1e9f, sum = 0;
      const int reps = 12;
      for (int rep = 0; rep < reps + 2; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_gather<2>, dim3(nblk), dim3(256), 0, 0, tab, facts, out, stride, R, NODES, PER, Q);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2) { best = std::min(best, ms); sum += ms; }
      }
      const double rows = (double)Q * NODES * PER;
      printf("%-48s stride %3d floats (%3d B): %8.1f us per launch (best %8.1f), %5.2f TB/s of row bytes, table %6.1f MB\n", d.name, stride,
             stride * 4, 1e3 * sum / reps, 1e3 * best, rows * 800 / (sum / reps * 1e-3) / 1e12, n * 4 / 1e6);
      CHECK(hipFree(tab));
    }
  }
  return 0;
}
