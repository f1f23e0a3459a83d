// Workgroup placement census (tools only): which workgroups of a persistent-size launch share a CU, and which wave
// slots they get.  hipcc --offload-arch=gfx950 -O2 -o census census.hip ; ./census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(512, 4) void k_census(unsigned* out, int spin) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // keep the workgroup resident for a while so that the whole grid is co-resident
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  lds[threadIdx.x] = x;
  __syncthreads();
  if (lds[(threadIdx.x + 1) & 511] == 12345.f) out[0] = 1;
}
int main() {
  const int G = 512;
  unsigned* d;
  hipMalloc(&d, G * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_census, dim3(G), dim3(512), 54272, 0, d, 200000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(G * 2);
  hipMemcpy(h.data(), d, G * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < G; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned wave = hw & 0xf, simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cuid;
    cu[key].push_back(b);
    if (b < 24 || (b >= 256 && b < 272)) printf("blk %3d hw %08x xcc %u se %u sh %u cu %2u simd %u wave %u\n", b, hw, xcc, se, sh, cuid, simd, wave);
  }
  printf("distinct CUs: %zu\n", cu.size());
  int shown = 0;
  std::map<int, int> delta;
  for (auto& kv : cu) {
    if (shown++ < 12) { printf("cu %06x:", kv.first); for (int b : kv.second) printf(" %d(w%u)", b, h[2 * b] & 0xf); printf("\n"); }
    if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
    else delta[-(int)kv.second.size()]++;
  }
  for (auto& kv : delta) printf("pair delta %d: %d CUs\n", kv.first, kv.second);
  return 0;
}
