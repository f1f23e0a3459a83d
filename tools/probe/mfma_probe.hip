// Micro-benchmarks behind the dense-kernel design decisions of round 4: what the bf16 matrix
// pipe of one SIMD sustains for the instruction mixes the bf16x3 kernels are made of.  Stand-alone (HIP runtime only):
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_probe tools/probe/mfma_probe.hip && tools/probe/mfma_probe
// Every kernel runs LOOPS iterations of a body of NM MFMAs per wave; reported: ns, TFLOP/s, shader cycles per MFMA and
// SIMD as measured with s_memtime (clock independent).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { F_NONE = 0, F_PKADD = 1, F_SUB2 = 2, F_SPLIT = 3, F_LDS = 4, F_LDS_SPLIT = 5, F_GLOAD = 6, F_CVT = 7 };

// NACC independent accumulators visited round robin (NACC = 1: one dependent chain; 2: the alternation the shipped
// kernels use inside a 6-product group); FILL: what sits between the MFMAs
template <int NACC, int FILL, int PER>
__global__ __launch_bounds__(512) void k_mfma16(const float* __restrict__ src, float* __restrict__ dst, int loops,
                                                long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 av = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4);
  f32x4 bv = *reinterpret_cast<const f32x4*>(src + 4096 + threadIdx.x * 4);
  bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
  f32x4 x = av, y = bv;                 // VALU filler state
  bf16x8 bb[2] = {b, b};                // LDS / global fillers: fragment read one step ahead of its use
  for (int i = threadIdx.x; i < 32768 / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = av;
  __syncthreads();
  int loff = lane * 16, goff = lane * 4;
  long long t0 = 0;
  if (cyc) t0 = __builtin_readcyclecounter();
  for (int it = 0; it < loops; ++it) {
    asm volatile("" : "+v"(loff), "+v"(goff));       // addresses opaque per iteration: the fragment reads stay in the loop
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      if (FILL == F_LDS || FILL == F_LDS_SPLIT || FILL == F_GLOAD)
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb[((m / PER) + 1) & 1], acc[m % NACC], 0, 0, 0);
      else
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % NACC], 0, 0, 0);
      if (m % PER == 0) {
        if (FILL == F_PKADD) {
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x)) : "v"(*reinterpret_cast<double*>(&y)));
        } else if (FILL == F_SUB2) {
          asm volatile("v_sub_f32 %0, %0, %2\n v_sub_f32 %1, %1, %3" : "+v"(x[0]), "+v"(x[1]) : "v"(y[0]), "v"(y[1]));
        } else if (FILL == F_CVT) {
          unsigned r;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[0]), "v"(x[1]));
          asm volatile("" :: "v"(r));
        } else if (FILL == F_SPLIT) {        // the per-pair work of the exact 3-way split with scalar subtractions
          unsigned h, lo16, hi16;
          asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n v_lshlrev_b32 %1, 16, %0\n v_and_b32 %2, 0xffff0000, %0\n"
                       "v_sub_f32 %3, %3, %1\n v_sub_f32 %4, %4, %2"
                       : "=&v"(h), "=&v"(lo16), "=&v"(hi16), "+v"(x[0]), "+v"(x[1]));
          asm volatile("" :: "v"(h));
        } else if (FILL == F_LDS || FILL == F_LDS_SPLIT) {
          f32x4 r = *reinterpret_cast<const f32x4*>(lds + loff + ((m * 1024) & 32767));
          bb[(m / PER) & 1] = __builtin_bit_cast(bf16x8, r);
          if (FILL == F_LDS_SPLIT) {
            unsigned h, lo16, hi16;
            asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n v_lshlrev_b32 %1, 16, %0\n v_and_b32 %2, 0xffff0000, %0\n"
                         "v_sub_f32 %3, %3, %1\n v_sub_f32 %4, %4, %2"
                         : "=&v"(h), "=&v"(lo16), "=&v"(hi16), "+v"(x[0]), "+v"(x[1]));
            asm volatile("" :: "v"(h));
          }
        } else if (FILL == F_GLOAD) {
          f32x4 r = *reinterpret_cast<const f32x4*>(src + goff + ((m * 256) & 8191));
          bb[(m / PER) & 1] = __builtin_bit_cast(bf16x8, r);
        }
      }
    }
  }
  if (cyc && threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = __builtin_readcyclecounter() - t0;
  f32x4 s = x + y;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s[0] == 1234.567f) dst[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
__global__ __launch_bounds__(512) void k_mfma32(const float* __restrict__ src, float* __restrict__ dst, int loops,
                                                long long* cyc) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f32x4 av = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4);
  f32x4 bv = *reinterpret_cast<const f32x4*>(src + 4096 + threadIdx.x * 4);
  bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
  long long t0 = 0;
  if (cyc) t0 = __builtin_readcyclecounter();
  for (int it = 0; it < loops; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
  }
  if (cyc && threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = __builtin_readcyclecounter() - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 1234.567f) dst[threadIdx.x] = s;
}

struct Case { const char* name; void (*fn)(const float*, float*, int, long long*); int threads; int mfma_per_loop; double flop_per_mfma; };

int main() {
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  float *src, *dst;
  long long* cyc;
  CHECK(hipMalloc(&src, 1 << 20));
  CHECK(hipMalloc(&dst, 1 << 20));
  CHECK(hipMalloc(&cyc, 64));
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 1000) - 0.5f;
  CHECK(hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice));
  const double F16 = 2.0 * 16 * 16 * 32, F32 = 2.0 * 32 * 32 * 16;
#define C16(name, NACC, FILL, PER, T) {name, k_mfma16<NACC, FILL, PER>, T, 48, F16}
  std::vector<Case> cases = {
    C16("16x16x32 8acc            1w/SIMD", 8, F_NONE, 1, 256), C16("16x16x32 8acc            2w/SIMD", 8, F_NONE, 1, 512),
    C16("16x16x32 2acc (dep dist 2) 1w", 2, F_NONE, 1, 256),     C16("16x16x32 2acc (dep dist 2) 2w", 2, F_NONE, 1, 512),
    C16("16x16x32 1acc (dep chain)  1w", 1, F_NONE, 1, 256),     C16("16x16x32 1acc (dep chain)  2w", 1, F_NONE, 1, 512),
    C16("16x16x32 4acc              1w", 4, F_NONE, 1, 256),     C16("16x16x32 4acc              2w", 4, F_NONE, 1, 512),
    C16("8acc + v_pk_add_f32 /mfma  1w", 8, F_PKADD, 1, 256),    C16("8acc + v_pk_add_f32 /mfma  2w", 8, F_PKADD, 1, 512),
    C16("8acc + 2 v_sub_f32 /mfma   1w", 8, F_SUB2, 1, 256),     C16("8acc + 2 v_sub_f32 /mfma   2w", 8, F_SUB2, 1, 512),
    C16("8acc + v_cvt_pk_bf16 /mfma 1w", 8, F_CVT, 1, 256),      C16("8acc + v_cvt_pk_bf16 /mfma 2w", 8, F_CVT, 1, 512),
    C16("8acc + split5 /mfma        1w", 8, F_SPLIT, 1, 256),    C16("8acc + split5 /mfma        2w", 8, F_SPLIT, 1, 512),
    C16("8acc + split5 /2 mfma      1w", 8, F_SPLIT, 2, 256),    C16("8acc + split5 /2 mfma      2w", 8, F_SPLIT, 2, 512),
    C16("8acc + ds_read_b128 /2mfma 1w", 8, F_LDS, 2, 256),      C16("8acc + ds_read_b128 /2mfma 2w", 8, F_LDS, 2, 512),
    C16("8acc + ds_read_b128 /mfma  1w", 8, F_LDS, 1, 256),      C16("8acc + ds_read_b128 /mfma  2w", 8, F_LDS, 1, 512),
    C16("8acc + lds + split5 /2mfma 1w", 8, F_LDS_SPLIT, 2, 256), C16("8acc + lds + split5 /2mfma 2w", 8, F_LDS_SPLIT, 2, 512),
    C16("2acc + lds + split5 /2mfma 2w", 2, F_LDS_SPLIT, 2, 512),
    C16("8acc + global 16B(L2) /4mfma 1w", 8, F_GLOAD, 4, 256),  C16("8acc + global 16B(L2) /4mfma 2w", 8, F_GLOAD, 4, 512),
    {"32x32x16 4acc              1w", k_mfma32<4>, 256, 24, F32}, {"32x32x16 4acc              2w", k_mfma32<4>, 512, 24, F32},
    {"32x32x16 1acc (dep chain)  1w", k_mfma32<1>, 256, 24, F32}, {"32x32x16 2acc              1w", k_mfma32<2>, 256, 24, F32},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int loops = 2000;
  printf("CUs %d; every case: one workgroup per CU, %d loops\n", cus, loops);
  printf("%-36s %9s %10s %12s %14s\n", "case", "us", "TFLOP/s", "cyc/MFMA/SIMD", "(by s_memtime)");
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(cyc, 0, 64));
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(c.fn, dim3(cus), dim3(c.threads), 32768, 0, src, dst, loops, rep ? cyc : nullptr);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      if (!rep) continue;
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      long long cy = 0;
      CHECK(hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost));
      const double waves_per_simd = c.threads / 256.0;
      const double n_mfma_simd = (double)loops * c.mfma_per_loop * waves_per_simd;
      const double tflops = n_mfma_simd * 4 * cus * c.flop_per_mfma / (ms * 1e-3) / 1e12;
      // s_memtime counts at a fixed 100 MHz on this family: report it raw next to the event time
      printf("%-36s %9.1f %10.1f %12.2f %14lld\n", c.name, ms * 1e3, tflops, ms * 1e-3 * 2.4e9 / n_mfma_simd, cy);
    }
  }
  return 0;
}
