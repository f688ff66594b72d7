// What does an instruction cost in the shadow of an fp32 MFMA?  One wave per SIMD (256-thread work-groups, one per CU), a loop of
// independent v_mfma_f32_16x16x4_f32 (33 accumulators) or v_mfma_f32_32x32x2_f32 (8 accumulators) with NF filler instructions
// after every MFMA: plain VALU (v_add_f32 on private registers), LDS reads (ds_read_b128) or global loads (L2-resident 1 KiB
// wave-loads).  Prints cycles per MFMA (s_memtime, median over work-groups).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/mfma_filler_probe tools/mfma_filler_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#ifndef NACC32
#define NACC32 8
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { F_VALU = 0, F_LDS = 1, F_GLOBAL = 2, F_VALU_DEP = 3 };

template <int MF, int NF, int KIND>
__global__ __launch_bounds__(256, 1) void k_probe(const float* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
  constexpr int NACC = MF == 16 ? 33 : NACC32;
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = src[i];
  __syncthreads();
  float a = src[tid], b = src[tid + 256];
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = src[tid + 512 + i];
  f32x4 ld[4] = {};
  typename std::conditional<MF == 16, f32x4, f32x16>::type acc[NACC] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      if constexpr (MF == 16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (KIND == F_VALU) v[(j * NF + f) & 7] += 1.0f;
        else if (KIND == F_VALU_DEP) v[0] = fmaf(v[0], 1.0001f, 0.5f);
        else if (KIND == F_LDS) { if (f == 0 && (j & 3) == 0) ld[(j >> 2) & 3] = *reinterpret_cast<const f32x4*>(lds + 4 * tid + 1024 * ((j >> 2) & 7)); }
        else { if (f == 0 && (j & 3) == 0) ld[(j >> 2) & 3] = *reinterpret_cast<const f32x4*>(src + 4 * tid + 1024 * ((it + j) & 63)); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += ld[i][0];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MF, int NF, int KIND>
int run(const char* label, const float* src, float* out, unsigned long long* st) {
  const int iters = 200, G = 256, NACC = MF == 16 ? 33 : NACC32;
  hipLaunchKernelGGL((k_probe<MF, NF, KIND>), dim3(G), dim3(256), 0, 0, src, out, st, iters);
  hipLaunchKernelGGL((k_probe<MF, NF, KIND>), dim3(G), dim3(256), 0, 0, src, out, st, iters);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(G);
  CHECK(hipMemcpy(h.data(), st, G * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-44s %6.1f cycles per MFMA (ideal %d)\n", label, (double)h[G / 2] / ((double)iters * NACC), MF == 16 ? 32 : 64);
  return 0;
}

int main() {
  float *src, *out;
  unsigned long long* st;
  CHECK(hipMalloc(&src, 1 << 20)); CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&st, 4096));
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CHECK(hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice));
#define R(MF, NF, KIND, label) if (run<MF, NF, KIND>(label, src, out, st)) return 1
  R(16, 0, F_VALU, "16x16x4 bare");
  R(16, 1, F_VALU, "16x16x4 + 1 v_add per MFMA");
  R(16, 2, F_VALU, "16x16x4 + 2 v_add per MFMA");
  R(16, 3, F_VALU, "16x16x4 + 3 v_add per MFMA");
  R(16, 4, F_VALU, "16x16x4 + 4 v_add per MFMA");
  R(16, 6, F_VALU, "16x16x4 + 6 v_add per MFMA");
  R(16, 2, F_VALU_DEP, "16x16x4 + 2 dependent v_fma per MFMA");
  R(16, 1, F_LDS, "16x16x4 + 1 ds_read_b128 per 4 MFMAs");
  R(16, 1, F_GLOBAL, "16x16x4 + 1 global_load_dwordx4 per 4 MFMAs");
  R(32, 0, F_VALU, "32x32x2 bare");
  R(32, 2, F_VALU, "32x32x2 + 2 v_add per MFMA");
  R(32, 4, F_VALU, "32x32x2 + 4 v_add per MFMA");
  R(32, 8, F_VALU, "32x32x2 + 8 v_add per MFMA");
  R(32, 12, F_VALU, "32x32x2 + 12 v_add per MFMA");
  R(32, 1, F_LDS, "32x32x2 + 1 ds_read_b128 per 4 MFMAs");
  R(32, 1, F_GLOBAL, "32x32x2 + 1 global_load_dwordx4 per 4 MFMAs");
  return 0;
}
