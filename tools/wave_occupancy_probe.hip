// Does a second and third wave per SIMD hide what one wave per SIMD exposes?  The inner loop of the row-owned forward in
// miniature: per "K-step" a wave issues NM v_mfma_f32_16x16x4_f32 (independent accumulators), NL global wave-loads of 1 KiB
// (L2-resident), one ds_read_b128 and NV VALU instructions, spread one piece per few MFMAs.  Same total work per SIMD in both
// shapes: 1 wave x (132 MFMA, 11 loads, 36 VALU) against 3 waves x (44 MFMA, 11 loads, 12 VALU) - the 3-wave shape issues three
// times the loads.  Prints cycles per K-step-equivalent of a SIMD (4224 = the MFMAs alone).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/wave_occupancy_probe tools/wave_occupancy_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int WAVES, int NT, int NL, int NV>   // waves per SIMD, accumulator tiles per wave, loads and VALU per wave and K-step
__global__ __launch_bounds__(256 * WAVES, 1) void k_probe(const float* __restrict__ w, float* __restrict__ out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = w[i];
  __syncthreads();
  f32x4 acc[NT] = {};
  constexpr int NB = NL > 0 ? NL : 1;
  f32x4 b[NB] = {}, bn[NB] = {};
  f32x4 a = *reinterpret_cast<const f32x4*>(lds + 4 * (tid & 255));
  const float* wp = w + 4 * (tid & 63) + 16384 * (tid >> 6);
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto kstep = [&](f32x4 (&bc)[NB], f32x4 (&bnx)[NB], int it) __attribute__((always_inline)) {
    constexpr int NSLOT = 4 * NT, STRIDE = NSLOT / (NL + 2) > 0 ? NSLOT / (NL + 2) : 1;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int s = e * NT + t;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bc[t % NB][e], acc[t], 0, 0, 0);
        if (s % STRIDE == 0) {
          const int p = s / STRIDE;
          if (p < NL) {
            typedef __attribute__((address_space(1))) const f32x4 gq;   // a global (not flat) load, pinned at this slot
            unsigned off = (unsigned)(((it + 1) & 15) * 256 * NB + p * 256) * 4u;
            asm volatile("" : "+v"(off));
            bnx[p] = *reinterpret_cast<gq*>((__attribute__((address_space(1))) const char*)(unsigned long long)wp + off);
          } else if (p == NL) {
            int o = 4 * ((tid + it) & 255);
            asm volatile("" : "+v"(o));
            a = *reinterpret_cast<const f32x4*>(lds + o);
          } else if (p == NL + 1) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 3] = fmaf(v[k & 3], 1.0001f, 0.5f);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  for (int it = 0; it < iters; it += 2) {
    kstep(b, bn, it);
    kstep(bn, b, it + 1);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
  for (int t = 0; t < NT; ++t) s += acc[t][0];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int WAVES, int NT, int NL, int NV>
int run(const char* label, const float* w, float* out, unsigned long long* st) {
  const int iters = 2000, G = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_probe<WAVES, NT, NL, NV>), dim3(G), dim3(256 * WAVES), 0, 0, w, out, st, iters);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_probe<WAVES, NT, NL, NV>), dim3(G), dim3(256 * WAVES), 0, 0, w, out, st, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // whole-kernel time (every wave of a SIMD done) at 2.4 GHz
  printf("%-64s %7.0f cycles per SIMD K-step at 2.4 GHz (%d MFMAs = %d cycles)\n", label, ms * 1e-3 * 2.4e9 / iters, 4 * NT * WAVES, 32 * 4 * NT * WAVES);
  return 0;
}

int main() {
  float *w, *out;
  unsigned long long* st;
  CHECK(hipMalloc(&w, 64 << 20)); CHECK(hipMalloc(&out, 4 << 20)); CHECK(hipMalloc(&st, 4096));
  CHECK(hipMemset(w, 0, 64 << 20));
  if (run<1, 33, 11, 0>("1 wave/SIMD : 132 MFMA + 11 loads + 1 LDS read", w, out, st)) return 1;
  if (run<1, 33, 11, 36>("1 wave/SIMD : 132 MFMA + 11 loads + 1 LDS read + 36 VALU", w, out, st)) return 1;
  if (run<3, 11, 11, 0>("3 waves/SIMD: 44 MFMA + 11 loads + 1 LDS read each", w, out, st)) return 1;
  if (run<3, 11, 11, 12>("3 waves/SIMD: 44 MFMA + 11 loads + 1 LDS read + 12 VALU each", w, out, st)) return 1;
  if (run<2, 11, 11, 12>("2 waves/SIMD: 44 MFMA + 11 loads + 1 LDS read + 12 VALU each", w, out, st)) return 1;
  if (run<1, 33, 0, 0>("1 wave/SIMD : 132 MFMA + 1 LDS read", w, out, st)) return 1;
  return 0;
}
