// Calibration probe: what one wave per SIMD (and 2, 4 waves) sustains on v_mfma_f32_32x32x2_f32, and the
// shader clock while doing it.  hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long* stamps, float* sink) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f + threadIdx.x * 2e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  if (s == 12345.678f) sink[0] = s;
}

template <int NACC>
void run(int blocks, int iters) {
  unsigned long long* d; float* sink;
  hipMalloc(&d, blocks * 16); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC><<<blocks, 256>>>(iters, d, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC><<<blocks, 256>>>(iters, d, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> cyc, ghz;
  for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[2*b] / ((double)iters * NACC)); ghz.push_back((double)h[2*b] / (double)h[2*b+1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  double flops = (double)blocks * 4 * iters * NACC * 32.0 * 32 * 2 * 2;
  printf("nacc %d blocks %5d iters %6d: %8.1f us  %6.1f TF  cycles/MFMA median %.1f  clock median %.2f GHz\n", NACC, blocks, iters,
         ms * 1e3, flops / (ms * 1e-3) / 1e12, cyc[blocks / 2], ghz[blocks / 2]);
  hipFree(d); hipFree(sink);
}

int main() {
  for (int it : {200, 2000, 20000}) {
    run<1>(256, it); run<2>(256, it); run<4>(256, it);
    run<2>(512, it); run<2>(1024, it); run<4>(1024, it);
  }
  return 0;
}
