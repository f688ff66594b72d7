// v_mfma_f32_4x4x1_16B_f32 on gfx950: operand / result lane layout and the issue rate of a DEPENDENT chain (the guides carry
// no table for this shape).  D[b] (4 x 4) += A[b] (4 x 1) * B[b] (1 x 4) for 16 independent blocks b.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma4x4_layout_probe tools/mfma4x4_layout_probe.hip && tools/mfma4x4_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(float* out /* [64][4] */, int which) {
  const int lane = threadIdx.x;
  // which = 0: A = lane id, B = 1 -> D[lane][j] names the lane whose A operand lands there; which = 1: the same for B
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(which == 0 ? (float)lane : 1.f, which == 0 ? 1.f : (float)lane, acc, 0, 0, 0);
  for (int j = 0; j < 4; ++j) out[4 * lane + j] = acc[j];
}

template <int CHAINS>
__global__ void k_chain(float* out, unsigned long long* cyc, int n) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1e-3f * threadIdx.x, b = 1.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4 * 4); hipMalloc(&cyc, 8);
  float ha[256], hb[256];
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out, 0);
  hipMemcpy(ha, out, sizeof(ha), hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out, 1);
  hipMemcpy(hb, out, sizeof(hb), hipMemcpyDeviceToHost);
  printf("D[lane][j] = A[la] * B[lb]: (la, lb) per (lane, j)\n");
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int j = 0; j < 4; ++j) printf("  j%d: A lane %2d, B lane %2d", j, (int)(ha[4 * lane + j] + 0.5f), (int)(hb[4 * lane + j] + 0.5f));
    printf("\n");
    if (lane == 7) { printf("  ...\n"); lane = 59; }
  }
  unsigned long long c;
  const int n = 1000;
  hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("one dependent chain:      %.1f cycles (s_memtime ticks) per MFMA\n", (double)c / (8.0 * n));
  hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("two independent chains:   %.1f per MFMA\n", (double)c / (16.0 * n));
  hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("four independent chains:  %.1f per MFMA\n", (double)c / (32.0 * n));
  return 0;
}
