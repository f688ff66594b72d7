// Experiment: a barrier-free, LDS-free NT GEMM (C[M,N] = A[M,K] * B[N,K]^T, fp32 MFMA 32x32x2): every wave owns one
// 32x32 output tile and fetches its own fragments straight from global memory in the MFMA layout (two 16-byte
// loads per operand per 16-deep K-step), three register sets deep.  Twice the L2->register traffic of the 64x64
// LDS-tiled kernel, no s_barrier, no LDS.  hipcc --offload-arch=gfx950 -O3 -o tools/nt_direct_probe tools/nt_direct_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Frag { float4 a0, a1, b0, b1; };

template <int TNB>   // column tiles per block (waves per block)
__global__ __launch_bounds__(64 * TNB, 4) void nt_direct(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                          int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                          int tiles_m, int groups_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware: consecutive block ids go round-robin over 8 XCDs; give each XCD a contiguous range of row tiles
  const int nb = tiles_m * groups_n;
  const int q = nb >> 3, r = nb & 7, x = blockIdx.x & 7, s = blockIdx.x >> 3;
  const int logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
  const int tm = logical / groups_n, gn = logical - tm * groups_n;
  const int tn = gn * TNB + wave;
  if (tn * 32 >= N) return;
  const float* ap = A + (size_t)(tm * 32 + l31) * lda + 8 * lhi;
  const float* bp = B + (size_t)(tn * 32 + l31) * ldb + 8 * lhi;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nt = K / 16;
  auto ld = [&](Frag& f, int i) {
    const int k0 = (i < nt ? i : nt - 1) * 16;
    f.a0 = *reinterpret_cast<const float4*>(ap + k0);
    f.a1 = *reinterpret_cast<const float4*>(ap + k0 + 4);
    f.b0 = *reinterpret_cast<const float4*>(bp + k0);
    f.b1 = *reinterpret_cast<const float4*>(bp + k0 + 4);
  };
  auto mm = [&](const Frag& f) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.x, f.b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.y, f.b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.z, f.b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.w, f.b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.x, f.b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.y, f.b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.z, f.b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.w, f.b1.w, acc, 0, 0, 0);
  };
  Frag f0, f1, f2;
  ld(f0, 0); ld(f1, 1); ld(f2, 2);
  int i = 0;
  for (; i + 2 < nt; i += 3) {
    mm(f0); ld(f0, i + 3);
    mm(f1); ld(f1, i + 4);
    mm(f2); ld(f2, i + 5);
  }
  if (i < nt) mm(f0);
  if (i + 1 < nt) mm(f1);
  const int col = tn * 32 + l31, rbase = tm * 32 + 4 * lhi;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) C[(size_t)(rbase + (rr & 3) + 8 * (rr >> 2)) * ldc + col] = acc[rr];
}

int main(int argc, char** argv) {
  const int shapes[][3] = {{24576, 352, 352}, {24576, 352, 448}, {5440, 352, 352}, {2752, 352, 352}, {1024, 352, 352}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
    float *A, *B, *C;
    hipMalloc(&A, hA.size() * 4 + 65536); hipMalloc(&B, hB.size() * 4 + 65536); hipMalloc(&C, (size_t)M * N * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const int tiles_m = M / 32, TNB = 4, groups_n = (N / 32 + TNB - 1) / TNB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    for (int it = 0; it < reps + 3; ++it) {
      if (it == 3) hipEventRecord(e0);
      nt_direct<4><<<tiles_m * groups_n, 256>>>(A, K, B, K, C, N, M, N, K, tiles_m, groups_n);
    }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 200; ++t) {
      const int r = rand() % M, c = rand() % N;
      double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
      maxerr = fmax(maxerr, fabs(ref - hC[(size_t)r * N + c]));
    }
    const double us = ms * 1e3 / reps;
    printf("%6d x %4d x %4d: %7.1f us  %6.1f TF  (max abs err of 200 samples %.2e)\n", M, N, K, us, 2.0 * M * N * K / us / 1e6, maxerr);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
