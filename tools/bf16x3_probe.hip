// Experiment: fp32 NT GEMM (C[M,N] = A[M,K] * B[N,K]^T) through the bf16 matrix pipe.  Every fp32 operand is split
// EXACTLY into three bf16 terms (x = x0 + x1 + x2, 8 significant bits each, by truncation and exact subtraction) on its
// way into LDS; a product a*b is then a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 (the three dropped terms are below
// 2^-24 of |a||b|), six v_mfma_f32_32x32x16_bf16 with fp32 accumulation instead of eight v_mfma_f32_32x32x2_f32 per
// 32x32x16 block: 192 matrix-pipe cycles instead of 512.  Prints time and the error against an fp64 product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bf16x3_probe tools/bf16x3_probe.hip && tools/bf16x3_probe [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;

constexpr int BM = 64, BN = 64, BK = 32, NTHREADS = 256;
constexpr int RS = BK + 8;                 // plane row stride in bf16 (80 bytes: 16-byte aligned rows)
constexpr int PLANE = 64 * RS;             // one plane of one operand (bf16 elements)
constexpr int STAGE = 2 * 3 * PLANE;       // A planes 0..2, B planes 0..2

__device__ __forceinline__ void split3(float x, u32& h0, u32& h1, u32& h2) {
  h0 = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h0);
  h1 = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(h1);
  h2 = __float_as_uint(r2);                // <= 8 significant bits left: its upper half is exact
}

// one float4 (4 consecutive k of one row) -> 4 bf16 per plane (8 bytes each)
__device__ __forceinline__ void split_store(unsigned short* base /* plane 0 of the operand */, int row, int kq, float4 v) {
  u32 a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
  split3(v.x, a0, a1, a2); split3(v.y, b0, b1, b2); split3(v.z, c0, c1, c2); split3(v.w, d0, d1, d2);
  uint2 p0 = make_uint2((a0 >> 16) | b0, (c0 >> 16) | d0);
  uint2 p1 = make_uint2((a1 >> 16) | b1, (c1 >> 16) | d1);
  uint2 p2 = make_uint2((a2 >> 16) | (b2 & 0xFFFF0000u), (c2 >> 16) | (d2 & 0xFFFF0000u));
  unsigned short* d = base + row * RS + 4 * kq;
  *reinterpret_cast<uint2*>(d) = p0;
  *reinterpret_cast<uint2*>(d + PLANE) = p1;
  *reinterpret_cast<uint2*>(d + 2 * PLANE) = p2;
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_bf16x3(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                             int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                             int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r31 = lane & 31, h = lane >> 5;
  const int nb = gridDim.x;
  const int q = nb >> 3, rr = nb & 7, x = blockIdx.x & 7, s = blockIdx.x >> 3;   // XCD-aware remap (as the engine's)
  const int logical = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + s;
  const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lrow = tid >> 3, kq = tid & 7;   // loader: rows lrow and lrow + 32, float4 kq of the K-step
  const float* ap = A + (size_t)(m0 + lrow) * lda + 4 * kq;
  const float* bp = B + (size_t)(n0 + lrow) * ldb + 4 * kq;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nt = K / BK;
  float4 ra[2], rb[2];
  auto ld = [&](int i) {
    const int k0 = (i < nt ? i : nt - 1) * BK;
    ra[0] = *reinterpret_cast<const float4*>(ap + k0);
    ra[1] = *reinterpret_cast<const float4*>(ap + (size_t)32 * lda + k0);
    rb[0] = *reinterpret_cast<const float4*>(bp + k0);
    rb[1] = *reinterpret_cast<const float4*>(bp + (size_t)32 * ldb + k0);
  };
  auto st = [&](int stage) {
    unsigned short* as = smem + stage * STAGE;
    unsigned short* bs = as + 3 * PLANE;
    split_store(as, lrow, kq, ra[0]); split_store(as, lrow + 32, kq, ra[1]);
    split_store(bs, lrow, kq, rb[0]); split_store(bs, lrow + 32, kq, rb[1]);
  };
  ld(0);
  st(0);
  ld(1);
  __syncthreads();
  for (int i = 0; i < nt; ++i) {
    const unsigned short* as = smem + (i & 1) * STAGE + (wm * 32 + r31) * RS + 8 * h;
    const unsigned short* bs = smem + (i & 1) * STAGE + 3 * PLANE + (wn * 32 + r31) * RS + 8 * h;
#pragma unroll
    for (int c = 0; c < BK / 16; ++c) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[p] = *reinterpret_cast<const bf16x8*>(as + p * PLANE + 16 * c);
        b[p] = *reinterpret_cast<const bf16x8*>(bs + p * PLANE + 16 * c);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    if (i + 1 < nt) st((i + 1) & 1);   // the other stage: its readers finished before the last barrier
    ld(i + 2);
    __syncthreads();
  }
  const int col = n0 + wn * 32 + r31, rbase = m0 + wm * 32 + 4 * h;
  if (col < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      if (row < M) C[(size_t)row * ldc + col] = acc[r];
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 24576, N = argc > 2 ? atoi(argv[2]) : 352, K = argc > 3 ? atoi(argv[3]) : 352;
  const int Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64;
  std::vector<float> hA((size_t)Mp * K, 0.f), hB((size_t)Np * K, 0.f);
  srand(1);
  auto rnd = []() { float u = 0.f; for (int i = 0; i < 12; ++i) u += (float)rand() / RAND_MAX; return u - 6.f; };   // ~N(0,1)
  for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) hA[(size_t)i * K + k] = rnd();
  for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) hB[(size_t)j * K + k] = rnd() * 0.05f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)Mp * Np * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  const int tiles_m = Mp / 64, tiles_n = Np / 64;
  dim3 grid(tiles_m * tiles_n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  for (int i = 0; i < reps + 5; ++i) {
    if (i == 5) hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gemm_nt_bf16x3, grid, dim3(NTHREADS), 0, 0, dA, K, dB, K, dC, Np, M, N, K, tiles_n);
  }
  hipEventRecord(e1, 0);
  hipError_t rc = hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("bf16x3 NT GEMM %dx%dx%d: %.1f us  %.1f TF (fp32-equivalent flops)  [%s]\n", M, N, K, us, 2.0 * M * N * K / us / 1e6,
         hipGetErrorString(rc));
  // accuracy on the first 128 rows against fp64, beside what a plain fp32 dot product (sequential fmaf) gives
  const int R = M < 128 ? M : 128;
  std::vector<float> hC((size_t)R * Np);
  hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0, num32 = 0, maxabs = 0, maxref = 0;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < N; ++j) {
      double ref = 0;
      float f32 = 0.f;
      for (int k = 0; k < K; ++k) {
        ref += (double)hA[(size_t)i * K + k] * (double)hB[(size_t)j * K + k];
        f32 = fmaf(hA[(size_t)i * K + k], hB[(size_t)j * K + k], f32);
      }
      const double d = hC[(size_t)i * Np + j] - ref, d32 = (double)f32 - ref;
      num += d * d; den += ref * ref; num32 += d32 * d32;
      if (fabs(d) > maxabs) maxabs = fabs(d);
      if (fabs(ref) > maxref) maxref = fabs(ref);
    }
  printf("error vs fp64 (first %d rows): rel-l2 %.3e  max|d|/max|ref| %.3e   (sequential fp32 fmaf dot: rel-l2 %.3e)\n", R,
         sqrt(num / den), maxabs / maxref, sqrt(num32 / den));
  return 0;
}
