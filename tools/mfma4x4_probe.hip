// Calibration probe for DESIGN.md section 8 (round 4): could a ROW-OWNED sampling step at n = 5429 - 24 rows
// (six 4-row units of v_mfma_f32_4x4x1_16B_f32) per work-group, 227 work-groups, every layer and the reverse update in one
// launch - feed its matrix pipes?  The instruction has a quarter of the 16x16x4 instruction's flops per operand register, and
// a 24-row work-group re-reads ALL weights (352 x 352 floats per layer) from L2.  This probe runs exactly that operand
// pattern with no arithmetic around it and reports the time of one "step" (three layers):
//   wave (wr, wc) of a 2 x 2 work-group owns row units 3 wr .. 3 wr + 2 and 64-column blocks 3 wc .. 3 wc + 2 (nine accumulator
//   quads); per group of four k: 3 ds_read_b128 (activations of its three units, broadcast reads: a 4x4x1 A operand is the same 4
//   values in all 16 blocks), 3 x 1 KiB wave-loads of k-quad-packed weights, 36 MFMAs (288 cycles), prefetched one group ahead.
// MODE bit0: issue the LDS reads, bit1: issue the weight loads (without a bit the operands of the first group are re-used).
// The MFMA-only figure is the floor (3 x 88 x 36 x 8 cycles = 76 k cycles per step); the kill criterion of the real kernel was
// <= 38 us per step INCLUDING staging, epilogues and the reverse update.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma4x4_probe tools/mfma4x4_probe.hip && tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NP = 352, KQ = NP / 4, LDA = NP + 4, ROWS = 24, NCB = 6;   // column blocks of 64 (the sixth half real)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_probe(const float* __restrict__ Wq /* [3][KQ][NCB][64][4] */, float* __restrict__ out, int steps) {
  __shared__ __attribute__((aligned(16))) float Act[ROWS * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  for (int i = tid; i < ROWS * LDA; i += 256) Act[i] = 1e-3f * (float)(i % 97);
  __syncthreads();
  f32x4 acc[3][3];
  for (int u = 0; u < 3; ++u)
    for (int c = 0; c < 3; ++c) acc[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A operand of unit u: lane -> row 4 (3 wr + u) + lane % 4, four consecutive k
  const float* abase = Act + (4 * 3 * wr + (lane & 3)) * LDA;
  const f32x4* wbase = reinterpret_cast<const f32x4*>(Wq) + (size_t)(3 * wc) * 64 + lane;
  f32x4 a0[3], b0[3], a1[3], b1[3];
  auto fetch = [&](f32x4 (&a)[3], f32x4 (&b)[3], int layer, int kq) __attribute__((always_inline)) {
    if (MODE & 1) {
#pragma unroll
      for (int u = 0; u < 3; ++u) a[u] = *reinterpret_cast<const f32x4*>(abase + 4 * u * LDA + 4 * kq);
    }
    if (MODE & 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) b[c] = wbase[((size_t)layer * KQ + kq) * NCB * 64 + c * 64];
    }
  };
  auto mm = [&](const f32x4 (&a)[3], const f32x4 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[u][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][e], b[c][e], acc[u][c], 0, 0, 0);
  };
  for (int u = 0; u < 3; ++u) { a0[u] = *reinterpret_cast<const f32x4*>(abase + 4 * u * LDA); a1[u] = a0[u]; }
  for (int c = 0; c < 3; ++c) { b0[c] = wbase[c * 64]; b1[c] = b0[c]; }
  for (int s = 0; s < steps; ++s)
    for (int layer = 0; layer < 3; ++layer) {
      for (int kq = 0; kq < KQ; kq += 2) {
        fetch(a1, b1, layer, kq + 1);
        mm(a0, b0);
        fetch(a0, b0, layer, (kq + 2 < KQ) ? kq + 2 : 0);
        mm(a1, b1);
      }
      // (a real kernel has an epilogue and a barrier here; the probe keeps the accumulators running)
    }
  float s = 0.f;
  for (int u = 0; u < 3; ++u)
    for (int c = 0; c < 3; ++c) s += acc[u][c][0] + acc[u][c][1] + acc[u][c][2] + acc[u][c][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static float run(const float* W, float* out, int grid, int steps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, W, out, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, W, out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / steps;
}

int main() {
  const size_t nw = (size_t)3 * KQ * NCB * 64 * 4;
  std::vector<float> h(nw);
  for (size_t i = 0; i < nw; ++i) h[i] = 1e-3f * (float)((i * 7) % 31);
  float *W, *out;
  hipMalloc(&W, nw * 4); hipMalloc(&out, 256 * 256 * 4);
  hipMemcpy(W, h.data(), nw * 4, hipMemcpyHostToDevice);
  const int steps = 78;
  printf("one step = three layers of a 24-row work-group (6 units x 6 column blocks x 352 k), us per step; the matrix work alone is 76 032 cycles\n");
  for (int grid : {1, 64, 227, 256}) {
    printf("grid %3d:  MFMA only %6.1f   + LDS A reads %6.1f   + weight loads %6.1f   both %6.1f\n", grid, run<0>(W, out, grid, steps),
           run<1>(W, out, grid, steps), run<2>(W, out, grid, steps), run<3>(W, out, grid, steps));
  }
  hipFree(W); hipFree(out);
  return 0;
}
