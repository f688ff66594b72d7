// LDS-DMA staging for the NT GEMM of the eps-net, measured against the engine's register-staged kernel
// (VERDICT r1 "next" item 5).  Stand-alone experiment: nothing here is linked into libsdrm_hip.so.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ldsdma_probe tools/ldsdma_probe.hip && tools/ldsdma_probe
//
// Both kernels: C[M,N] = xf(A)[M,K] * B[N,K]^T, fp32 v_mfma_f32_32x32x2_f32, block tile 64x64x16, four waves (2x2), one
// 32x32 accumulator per wave, XCD-aware tile order, fragments double-buffered in registers, one barrier per K-step.
//   R  the engine's kernel (csrc/gemm.h, gemm_kernel<Cfg0, NT, EPI_PLAIN>): operands global -> VGPR -> ds_write_b128 into a
//      k-minor LDS image with padded rows (stride BK+4), two LDS stages, loads four K-steps ahead in two register sets,
//      every pipeline piece placed in an MFMA shadow.
//   G<S> this file: operands global -> LDS by global_load_lds_dwordx4 (no VGPR round trip, no ds_write), S LDS stages of
//      8 KB, loads S-1 K-steps ahead, counted vmcnt + raw s_barrier.  An LDS-DMA instruction writes 1 KiB of lane-linear
//      LDS (wave-uniform base + lane x 16 B), so rows cannot be padded: the image is [row][16 floats] and the bank
//      conflicts of the b128 fragment reads are removed by an XOR of the 16-byte slot with (row >> 2) & 3, applied to
//      the per-lane SOURCE address and to the read (cdna_hip_programming.md rule 21).  PReLU-on-load moves to
//      fragment-read time (there is no register stage to apply it in).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sdrm_amd/csrc/gemm.h"

using namespace sdrm;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N < 0, "add the count");
}

namespace sdrm {
// S = LDS stages; XFA = PReLU on the A fragments
template <int S, int XFA>
__global__ __launch_bounds__(256, 4) void gemm_glds(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                    float* __restrict__ C, int ldc, int K, int tiles_n, int nblocks, int limA,
                                                    int limB, const float* slope) {
  constexpr int BM = 64, BN = 64, BK = 16;
  constexpr int OP = BM * BK;                 // floats per operand per stage (4 KB)
  __shared__ __attribute__((aligned(1024))) float smem[S * 2 * OP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int logical = xcd_remap((int)blockIdx.x, nblocks);
  const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nt = K / BK;
  const float sl = XFA ? *slope : 0.f;

  // DMA source of this lane: wave w stages rows 16w .. 16w+15 of both operands; lane j -> row 16w + j/4, physical 16-byte
  // slot j%4, which holds logical k-quad (j%4) ^ ((row >> 2) & 3)
  const int drow = 16 * wave + (lane >> 2);
  const int dq = (lane & 3) ^ ((drow >> 2) & 3);
  const float* gA = A + (size_t)(m0 + drow) * lda + 4 * dq;
  const float* gB = B + (size_t)(n0 + drow) * ldb + 4 * dq;
  auto dma = [&](int step, int stage) {
    const int k0 = min(step, nt - 1) * BK;   // past the end: re-read the last step (never consumed; keeps vmcnt uniform)
    float* sa = smem + stage * 2 * OP + 16 * wave * BK;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gA + k0), (lds_ptr_t)sa, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gB + k0), (lds_ptr_t)(sa + OP), 16, 0, 0);
  };
  // fragment addresses: row r of the wave's 32-row strip, logical quads 2*lhi and 2*lhi+1 (lane-half h takes k = 8h..8h+7)
  const int ra = wm * 32 + l31, rb = wn * 32 + l31;
  const int sa = (ra >> 2) & 3, sb = (rb >> 2) & 3;
  const int offA0 = ra * BK + 4 * ((2 * lhi) ^ sa), offA1 = ra * BK + 4 * ((2 * lhi + 1) ^ sa);
  const int offB0 = OP + rb * BK + 4 * ((2 * lhi) ^ sb), offB1 = OP + rb * BK + 4 * ((2 * lhi + 1) ^ sb);
  // The fragment reads are inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read it can see
  // while an LDS-DMA into the same array is outstanding (first build of this probe: the pipeline drained every K-step),
  // and it does not count asm reads at all - their completion is the explicit lgkmcnt(0) ahead of each barrier.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t aA0 = lds0 + 4u * offA0, aA1 = lds0 + 4u * offA1, aB0 = lds0 + 4u * offB0, aB1 = lds0 + 4u * offB1;
  v4f fa[2][2], fb[2][2];
  auto rd = [&](int stage, v4f (&a)[2], v4f (&b)[2]) {
    const uint32_t so = (uint32_t)stage * (2u * OP * 4u);
    asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aA0 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(a[1]) : "v"(aA1 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(aB0 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(b[1]) : "v"(aB1 + so));
  };
  auto landed = [&](v4f (&a)[2], v4f (&b)[2]) {   // the asm reads have completed: their registers may be touched
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    __builtin_amdgcn_sched_barrier(0);
    if (XFA) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        a[q].x = prelu_f(a[q].x, sl); a[q].y = prelu_f(a[q].y, sl); a[q].z = prelu_f(a[q].z, sl); a[q].w = prelu_f(a[q].w, sl);
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool active = (m0 + wm * 32 < limA) && (n0 + wn * 32 < limB);

  // prologue: S-1 steps in flight, the first one landed and read
#pragma unroll
  for (int s = 0; s < S - 1; ++s) dma(s, s);
  wait_vmcnt<2 * (S - 2)>();
  __builtin_amdgcn_s_barrier();
  rd(0, fa[0], fb[0]);
  landed(fa[0], fb[0]);

  auto mfmas = [&](const v4f (&a)[2], const v4f (&b)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
    }
  };
  // step i: (1) DMA of step i+S-1 into the stage step i-1's fragments were read from two barriers ago, (2) wait until this
  // wave's DMAs of step i+1 have landed, (3) barrier: everybody's have, and everybody is done reading stage (i-1)%S,
  // (4) read step i+1's fragments, (5) MFMAs of step i out of registers (they cover (4) and the DMA issue of the next step)
  auto step = [&](int i, int cur, int stage_next, int stage_dma) {
    dma(i + S - 1, stage_dma);
    wait_vmcnt<2 * (S - 2)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
      rd(stage_next, fa[cur ^ 1], fb[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa[cur], fb[cur]);
      __builtin_amdgcn_sched_barrier(0);   // the wait below goes BEHIND the MFMAs (hipcc hoisted it to after the first one)
      landed(fa[cur ^ 1], fb[cur ^ 1]);   // before this wave reaches the next barrier: the stage may be recycled after it
    }
  };
  int st_next = 1 % S, st_dma = (S - 1) % S;
  int i = 0;
  for (; i + 1 < nt; i += 2) {
    step(i, 0, st_next, st_dma);
    st_next = (st_next + 1) % S; st_dma = (st_dma + 1) % S;
    step(i + 1, 1, st_next, st_dma);
    st_next = (st_next + 1) % S; st_dma = (st_dma + 1) % S;
  }
  if (i < nt) step(i, 0, st_next, st_dma);
  wait_vmcnt<0>();
  if (!active) return;
  const int rbase = m0 + wm * 32 + 4 * lhi, col = n0 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[r];
}

}  // namespace sdrm

typedef TileCfg<64, 64, 2, 2, 4, 16> Cfg0;

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Case { int M, N, K; const char* label; };

template <int XFA>
void run_case(const Case& c, int reps, int rounds) {
  const int Mp = round_up(c.M, 128), Np = round_up(c.N, 128);
  std::vector<float> hA((size_t)Mp * c.K, 0.f), hB((size_t)Np * c.K, 0.f);
  srand(7);
  for (int i = 0; i < c.M; ++i) for (int k = 0; k < c.K; ++k) hA[(size_t)i * c.K + k] = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (int i = 0; i < c.N; ++i) for (int k = 0; k < c.K; ++k) hB[(size_t)i * c.K + k] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  float *dA, *dB, *dC0, *dC1, *dSlope;
  const size_t slack = 8192;
  CHECK(hipMalloc(&dA, (hA.size() + slack) * 4)); CHECK(hipMalloc(&dB, (hB.size() + slack) * 4));
  CHECK(hipMalloc(&dC0, ((size_t)Mp * Np + slack) * 4)); CHECK(hipMalloc(&dC1, ((size_t)Mp * Np + slack) * 4));
  CHECK(hipMalloc(&dSlope, 4));
  CHECK(hipMemset(dA, 0, (hA.size() + slack) * 4)); CHECK(hipMemset(dB, 0, (hB.size() + slack) * 4));
  CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  const float slope = 0.25f;
  CHECK(hipMemcpy(dSlope, &slope, 4, hipMemcpyHostToDevice));
  const int tiles_m = (c.M + 63) / 64, tiles_n = (c.N + 63) / 64, nblocks = tiles_m * tiles_n;
  GemmArgs a{};
  a.A = dA; a.lda = c.K; a.limA = c.M; a.B = dB; a.ldb = c.K; a.limB = c.N; a.C = dC0; a.ldc = Np; a.K = c.K; a.kchunk = c.K;
  a.slopeA = dSlope;
  if (!gemm_set_grid(a, tiles_m, tiles_n, 1)) { fprintf(stderr, "grid too large\n"); exit(1); }   // tiles, K-slices and the magics gemm_body divides by
  auto launch = [&](int which) {
    switch (which) {
      case 0: hipLaunchKernelGGL((gemm_kernel<Cfg0, LD_KCONTIG, LD_KCONTIG, XFA, XF_NONE, EPI_PLAIN>), dim3(nblocks), dim3(256), 0, 0, a); break;
      case 1: hipLaunchKernelGGL((gemm_glds<3, XFA>), dim3(nblocks), dim3(256), 0, 0, dA, c.K, dB, c.K, dC1, Np, c.K, tiles_n, nblocks, c.M, c.N, dSlope); break;
      case 2: hipLaunchKernelGGL((gemm_glds<4, XFA>), dim3(nblocks), dim3(256), 0, 0, dA, c.K, dB, c.K, dC1, Np, c.K, tiles_n, nblocks, c.M, c.N, dSlope); break;
      default: hipLaunchKernelGGL((gemm_glds<6, XFA>), dim3(nblocks), dim3(256), 0, 0, dA, c.K, dB, c.K, dC1, Np, c.K, tiles_n, nblocks, c.M, c.N, dSlope); break;
    }
  };
  // correctness of every DMA variant against the engine's kernel (same fp32 MFMA chain per accumulator -> identical bits)
  std::vector<float> r0((size_t)Mp * Np), r1((size_t)Mp * Np);
  launch(0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(r0.data(), dC0, r0.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int v = 1; v <= 3; ++v) {
    CHECK(hipMemset(dC1, 0xff, (size_t)Mp * Np * 4));
    launch(v);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(r1.data(), dC1, r1.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < c.M; ++i)
      for (int j = 0; j < c.N; ++j) {
        const double d = std::fabs((double)r0[(size_t)i * Np + j] - (double)r1[(size_t)i * Np + j]);
        if (!(d <= worst)) worst = d;
      }
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<std::vector<float>> us(4);
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < 4; ++v) {      // variants interleaved in one process (rule 24)
      launch(v);
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < reps; ++k) launch(v);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      us[v].push_back(ms * 1e3f / reps);
    }
  const double fl = 2.0 * c.M * c.N * c.K;
  printf("%-26s xf=%d %6dx%4dx%4d  max|G-R| %.1e |", c.label, XFA, c.M, c.N, c.K, worst);
  const char* names[4] = {"R", "G3", "G4", "G6"};
  for (int v = 0; v < 4; ++v) {
    std::sort(us[v].begin(), us[v].end());
    const float med = us[v][us[v].size() / 2], mn = us[v][0];
    printf("  %s med %6.2f us (%5.1f TF) min %6.2f", names[v], med, fl / med / 1e6, mn);
  }
  printf("\n");
  CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC0)); CHECK(hipFree(dC1)); CHECK(hipFree(dSlope));
}

int main(int argc, char** argv) {
  if (argc > 1) {   // profiling mode: few launches of the train-size shapes only (rocprofv3 --pmc passes)
    const Case c{24576, 352, 352, "train fwd hidden B=8192"};
    run_case<0>(c, 3, 1);
    run_case<1>(c, 3, 1);
    return 0;
  }
  const Case cases[] = {{24576, 352, 352, "train fwd hidden B=8192"}, {24576, 352, 448, "train fwd layer0 B=8192"},
                        {5440, 352, 352, "sample fwd n=5429"}, {3072, 352, 352, "8-GPU train shard"}, {704, 352, 352, "8-GPU sample shard"}};
  printf("# R = engine kernel (register staging, padded k-minor LDS, 2 stages); G<S> = LDS-DMA staging, S stages of 8 KB; flops on the launch dims\n");
  for (const Case& c : cases) {
    run_case<0>(c, 20, 7);
    run_case<1>(c, 20, 7);
  }
  return 0;
}
