// Probe of the strip-owned split-K weight gradients (sdrm_amd/csrc/wgrad2.h) against the engine's batched 64x64-tile launch
// (gemm_batch_kernel, csrc/gemm.h) on the same operands, ML-1M shapes: 24576 stacked rows, three problems (layer 0: 352 x 448,
// hidden and out: 352 x 352).  Checks sampled outputs against fp64 sums on the host, then times both (interleaved).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/wgrad2_probe tools/wgrad2_probe.hip && tools/wgrad2_probe [rows]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../sdrm_amd/csrc/wgrad2.h"
#include "wgrad_strips16_experiment.h"

using namespace sdrm;

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t _e = (x);                                                                \
    if (_e != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(_e)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef TileCfg<64, 64, 2, 2, 4, 16> Cfg0;
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <typename T>
T* dalloc(size_t n) {
  T* p;
  CHECK(hipMalloc(&p, (n + 8192) * sizeof(T)));
  CHECK(hipMemset(p, 0, (n + 8192) * sizeof(T)));
  return p;
}

int main(int argc, char** argv) {
  const int MP = argc > 1 ? atoi(argv[1]) : 24576;
  constexpr int NT = 11;
  const int WP = 32 * NT, K0 = WP + 96;
  std::mt19937 rng(3);
  std::normal_distribution<float> nrm(0.f, 1.f);
  // operands: A_p [MP][WP] gradients, B_p [MP][ldb] forward operands
  const int ldb[3] = {K0, WP, WP};
  std::vector<float> hA[3], hB[3];
  float *dA[3], *dB[3], *dslab[3], *dslab_old[3];
  const int S_old = 24, kc_old = 1024;
  const int units_expected = (K0 / 32 + 2 * (WP / 32) + 3) / 4;
  const int S = 256 / units_expected;
  const int kchunk = round_up((MP + S - 1) / S, WG2_BK);
  const int slices = (MP + kchunk - 1) / kchunk;
  for (int p = 0; p < 3; ++p) {
    hA[p].resize((size_t)MP * WP); hB[p].resize((size_t)MP * ldb[p]);
    for (auto& v : hA[p]) v = nrm(rng) * 1e-3f;
    for (auto& v : hB[p]) v = nrm(rng);
    dA[p] = dalloc<float>(hA[p].size()); dB[p] = dalloc<float>(hB[p].size());
    CHECK(hipMemcpy(dA[p], hA[p].data(), hA[p].size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB[p], hB[p].data(), hB[p].size() * 4, hipMemcpyHostToDevice));
    dslab[p] = dalloc<float>((size_t)64 * WP * ldb[p]);
    dslab_old[p] = dalloc<float>((size_t)64 * WP * ldb[p]);
  }
  Wg2Args a{};
  int ktiles[3];
  for (int p = 0; p < 3; ++p) {
    a.p[p].A = dA[p]; a.p[p].lda = WP; a.p[p].B = dB[p]; a.p[p].ldb = ldb[p];
    a.p[p].slab = dslab[p]; a.p[p].ldc = ldb[p]; a.p[p].slab_stride = (size_t)WP * ldb[p];
    ktiles[p] = ldb[p] / 32;
  }
  const int units = wg2_plan(ktiles, 3, a);
  a.slices = slices; a.rows = MP; a.kchunk = kchunk;
  const int grid = units * slices;
  printf("# %d rows: %d strips in %d units x %d slices of %d rows = %d work-groups (grid %d), LDS %zu B\n", MP, ktiles[0] + ktiles[1] + ktiles[2],
         units, slices, kchunk, units * slices, grid, Wg2Cfg<NT>::LDS_BYTES);
  auto launch_new = [&]() { hipLaunchKernelGGL((k_wgrad_strips<NT>), dim3(grid), dim3(NTHREADS), 0, 0, a); };
  // the 16-wide strips: twice the units, half the slices (slabs of their own)
  Wg2Args a16{};
  float* dslab16[3];
  int ktiles16[3];
  for (int p = 0; p < 3; ++p) {
    dslab16[p] = dalloc<float>((size_t)64 * WP * ldb[p]);
    a16.p[p] = a.p[p]; a16.p[p].slab = dslab16[p];
    ktiles16[p] = ldb[p] / 16;
  }
  const int units16 = wg2_plan(ktiles16, 3, a16);
  const int S16 = 256 / units16;
  const int kchunk16 = round_up((MP + S16 - 1) / S16, WG2_BK);
  const int slices16 = (MP + kchunk16 - 1) / kchunk16;
  a16.slices = slices16; a16.rows = MP; a16.kchunk = kchunk16;
  const int grid16 = units16 * slices16;
  printf("# 16-wide strips: %d units x %d slices of %d rows = %d work-groups, LDS %zu B\n", units16, slices16, kchunk16, grid16, Wg2Cfg16<2 * NT>::LDS_BYTES);
  auto launch_16 = [&]() { hipLaunchKernelGGL((k_wgrad_strips16<2 * NT>), dim3(grid16), dim3(NTHREADS), 0, 0, a16); };

  // the engine's batched launch on the same operands (24 slices of 1024 rows)
  GemmBatch gb{};
  gb.n = 3;
  int gsz = 0;
  for (int p = 0; p < 3; ++p) {
    GemmArgs& g = gb.p[p];
    g.A = dA[p]; g.lda = WP; g.limA = WP; g.B = dB[p]; g.ldb = ldb[p]; g.limB = ldb[p];
    g.C = dslab_old[p]; g.ldc = ldb[p]; g.K = MP; g.kchunk = kc_old; g.slab_stride = (size_t)WP * ldb[p];
    const int S_here = (MP + kc_old - 1) / kc_old;
    if (!gemm_set_grid(g, (WP + 63) / 64, (ldb[p] + 63) / 64, S_here)) return 1;
    gb.start[p] = gsz;
    gsz += round_up(g.nblocks * S_here, 8);
  }
  gb.start[3] = gsz;
  (void)S_old;
  auto launch_old = [&]() {
    hipLaunchKernelGGL((gemm_batch_kernel<Cfg0, LD_MCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_SLAB>), dim3(gsz), dim3(NTHREADS), 0, 0, gb);
  };

  launch_new();
  launch_16();
  launch_old();
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  // sampled outputs: sum over slices of the slabs against fp64 on the host
  double worst = 0, worst_old = 0, worst16 = 0, ref_max = 0;
  for (int p = 0; p < 3; ++p) {
    std::vector<float> hs((size_t)slices * WP * ldb[p]), ho((size_t)((MP + kc_old - 1) / kc_old) * WP * ldb[p]), h16((size_t)slices16 * WP * ldb[p]);
    CHECK(hipMemcpy(hs.data(), dslab[p], hs.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h16.data(), dslab16[p], h16.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ho.data(), dslab_old[p], ho.size() * 4, hipMemcpyDeviceToHost));
    for (int smp = 0; smp < 400; ++smp) {
      const int n = (int)(rng() % WP), k = (int)(rng() % ldb[p]);
      double ref = 0;
      for (int m = 0; m < MP; ++m) ref += (double)hA[p][(size_t)m * WP + n] * (double)hB[p][(size_t)m * ldb[p] + k];
      double got = 0, old = 0, g16 = 0;
      for (int s = 0; s < slices; ++s) got += hs[(size_t)s * WP * ldb[p] + (size_t)n * ldb[p] + k];
      for (int s = 0; s < slices16; ++s) g16 += h16[(size_t)s * WP * ldb[p] + (size_t)n * ldb[p] + k];
      worst16 = std::max(worst16, std::fabs(g16 - ref));
      for (int s = 0; s < (MP + kc_old - 1) / kc_old; ++s) old += ho[(size_t)s * WP * ldb[p] + (size_t)n * ldb[p] + k];
      worst = std::max(worst, std::fabs(got - ref)); worst_old = std::max(worst_old, std::fabs(old - ref));
      ref_max = std::max(ref_max, std::fabs(ref));
    }
  }
  const bool ok = worst <= 1e-5 * ref_max && worst16 <= 1e-5 * ref_max;
  printf("max|err| over 1200 sampled outputs: strips %.3e, 16-wide strips %.3e, 64x64 batch %.3e (max|ref| %.3e): %s\n", worst, worst16, worst_old, ref_max,
         ok ? "OK" : "FAILED");

  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> us[3];
  for (int r = 0; r < 7; ++r)
    for (int v = 0; v < 3; ++v) {
      auto go = [&]() { if (v == 0) launch_new(); else if (v == 1) launch_old(); else launch_16(); };
      go();
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < 20; ++k) go();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      us[v].push_back(ms * 1e3f / 20);
    }
  const double fl = 2.0 * MP * (340.0 * 418 + 340.0 * 340 + 340.0 * 340);
  const char* names[3] = {"strip-owned (one work-group per CU)", "64x64 tiles, batched (the engine's)", "strip-owned, 16-wide strips"};
  for (int v = 0; v < 3; ++v) {
    std::sort(us[v].begin(), us[v].end());
    printf("%-40s med %7.2f us  min %7.2f us  (%5.1f TF on the unpadded dims, frac %.3f)\n", names[v], us[v][3], us[v][0], fl / us[v][3] / 1e6,
           fl / us[v][3] / 1e6 / 157.3);
  }
  return ok ? 0 : 1;
}
