// Probe of the row-owned input gradient (sdrm_amd/csrc/dgrad_rows.h) against the engine's 64x64-tile NT launch with the
// EPI_DPRELU epilogue (csrc/gemm.h) on the same operands, ML-1M shape: 24576 stacked rows, 352 x 352.  Checks sampled outputs
// and the slope partial sum against fp64 on the host, then times both (interleaved).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/dgrad_rows_probe tools/dgrad_rows_probe.hip && tools/dgrad_rows_probe [rows]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../sdrm_amd/csrc/dgrad_rows.h"

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

template <typename T>
T* dalloc(size_t n) {
  T* p;
  CHECK(hipMalloc(&p, (n + 8192) * sizeof(T)));
  CHECK(hipMemset(p, 0, (n + 8192) * sizeof(T)));
  return p;
}

#ifndef PROBE_CT
#define PROBE_CT 11
#endif

int main(int argc, char** argv) {
  const int MP = argc > 1 ? atoi(argv[1]) : 24576;
  constexpr int CT = PROBE_CT;
  const int NP = 32 * CT, W = NP - 12;   // logical width: the pad rows / columns of the weight are zero
  if (MP % RC_ROWS) { fprintf(stderr, "rows must be a multiple of %d\n", RC_ROWS); return 1; }
  std::mt19937 rng(5);
  std::normal_distribution<float> nrm(0.f, 1.f);
  std::vector<float> hG((size_t)MP * NP), hP((size_t)MP * NP), hW((size_t)NP * NP, 0.f), hWT((size_t)NP * NP, 0.f), hF((size_t)NP * NP, 0.f);
  for (auto& v : hG) v = nrm(rng) * 1e-3f;
  for (auto& v : hP) v = nrm(rng);
  for (int k = 0; k < W; ++k)        // W[k = out][n = in]
    for (int n = 0; n < W; ++n) {
      const float w = nrm(rng) * 0.05f;
      hW[(size_t)k * NP + n] = w;
      hWT[(size_t)n * NP + k] = w;                       // the NT kernel's operand: [n][k], k contiguous
      hF[wfrag_index(n, k, NP / 16, rc_light_klast(W, NP))] = w;   // W = NP - 12: the last K-step holds four real k - the compact form
    }
  const float slope = 0.25f;
  float *dG = dalloc<float>(hG.size()), *dP = dalloc<float>(hP.size()), *dWT = dalloc<float>(hWT.size()), *dF = dalloc<float>(hF.size());
  float *dO = dalloc<float>(hG.size()), *dO2 = dalloc<float>(hG.size()), *dS = dalloc<float>(1), *dpart = dalloc<float>(1 << 16), *dpart2 = dalloc<float>(1 << 16);
  CHECK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dWT, hWT.data(), hWT.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dF, hF.data(), hF.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dS, &slope, 4, hipMemcpyHostToDevice));

  DgradRowsArgs a{};
  a.G = dG; a.ldg = NP; a.WfT = dF; a.pre = dP; a.ldp = NP; a.slope = dS; a.out = dO; a.ldo = NP; a.slope_part = dpart;
  const int grid = MP / RC_ROWS;
  auto launch_new = [&]() { hipLaunchKernelGGL((k_dgrad_rows<CT, true>), dim3(grid), dim3(NTHREADS), 0, 0, a); };

  GemmArgs g{};
  g.A = dG; g.lda = NP; g.limA = MP; g.B = dWT; g.ldb = NP; g.limB = NP; g.C = dO2; g.ldc = NP; g.K = NP; g.kchunk = NP;
  g.aux = dP; g.ldaux = NP; g.slopeE = dS; g.slope_partial = dpart2;
  if (!gemm_set_grid(g, (MP + 63) / 64, (NP + 63) / 64, 1)) return 1;
  const int gsz = g.nblocks;
  auto launch_old = [&]() {
    hipLaunchKernelGGL((gemm_kernel<Cfg0, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_DPRELU>), dim3(gsz), dim3(NTHREADS), 0, 0, g);
  };
  launch_new();
  launch_old();
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());

  std::vector<float> hO(hG.size()), hO2(hG.size()), hpart(grid), hpart2(gsz);
  CHECK(hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hO2.data(), dO2, hO2.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hpart.data(), dpart, grid * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hpart2.data(), dpart2, gsz * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst2 = 0, ref_max = 0;
  for (int smp = 0; smp < 4000; ++smp) {
    const int r = (int)(rng() % MP), n = (int)(rng() % NP);
    double acc = 0;
    for (int k = 0; k < NP; ++k) acc += (double)hG[(size_t)r * NP + k] * (double)hW[(size_t)k * NP + n];
    const double ref = acc * (hP[(size_t)r * NP + n] > 0.f ? 1.0 : (double)slope);
    worst = std::max(worst, std::fabs(hO[(size_t)r * NP + n] - ref));
    worst2 = std::max(worst2, std::fabs(hO2[(size_t)r * NP + n] - ref));
    ref_max = std::max(ref_max, std::fabs(ref));
  }
  // the slope partial sum over the first 192 rows (two work-groups), against fp64
  double sref = 0, sgot = hpart[0] + (grid > 1 ? hpart[1] : 0.f);
  const int srows = std::min(MP, 2 * RC_ROWS);
  for (int r = 0; r < srows; ++r)
    for (int n = 0; n < NP; ++n) {
      double acc = 0;
      for (int k = 0; k < NP; ++k) acc += (double)hG[(size_t)r * NP + k] * (double)hW[(size_t)k * NP + n];
      sref += acc * std::min((double)hP[(size_t)r * NP + n], 0.0);
    }
  double tot = 0, tot2 = 0;
  for (float v : hpart) tot += v;
  for (float v : hpart2) tot2 += v;
  const bool ok = worst <= 2e-5 * ref_max && std::fabs(sgot - sref) <= 1e-4 * std::fabs(sref) + 1e-7 && std::fabs(tot - tot2) <= 1e-4 * std::fabs(tot2) + 1e-6;
  printf("# %d rows x %d (CT = %d): %d work-groups of %d rows\n", MP, NP, CT, grid, RC_ROWS);
  printf("max|err| over 4000 sampled outputs: row-owned %.3e, 64x64 tiles %.3e (max|ref| %.3e); slope partial of 192 rows %.6e (fp64 %.6e); "
         "totals %.6e / %.6e: %s\n", worst, worst2, ref_max, sgot, sref, tot, tot2, ok ? "OK" : "FAILED");

#ifdef DR_STAMPS
  {
    unsigned long long* dst = dalloc<unsigned long long>(8 * (size_t)grid);
    a.stamps = dst;
    for (int k = 0; k < 20; ++k) launch_new();
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(8 * (size_t)grid);
    CHECK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
    a.stamps = nullptr;
    const char* seg[4] = {"prologue", "main loop", "tail (pre-activation loads)", "epilogue"};
    printf("# stamps of the last of 20 back-to-back launches: cycles per segment, median / p90 / max over %d work-groups\n", grid);
    for (int sgi = 0; sgi < 4; ++sgi) {
      std::vector<unsigned long long> d(grid);
      for (int gi = 0; gi < grid; ++gi) d[gi] = h[8 * gi + sgi + 1] - h[8 * gi + sgi];
      std::sort(d.begin(), d.end());
      printf("  %-28s %8llu %8llu %8llu\n", seg[sgi], d[grid / 2], d[grid * 9 / 10], d[grid - 1]);
    }
    std::vector<unsigned long long> d(grid), rt(grid);
    for (int gi = 0; gi < grid; ++gi) { d[gi] = h[8 * gi + 4] - h[8 * gi]; rt[gi] = h[8 * gi + 5]; }
    std::sort(d.begin(), d.end()); std::sort(rt.begin(), rt.end());
    printf("  %-28s %8llu %8llu %8llu   last exit - first exit %.2f us (100 MHz clock)\n", "work-group", d[grid / 2], d[grid * 9 / 10], d[grid - 1],
           (double)(rt[grid - 1] - rt[0]) / 100.0);
  }
#endif
  // the chain: loss seeds + two layers (output layer, hidden layer) in one launch, on synthetic Y / x0 / partial sums
  DgradChainArgs ch{};
  {
    const int B = RC_USERS * grid;
    std::vector<float> hY((size_t)MP * NP), hX((size_t)B * W);
    std::uniform_real_distribution<float> uni(-0.9f, 0.9f);
    for (auto& v : hY) v = uni(rng);
    for (auto& v : hX) v = uni(rng);
    std::vector<double> hpartd(4 * (size_t)grid);
    for (int gi = 0; gi < grid; ++gi) { hpartd[4 * gi] = 30.0; hpartd[4 * gi + 1] = 20.0; hpartd[4 * gi + 2] = 1.0; hpartd[4 * gi + 3] = 900.0; }
    float *dY2 = dalloc<float>(hY.size()), *dX = dalloc<float>(hX.size()), *dG1 = dalloc<float>(hG.size()), *dloss = dalloc<float>(1);
    double* dpd = dalloc<double>(hpartd.size());
    CHECK(hipMemcpy(dY2, hY.data(), hY.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dpd, hpartd.data(), hpartd.size() * 8, hipMemcpyHostToDevice));
    ch.seed.sums = nullptr; ch.seed.Y = dY2; ch.seed.x0 = dX; ch.seed.dY = dG; ch.seed.loss = dloss;
    ch.seed.B = B; ch.seed.L = W; ch.seed.LP = NP; ch.seed.MP = MP; ch.seed.grouped = 1;
    ch.seed.part = dpd; ch.seed.nblk = grid; ch.seed.count = (double)B * W;
    ch.nlayers = 2;
    ch.layer[0] = a; ch.layer[0].out = dG1;
    ch.layer[1] = a; ch.layer[1].G = dG1; ch.layer[1].out = dO; ch.layer[1].slope_part = dpart + 4096;
  }
  auto launch_chain = [&]() { hipLaunchKernelGGL((k_dgrad_chain<CT, true>), dim3(grid), dim3(NTHREADS), 0, 0, ch); };
#ifdef DR_STAMPS
  {
    unsigned long long* dss = dalloc<unsigned long long>(grid);
    ch.seed_stamps = dss;
    for (int k = 0; k < 10; ++k) launch_chain();
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), dss, grid * 8, hipMemcpyDeviceToHost));
    ch.seed_stamps = nullptr;
    std::sort(h.begin(), h.end());
    printf("# chain: seed stage %llu / %llu / %llu cycles (median / p90 / max over %d work-groups)\n", h[grid / 2], h[grid * 9 / 10], h[grid - 1], grid);
  }
#endif
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  {
    launch_chain();
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < 20; ++k) launch_chain();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms * 1e3f / 20);
    }
    std::sort(t.begin(), t.end());
    printf("%-40s med %7.2f us  min %7.2f us\n", "chain: seeds + 2 layers, one launch", t[3], t[0]);
    CHECK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));   // the seeds overwrote the probe's G
  }
  std::vector<float> us[2];
  for (int r = 0; r < 7; ++r)
    for (int v = 0; v < 2; ++v) {
      auto go = [&]() { if (v == 0) launch_new(); else launch_old(); };
      go();
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < 20; ++k) go();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      us[v].push_back(ms * 1e3f / 20);
    }
  const double fl = 2.0 * MP * (double)W * W;
  const char* names[2] = {"row-owned (one work-group per CU)", "64x64 tiles (the engine's NT launch)"};
  for (int v = 0; v < 2; ++v) {
    std::sort(us[v].begin(), us[v].end());
    printf("%-40s med %7.2f us  min %7.2f us  (%5.1f TF on the unpadded dims, frac %.3f)\n", names[v], us[v][3], us[v][0], fl / us[v][3] / 1e6,
           fl / us[v][3] / 1e6 / 157.3);
  }
  return ok ? 0 : 1;
}
