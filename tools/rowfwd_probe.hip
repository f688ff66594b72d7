// Probe of the row-owned train forward (sdrm_amd/csrc/rowchain.h): staging + all three layers of the
// ML-1M eps-net for 24576 stacked rows in ONE launch, one work-group per CU, against the per-layer path's three NT GEMM
// launches (the engine's own gemm_kernel, same process, interleaved).  Checks a sample of rows against an fp64 host
// reference (EXPLICIT randoms), then times EXPLICIT and PHILOX staging.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/rowfwd_probe tools/rowfwd_probe.hip
//   tools/rowfwd_probe [B]          (kill criterion: <= 150 us at B = 8192 against 182 + 18.7 us)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../sdrm_amd/csrc/rowchain.h"

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
  const int B = argc > 1 ? atoi(argv[1]) : 8192;
  const int quick = argc > 2 ? atoi(argv[2]) : 0;   // profiling mode: few launches
  constexpr int CT = 11;
  const int L = 340, W = 340, T = 78, H = 1;
  const int NP = 32 * CT, NCT = 2 * CT, TP = round_up(T + 1, 32), K0 = NP + TP;
  const int G = (B + RC_USERS - 1) / RC_USERS, MP = round_up(G * RC_ROWS, 128);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> uni(-1.f, 1.f);
  std::normal_distribution<float> nrm(0.f, 1.f);

  // weights [out][in] (latent part of layer 0 only), padded to NP x NP with zeros, and their fragment-packed copies
  std::vector<float> hW[3], hWf[3], hWc[3];
  const float sc[3] = {1.f / std::sqrt((float)(L + T)), 1.f / std::sqrt((float)W), 1.f / std::sqrt((float)W)};
  for (int l = 0; l < 3; ++l) {
    hW[l].assign((size_t)NP * NP, 0.f);
    for (int n = 0; n < W; ++n)
      for (int k = 0; k < L; ++k) hW[l][(size_t)n * NP + k] = uni(rng) * sc[l];
    hWf[l].assign((size_t)NP * NP, 0.f);
    for (int n = 0; n < NP; ++n)
      for (int k = 0; k < NP; ++k) hWf[l][wfrag_index(n, k, NCT, rc_light_klast(L, NP))] = hW[l][(size_t)n * NP + k];   // compact last K-step
  }
  std::vector<float> hB0tab((size_t)(T + 1) * NP, 0.f), hbh(NP, 0.f), hbo(NP, 0.f);
  for (int t = 0; t <= T; ++t) for (int n = 0; n < W; ++n) hB0tab[(size_t)t * NP + n] = uni(rng) * 0.1f;
  for (int n = 0; n < W; ++n) { hbh[n] = uni(rng) * 0.05f; hbo[n] = uni(rng) * 0.05f; }
  std::vector<float> hx0((size_t)B * L), hnoise((size_t)B * L), hsq(T + 1), hom(T + 1);
  std::vector<int64_t> ht(B);
  std::vector<uint8_t> hkeep((size_t)3 * B * L);
  for (auto& v : hx0) v = nrm(rng);
  for (auto& v : hnoise) v = nrm(rng);
  for (auto& v : ht) v = 1 + (int)(rng() % T);
  for (auto& v : hkeep) v = rng() & 1;
  for (int t = 0; t <= T; ++t) { const float ab = std::exp(-0.01f * t * t / T); hsq[t] = std::sqrt(ab); hom[t] = 1.f - ab; }
  const float slope0 = 0.25f, slopeh = 0.2f;

  float *dWf[3], *dWc[3];
  for (int l = 0; l < 3; ++l) {
    dWf[l] = dalloc<float>(hWf[l].size());
    CHECK(hipMemcpy(dWf[l], hWf[l].data(), hWf[l].size() * 4, hipMemcpyHostToDevice));
    dWc[l] = dalloc<float>((size_t)round_up(NP, 128) * NP);
    CHECK(hipMemcpy(dWc[l], hW[l].data(), hW[l].size() * 4, hipMemcpyHostToDevice));
  }
  float* dB0tab = dalloc<float>(hB0tab.size()); CHECK(hipMemcpy(dB0tab, hB0tab.data(), hB0tab.size() * 4, hipMemcpyHostToDevice));
  float* dbh = dalloc<float>(NP); CHECK(hipMemcpy(dbh, hbh.data(), NP * 4, hipMemcpyHostToDevice));
  float* dbo = dalloc<float>(NP); CHECK(hipMemcpy(dbo, hbo.data(), NP * 4, hipMemcpyHostToDevice));
  float* dx0 = dalloc<float>(hx0.size()); CHECK(hipMemcpy(dx0, hx0.data(), hx0.size() * 4, hipMemcpyHostToDevice));
  float* dnoise = dalloc<float>(hnoise.size()); CHECK(hipMemcpy(dnoise, hnoise.data(), hnoise.size() * 4, hipMemcpyHostToDevice));
  int64_t* dt = dalloc<int64_t>(B); CHECK(hipMemcpy(dt, ht.data(), B * 8, hipMemcpyHostToDevice));
  uint8_t* dkeep = dalloc<uint8_t>(hkeep.size()); CHECK(hipMemcpy(dkeep, hkeep.data(), hkeep.size(), hipMemcpyHostToDevice));
  float* dsq = dalloc<float>(T + 1); CHECK(hipMemcpy(dsq, hsq.data(), (T + 1) * 4, hipMemcpyHostToDevice));
  float* dom = dalloc<float>(T + 1); CHECK(hipMemcpy(dom, hom.data(), (T + 1) * 4, hipMemcpyHostToDevice));
  float* dsl = dalloc<float>(2);
  { const float s2[2] = {slope0, slopeh}; CHECK(hipMemcpy(dsl, s2, 8, hipMemcpyHostToDevice)); }
  float* dU = dalloc<float>((size_t)MP * K0);
  float* dpre = dalloc<float>((size_t)(H + 1) * MP * NP);
  float* dY = dalloc<float>((size_t)MP * NP);
  float* dact = dalloc<float>((size_t)(H + 1) * MP * NP);
  int* dtdev = dalloc<int>(B);
  double* dpart = dalloc<double>((size_t)4 * G);

  RowChainArgs a{};
  a.x0 = dx0; a.noise = dnoise; a.t = dt; a.keep = dkeep; a.sqrt_ab = dsq; a.one_minus_ab = dom;
  a.B = B; a.L = L; a.T = T; a.H = H; a.mode = 0; a.seed_lo = 123u; a.seed_hi = 0u; a.step = 5u; a.row0 = 0; a.nd = 1.f;
  a.W0f = dWf[0]; a.Whf = dWf[1]; a.Wof = dWf[2]; a.bh = dbh; a.bo = dbo; a.B0tab = dB0tab; a.ldtab = NP;
  a.slope0 = dsl; a.slopeh = dsl + 1;
  a.U = dU; a.K0 = K0; a.LPs = NP; a.tdev = dtdev; a.pre = dpre; a.pre_stride = (size_t)MP * NP; a.ldp = NP; a.Y = dY; a.ldy = NP;
  a.loss_part = dpart; a.act = dact; a.ones_col = -1; a.light = 1;

  auto launch_row = [&](int mode) {
    RowChainArgs b = a;
    b.mode = mode;
    if (mode) { b.noise = nullptr; b.t = nullptr; b.keep = nullptr; }
    hipLaunchKernelGGL((k_row_fwd<CT, true>), dim3(G), dim3(NTHREADS), 0, 0, b);
  };
  // the per-layer path's three NT launches on the same shapes (inputs: whatever the row kernel left in U / pre)
  auto launch_layers = [&]() {
    for (int l = 0; l < 3; ++l) {
      GemmArgs ga{};
      ga.A = l == 0 ? dU : dpre + (size_t)(l - 1) * MP * NP; ga.lda = l == 0 ? K0 : NP; ga.limA = MP;
      ga.B = dWc[l]; ga.ldb = NP; ga.limB = NP;
      ga.C = l < 2 ? dpre + (size_t)l * MP * NP : dY; ga.ldc = NP; ga.K = NP; ga.kchunk = NP;
      ga.bias = l == 2 ? dbo : dbh; ga.slopeA = l == 1 ? dsl : dsl + 1;
      const int tiles_m = (MP + 63) / 64, tiles_n = (NP + 63) / 64;
      if (!gemm_set_grid(ga, tiles_m, tiles_n, 1)) { fprintf(stderr, "grid too large\n"); exit(1); }
      ga.rows_valid = MP; ga.cols_valid = NP;
      if (l == 0) hipLaunchKernelGGL((gemm_kernel<Cfg0, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_BIAS>), dim3(ga.nblocks), dim3(256), 0, 0, ga);
      else if (l == 1) hipLaunchKernelGGL((gemm_kernel<Cfg0, LD_KCONTIG, LD_KCONTIG, XF_PRELU, XF_NONE, EPI_BIAS>), dim3(ga.nblocks), dim3(256), 0, 0, ga);
      else hipLaunchKernelGGL((gemm_kernel<Cfg0, LD_KCONTIG, LD_KCONTIG, XF_PRELU, XF_NONE, EPI_BIAS_TANH>), dim3(ga.nblocks), dim3(256), 0, 0, ga);
    }
  };

  // ---- correctness (EXPLICIT randoms) against an fp64 host reference on a sample of users
  launch_row(0);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  std::vector<float> gU((size_t)MP * K0), gpre((size_t)2 * MP * NP), gY((size_t)MP * NP);
  std::vector<double> gpart((size_t)4 * G);
  CHECK(hipMemcpy(gU.data(), dU, gU.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(gpre.data(), dpre, gpre.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(gY.data(), dY, gY.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(gpart.data(), dpart, gpart.size() * 8, hipMemcpyDeviceToHost));
  double eU = 0, e0 = 0, e1 = 0, eY = 0, m0 = 0, m1 = 0, mY = 0;
  const int users[] = {0, 1, 17, 31, 32, 33, B / 2 + 5, B - 33, B - 1};
  double sums_ref[4] = {0, 0, 0, 0};
  std::vector<double> Yref((size_t)3 * L);
  for (int usr : users) {
    if (usr < 0 || usr >= B) continue;
    const int tt = (int)ht[usr];
    for (int pass = 0; pass < 3; ++pass) {
      const size_t row = (size_t)rc_row(pass, usr);
      std::vector<double> u(NP, 0.0), p0(NP, 0.0), p1(NP, 0.0), y(NP, 0.0);
      for (int c = 0; c < L; ++c) {
        const size_t idx = (size_t)usr * L + c;
        const float x = hx0[idx], ee = hnoise[idx];
        const float v = pass == 0 ? hsq[tt] * x + hom[tt] * ee : (pass == 1 ? x : x + 0.1f * ee);
        u[c] = hkeep[(size_t)pass * B * L + idx] ? 2.0 * v : 0.0;
        eU = std::max(eU, std::fabs(u[c] - (double)gU[row * K0 + c]));
      }
      for (int h = 0; h < TP; ++h) eU = std::max(eU, std::fabs((h == tt ? 1.0 : 0.0) - (double)gU[row * K0 + NP + h]));
      for (int n = 0; n < W; ++n) {
        double s = hB0tab[(size_t)tt * NP + n];
        for (int k = 0; k < L; ++k) s += u[k] * hW[0][(size_t)n * NP + k];
        p0[n] = s;
        e0 = std::max(e0, std::fabs(s - (double)gpre[row * NP + n])); m0 = std::max(m0, std::fabs(s));
      }
      for (int n = 0; n < W; ++n) {
        double s = hbh[n];
        for (int k = 0; k < W; ++k) s += (p0[k] > 0 ? p0[k] : slope0 * p0[k]) * hW[1][(size_t)n * NP + k];
        p1[n] = s;
        e1 = std::max(e1, std::fabs(s - (double)gpre[(size_t)MP * NP + row * NP + n])); m1 = std::max(m1, std::fabs(s));
      }
      for (int n = 0; n < L; ++n) {
        double s = hbo[n];
        for (int k = 0; k < W; ++k) s += (p1[k] > 0 ? p1[k] : slopeh * p1[k]) * hW[2][(size_t)n * NP + k];
        y[n] = std::tanh(s);
        eY = std::max(eY, std::fabs(y[n] - (double)gY[row * NP + n])); mY = std::max(mY, std::fabs(y[n]));
        Yref[(size_t)pass * L + n] = y[n];
      }
    }
  }
  // the activation copies: prelu of the GPU's own pre-activations, every element of every group's rows
  double eA = 0;
  {
    std::vector<float> gact((size_t)2 * MP * NP);
    CHECK(hipMemcpy(gact.data(), dact, gact.size() * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < 2; ++l)
      for (size_t r = 0; r < (size_t)G * RC_ROWS; ++r)
        for (int c = 0; c < NP; ++c) {
          const float p = gpre[(size_t)l * MP * NP + r * NP + c], sl = l == 0 ? slope0 : slopeh;
          eA = std::max(eA, std::fabs((double)(p > 0.f ? p : sl * p) - (double)gact[(size_t)l * MP * NP + r * NP + c]));
        }
  }
  // loss partial sums of every group against the GPU's own Y (fp64 on the host)
  double eS = 0, mS = 0;
  for (int g = 0; g < G; ++g) {
    double s[4] = {0, 0, 0, 0};
    for (int u = 0; u < RC_USERS; ++u) {
      const int usr = g * RC_USERS + u;
      if (usr >= B) continue;
      for (int c = 0; c < L; ++c) {
        const double P = gY[(size_t)rc_row(0, usr) * NP + c], S = gY[(size_t)rc_row(1, usr) * NP + c], Q = gY[(size_t)rc_row(2, usr) * NP + c];
        const double R = P - hx0[(size_t)usr * L + c], D = (Q - S) / 0.01 - R;
        s[0] += D * D; s[1] += (R - S) * (R - S); s[2] += R; s[3] += R * R;
      }
    }
    for (int j = 0; j < 4; ++j) { eS = std::max(eS, std::fabs(s[j] - gpart[4 * (size_t)g + j])); mS = std::max(mS, std::fabs(s[j])); sums_ref[j] += s[j]; }
  }
  printf("# B=%d G=%d work-groups, LDS %zu B per work-group\n", B, G, RowChainCfg<CT>::LDS_BYTES);
  printf("max|act - prelu(pre)| %.2e\n", eA);
  printf("max|err|: U %.2e  pre0 %.2e (max %.2f)  pre1 %.2e (max %.2f)  Y %.2e (max %.2f)  loss partials %.2e (max %.3e)\n", eU, e0, m0, e1, m1,
         eY, mY, eS, mS);
  const bool ok = eA < 1e-6 && eU < 1e-6 && e0 < 1e-4 * m0 && e1 < 1e-4 * m1 && eY < 1e-4 * mY && eS < 1e-3 * mS;   // float32 partial sums: per-quad f32, then f64
  printf("parity %s\n", ok ? "OK" : "FAILED");

#ifdef RC_STAMPS
  {
    unsigned long long* dst = dalloc<unsigned long long>((size_t)16 * G);
    RowChainArgs b = a;
    b.mode = 1; b.noise = nullptr; b.t = nullptr; b.keep = nullptr; b.stamps = dst;
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((k_row_fwd<CT, true>), dim3(G), dim3(NTHREADS), 0, 0, b);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)16 * G);
    CHECK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
    const char* seg[10] = {"staging", "L0 loop", "L0 barrier", "L0 epilogue", "L1 loop", "L1 barrier", "L1 epilogue", "L2 loop", "Y + loss terms",
                           "loss fold"};
    printf("# stamps (PHILOX staging, last of 20 back-to-back launches): cycles per segment, median / p90 / max over %d work-groups\n", G);
    for (int sgm = 0; sgm < 10; ++sgm) {
      std::vector<long long> d;
      for (int g = 0; g < G; ++g) d.push_back((long long)(hs[16 * (size_t)g + sgm + 1] - hs[16 * (size_t)g + sgm]));
      std::sort(d.begin(), d.end());
      printf("  %-14s %8lld %8lld %8lld\n", seg[sgm], d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    }
    std::vector<long long> tot;
    unsigned long long first = ~0ull, lastt = 0;
    for (int g = 0; g < G; ++g) { tot.push_back((long long)(hs[16 * (size_t)g + 10] - hs[16 * (size_t)g])); first = std::min(first, hs[16 * (size_t)g]); lastt = std::max(lastt, hs[16 * (size_t)g + 10]); }
    std::sort(tot.begin(), tot.end());
    printf("  %-14s %8lld %8lld %8lld   first entry -> last exit %llu cycles\n", "work-group", tot[tot.size() / 2], tot[tot.size() * 9 / 10], tot.back(), lastt - first);
  }
#endif
  // ---- timing: variants interleaved in one process
  hipEvent_t ev0, ev1;
  CHECK(hipEventCreate(&ev0)); CHECK(hipEventCreate(&ev1));
  const int reps = quick ? 3 : 20, rounds = quick ? 1 : 7;
  std::vector<float> us[3];
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < 3; ++v) {
      auto go = [&]() { if (v == 0) launch_row(0); else if (v == 1) launch_row(1); else launch_layers(); };
      go();
      CHECK(hipEventRecord(ev0, 0));
      for (int k = 0; k < reps; ++k) go();
      CHECK(hipEventRecord(ev1, 0));
      CHECK(hipEventSynchronize(ev1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, ev0, ev1));
      us[v].push_back(ms * 1e3f / reps);
    }
  const double fl = 2.0 * 3 * B * ((double)W * L + (double)W * W + (double)L * W);
  const char* names[3] = {"row-owned, EXPLICIT staging", "row-owned, PHILOX staging", "per-layer: 3 NT launches (no staging)"};
  for (int v = 0; v < 3; ++v) {
    std::sort(us[v].begin(), us[v].end());
    printf("%-40s med %7.2f us  min %7.2f us  (%5.1f TF on the unpadded dims)\n", names[v], us[v][us[v].size() / 2], us[v][0],
           fl / us[v][us[v].size() / 2] / 1e6);
  }
  return ok ? 0 : 1;
}
