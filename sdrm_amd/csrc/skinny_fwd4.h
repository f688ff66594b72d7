// The train FORWARD of a narrow eps-net (padded widths <= 64) on 4-row MFMA units: v_mfma_f32_4x4x1_16B_f32 (round 4).
//
// k_skinny_fwd (skinny_step.h) owns 16 users per work-group - the smallest group of v_mfma_f32_16x16x4_f32 row tiles whose P, S
// and Q rows meet in one work-group - so ADM's 850 users are 54 work-groups on 54 of the 256 CUs, and a layer costs the matrix
// pipes of ONE CU 108 MFMAs of 32 cycles: 864 - 1152 cycles per layer and SIMD, nine waves through a barrier (2.6 k cycles
// measured per layer, 8.7 us for the seven layers of ADM).  The 4x4x1 instruction multiplies 16 independent 4 x 4 blocks at the
// same rate (tools/mfma4x4_layout_probe.hip: D[lane 4b + l][reg j] = A[lane 4b + j] * B[lane 4b + l], 8.7 cycles per
// instruction with independent accumulators, 15.6 in one dependent chain), which makes FOUR rows the unit of ownership:
//   * a work-group owns 4 USERS (12 stacked rows): 213 work-groups at ADM's batch, one per CU;
//   * wave p owns the 4 rows of pass p through ALL layers: srcA = one k-row of the layer's weights (lane l = output column l,
//     64 columns per instruction), srcB = the wave's 4 activations of that k (the same four values in every block), so lane
//     (b, l) ends up with row l, columns 4b .. 4b + 3 of the layer's output - 16 bytes: the pre-activation store, and after
//     PReLU the wave's own LDS tile, from where the next layer's srcB quads come back as broadcast reads.  No other wave
//     ever touches that tile: NO barrier between layers, a layer is K MFMAs in two chains + one LDS round trip;
//   * the weights sit in LDS once per work-group as [k][64 columns] images (transposed on the way in, row stride 65);
//   * staging, the users' rows of the layer-0 table (E = temb[t] We^T + be, b0 + E W0e^T: the 16x16x4 products of skinny_step.h,
//     rows 4 .. 15 of their tiles idle) and the loss partial sums as there; ONE partial per 4 users.
// The backward (k_skinny_bwd, 16 users per work-group) reads what this kernel leaves in the same grouped-by-16 stacked order:
// row(pass, user) = 48 (user / 16) + 16 pass + user % 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skinny_step.h"

namespace sdrm {

constexpr int SK4_USERS = 4;
constexpr int SK4_WS = 65;   // LDS row stride of a weight image [k][64 columns]: odd, so the transposing stores spread over the banks
constexpr int SK4_XS = 68;   // row stride of the [4][64] activation tiles (16-byte rows)
constexpr int SK4_THREADS = 192;

template <int NL, int NW>
__host__ __device__ inline size_t sk4_fwd_lds_floats(int TPe) {
  // weight images (16 NL + 32 NW k-rows), X tiles [3][4], Y tiles [3][4], x0 [4], B0 rows [16], trow [16], loss sums [3][4] doubles;
  // intab: E rows and temb rows [16][TPe + 4] each, emb_layer.weight [TPe][TPe + 4], W0e [16 NW][TPe + 4]
  return (size_t)(16 * NL + 32 * NW) * SK4_WS + (size_t)(12 + 12 + 4 + 16) * SK4_XS + 16 + 24 +
         (TPe > 0 ? (size_t)(32 + TPe + 16 * NW) * (TPe + 4) : 0) + 8;
}

// one layer of a wave's four rows: K = 4 KQ, srcA = the wave's register copy of the weight image (w[k]: lane l = output column
// l), activations X [4][SK4_XS] (the wave's own LDS tile); lane (b = lane / 4, l = lane % 4) gets row l, columns 4b .. 4b + 3
template <int KQ>
__device__ __forceinline__ f32x4 sk4_layer(const float (&w)[4 * KQ], const float* __restrict__ X, int lane) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  const float* xrow = X + (lane & 3) * SK4_XS;
  f32x4 xq[KQ];
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq) xq[kq] = *reinterpret_cast<const f32x4*>(xrow + 4 * kq);
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq) {
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 0], xq[kq][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 1], xq[kq][1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 2], xq[kq][2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 3], xq[kq][3], acc1, 0, 0, 0);
  }
  // every LDS read of the layer in front of its MFMAs (left alone the scheduler pairs each read with its four MFMAs: KQ exposed
  // LDS round trips per layer)
  __builtin_amdgcn_sched_group_barrier(0x100, KQ, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 4 * KQ, 0);
  return f32x4{acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
}

// [out][in] padded copy (row stride ld) -> LDS image [k][SK4_WS] for k < 4 KQ, columns < NC: 16-byte global reads along k.
// Two phases, so that every read of the kernel's prologue is in flight before the first value is used (a work-group is a
// chain of dependent memory round trips - 3 k cycles each behind another kernel's stores: a loop of load-then-store
// iterations pays one per iteration).
template <int KQ, int NC, int NB>
__device__ __forceinline__ void sk4_weights_request(f32x4 (&v)[NB], const float* __restrict__ Wc, int ld, int tid) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {   // (unconditional reads at clamped indices: a read inside a branch is waited for at the branch's end)
    const int f = min(tid + u * SK4_THREADS, NC * KQ - 1), c = f / KQ, kq = f - c * KQ;
    v[u] = *reinterpret_cast<const f32x4*>(Wc + (size_t)c * ld + 4 * kq);
  }
}
template <int KQ, int NC, int NB>
__device__ __forceinline__ void sk4_weights_store(const f32x4 (&v)[NB], float* __restrict__ Wl, int tid) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int f = tid + u * SK4_THREADS, c = f / KQ, kq = f - c * KQ;
    if (f < NC * KQ) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Wl[(4 * kq + e) * SK4_WS + c] = v[u][e];
    }
  }
}

constexpr int SK4_TAB_BATCH = 24;   // 16-byte reads of the table operands a thread keeps in flight: rows of 32 quads (no division by
                                    // the row length), 144 rows = emb_layer.weight and W0e at T <= 96, W <= 48

template <int NL, int NW>
__global__ __launch_bounds__(SK4_THREADS) void k_skinny_fwd4(const SkStepArgs a) {
  constexpr int LPk = 16 * NL, WPk = 16 * NW, XS = SK4_XS;
  extern __shared__ __attribute__((aligned(16))) float sk4sh[];
  float* Wl0 = sk4sh;                              // [LPk][65]: dnn.0.weight[:, :L]^T
  float* Wlh = Wl0 + LPk * SK4_WS;                 // [WPk][65]: the hidden layer's weight^T
  float* Wlo = Wlh + WPk * SK4_WS;                 // [WPk][65]: the out layer's weight^T
  float* Xt = Wlo + WPk * SK4_WS;                  // [3 passes][4][XS]: a wave's activation tile
  float* Yt = Xt + 12 * XS;                        // [3][4][XS]: tanh outputs (loss sums)
  float* x0s = Yt + 12 * XS;                       // [4][XS]
  float* B0s = x0s + 4 * XS;                       // [16][XS]: b0 + C0[t_user] (rows 0 .. 3 are the users)
  int* trow = reinterpret_cast<int*>(B0s + 16 * XS);                      // [16]
  double* red = reinterpret_cast<double*>(B0s + 16 * XS + 16);            // [3 waves][4]
  float* Es = B0s + 16 * XS + 16 + 24;             // intab: [16][TPe + 4]
  float* Ts = Es + 16 * (a.TPe + 4);               //        [16][TPe + 4]
  float* WeS = Ts + 16 * (a.TPe + 4);              //        [TPe][TPe + 4]
  float* W0eS = WeS + a.TPe * (a.TPe + 4);         //        [WPk][TPe + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int pass = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;         // 16x16x4 fragments (the table products)
  const int b4 = lane >> 2, l4 = lane & 3;          // 4x4x1 blocks: lane (b4, l4) = row l4, columns 4 b4 ..
  const int c4 = 4 * b4;
  const bool wcols = c4 < WPk, lcols = c4 < LPk;
  const int wlw = lane < WPk ? lane : 0, wll = lane < LPk ? lane : 0;   // srcA lane -> output column (idle lanes re-read column 0)
  const int nke = a.TPe >> 4, lde = a.TPe + 4;
  const int qpr = a.TPe >> 2;   // 16-byte pieces of a row of emb_layer.weight / W0e

  // ---- every read of the prologue that does not depend on the users' timesteps, requested now: the three weight matrices,
  // the operands of the table products, biases, slopes
  constexpr int NB0 = (WPk * (LPk / 4) + SK4_THREADS - 1) / SK4_THREADS, NBH = (WPk * (WPk / 4) + SK4_THREADS - 1) / SK4_THREADS,
                NBO = (LPk * (WPk / 4) + SK4_THREADS - 1) / SK4_THREADS;
  f32x4 v0[NB0], vh[NBH], vo[NBO], vt[SK4_TAB_BATCH];
  sk4_weights_request<LPk / 4, WPk, NB0>(v0, a.W0c, a.K0, tid);
  sk4_weights_request<WPk / 4, WPk, NBH>(vh, a.Whc, a.WPs, tid);
  sk4_weights_request<WPk / 4, LPk, NBO>(vo, a.Woc, a.WPs, tid);
  // (thread -> row f / 32, quad f % 32 of a row of at most 32 quads: no division by the row length; quads beyond it idle)
  const int ntr = a.intab ? a.TPe + WPk : 0;   // rows: emb_layer.weight, then W0e
  auto tab_src = [&](int r, int q) __attribute__((always_inline)) {
    return r < a.TPe ? a.WeP + (size_t)r * a.TPe + 4 * q : a.W0eP + (size_t)(r - a.TPe) * a.TPe + 4 * q;
  };
  if (a.intab) {
#pragma unroll
    for (int u = 0; u < SK4_TAB_BATCH; ++u) {
      const int f = tid + u * SK4_THREADS;
      vt[u] = *reinterpret_cast<const f32x4*>(tab_src(min(f >> 5, ntr - 1), min(f & 31, qpr - 1)));
    }
  }
  f32x4 bh4 = {0.f, 0.f, 0.f, 0.f}, bo4 = bh4;
  bh4 = *reinterpret_cast<const f32x4*>(a.bh + (wcols ? c4 : 0));   // (idle lanes: column 0, never stored)
  bo4 = *reinterpret_cast<const f32x4*>(a.bo + (lcols ? c4 : 0));
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;
  float bev[3] = {0.f, 0.f, 0.f}, b0c[2] = {0.f, 0.f};   // emb_layer.bias / dnn.0.bias at this wave's column tiles of the table products
  if (a.intab) {
#pragma unroll
    for (int i = 0; i < 3; ++i) bev[i] = a.be[min((pass + 3 * i) * 16 + li, a.T - 1)];   // (clamped: used only where the column is real)
#pragma unroll
    for (int i = 0; i < 2; ++i) b0c[i] = a.b0[min(pass + 3 * i, NW - 1) * 16 + li];
  }
  bool first = true;
  float w0r[LPk], whr[WPk], wor[WPk];   // this wave's srcA operands: k-rows of the three weight images, lane = output column
  const int G4 = a.NP;   // 4 x ceil(B / 16): every row of every 16-user group of the backward is written, users or not
  for (int g = blockIdx.x; g < G4; g += gridDim.x) {
    const int u0 = SK4_USERS * g;
    // first stacked row of this wave's four: grouped-by-16 order
    const size_t prow0 = (size_t)SK_ROWS * (u0 / SK_USERS) + 16 * pass + (u0 % SK_USERS);
    // ---- requests first.  Staging lane -> (user lane / 16, column quad lane % 16)
    const int ur = lane >> 4, c0 = 4 * (lane & 15);
    const int usr = u0 + ur;
    const bool stg = c0 < LPk && usr < a.B;
    float xs[4] = {0.f, 0.f, 0.f, 0.f}, ens[4] = {0.f, 0.f, 0.f, 0.f};
    bool kps[4] = {false, false, false, false};
    if (stg) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = c0 + j;
        if (cc < a.L) {
          const size_t idx = (size_t)usr * a.L + cc;
          xs[j] = a.x0[idx];
          if (a.mode == 0) { ens[j] = a.noise[idx]; kps[j] = a.keep[(size_t)pass * a.B * a.L + idx] != 0; }
        }
      }
    }
    if (tid < 16) {   // the users' timesteps (train_SDRM.py:327), once per work-group; entries 4 .. 15: no user
      const int uu = u0 + tid;
      int t0 = -1;
      if (tid < SK4_USERS && uu < a.B) {
        if (a.mode == 0) {
          t0 = (int)a.t[uu];
        } else {
          const U4 w = philox4x32_10((uint32_t)(a.row0 + uu), 0u, PURPOSE_TRAIN_T, a.step, a.seed_lo, a.seed_hi);
          t0 = 1 + (int)bounded(w.x, (uint32_t)a.T);
        }
        t0 = min(max(t0, 0), a.T);
        a.tdev[uu] = t0;
      }
      trow[tid] = t0;
    }
    lds_barrier();
    f32x4 tq = {0.f, 0.f, 0.f, 0.f};
    if (a.intab) {   // at most 32 quads per row: the four users' temb rows are at most 128 threads' worth
      if (tid < 4 * qpr && trow[tid / qpr] >= 0) tq = *reinterpret_cast<const f32x4*>(a.tembP + (size_t)trow[tid / qpr] * a.TPs + 4 * (tid % qpr));
    } else if (tid < 4 * 16) {      // the table rows as they are: thread -> (user tid / 16, column quad tid % 16)
      const int r = tid >> 4, q = tid & 15;
      if (4 * q < WPk) tq = *reinterpret_cast<const f32x4*>(a.B0tab + (size_t)max(trow[r], 0) * a.WPs + 4 * q);
    }
    if (first) {
      // the prologue's reads land: weight images and table operands -> LDS (everything was requested before the first wait)
      sk4_weights_store<LPk / 4, WPk, NB0>(v0, Wl0, tid);
      sk4_weights_store<WPk / 4, WPk, NBH>(vh, Wlh, tid);
      sk4_weights_store<WPk / 4, LPk, NBO>(vo, Wlo, tid);
      if (a.intab) {
#pragma unroll
        for (int u = 0; u < SK4_TAB_BATCH; ++u) {
          const int f = tid + u * SK4_THREADS, r = f >> 5, q = f & 31;
          if (r < ntr && q < qpr) *reinterpret_cast<f32x4*>(WeS + r * lde + 4 * q) = vt[u];   // (W0eS follows WeS with the same row stride)
        }
        for (int f = tid + SK4_TAB_BATCH * SK4_THREADS; (f >> 5) < ntr; f += SK4_THREADS) {   // (more than 144 rows: the rest, a round trip each)
          const int r = f >> 5, q = f & 31;
          if (q < qpr) *reinterpret_cast<f32x4*>(WeS + r * lde + 4 * q) = *reinterpret_cast<const f32x4*>(tab_src(r, q));
        }
      }
      if (a.intab)
        for (int f = tid; f < 12 * lde; f += SK4_THREADS) Ts[4 * lde + f] = 0.f;   // rows 4 .. 15 of the temb tile: no users
    }
    // ---- staging: the three pass waves draw the same Philox words (the noise is shared by the passes, the keep bits are
    // bits 0 .. 2 of the same words)
    {
      f32x4 uv = {0.f, 0.f, 0.f, 0.f};
      if (stg) {
        const int t0 = trow[ur];
        if (a.mode != 0 && c0 < a.L) {
          const U4 w = philox4x32_10((uint32_t)(a.row0 + usr), (uint32_t)(c0 >> 2), PURPOSE_TRAIN_ELEM, a.step, a.seed_lo, a.seed_hi);
          box_muller(w.x, w.y, ens[0], ens[1]);
          box_muller(w.z, w.w, ens[2], ens[3]);
          const uint32_t bits[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { ens[j] *= a.nd; kps[j] = (bits[j] >> pass) & 1u; }
        }
        const float sa = a.sqrt_ab[t0], sb = a.one_minus_ab[t0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c0 + j < a.L) {
            const float x = xs[j], e1 = ens[j];
            const float v = pass == 0 ? sa * x + sb * e1 : (pass == 1 ? x : x + MU * e1);
            uv[j] = kps[j] ? 2.f * v : 0.f;
          }
        }
      }
      if (c0 < LPk) {
        *reinterpret_cast<f32x4*>(Xt + (4 * pass + ur) * XS + c0) = uv;
        *reinterpret_cast<f32x4*>(a.U + (prow0 + ur) * a.K0 + c0) = uv;
        if (pass == 1) *reinterpret_cast<f32x4*>(x0s + ur * XS + c0) = f32x4{xs[0], xs[1], xs[2], xs[3]};   // x0 for the loss sums
      }
    }
    if (a.intab) {
      // the users' temb rows -> LDS; E rows (wave w: column tiles w, w + 3, ..); b0 + E * W0e^T (column tiles likewise)
      if (tid < 4 * qpr) *reinterpret_cast<f32x4*>(Ts + (tid / qpr) * lde + 4 * (tid % qpr)) = tq;
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int tl = pass + 3 * i;
        if (tl < nke) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
          for (int u = 0; u < nke; ++u) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(Ts + li * lde + 16 * u + 4 * lq);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(WeS + (tl * 16 + li) * lde + 16 * u + 4 * lq);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) Es[(4 * lq + r) * lde + tl * 16 + li] = tl * 16 + li < a.T ? acc0[r] + acc1[r] + bev[i] : 0.f;
        }
      }
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ct = pass + 3 * i;
        if (ct < NW) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
          for (int u = 0; u < nke; ++u) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(Es + li * lde + 16 * u + 4 * lq);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(W0eS + (ct * 16 + li) * lde + 16 * u + 4 * lq);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) B0s[(4 * lq + r) * XS + ct * 16 + li] = acc0[r] + acc1[r] + b0c[i];
        }
      }
    } else if (tid < 4 * 16) {
      *reinterpret_cast<f32x4*>(B0s + (tid >> 4) * XS + 4 * (tid & 15)) = tq;
    }
    lds_barrier();   // the staging tiles, the weight images (first group) and the table rows are complete
    if (first) {   // this wave's srcA operands out of the weight images, for every group it works on
#pragma unroll
      for (int k = 0; k < LPk; ++k) w0r[k] = Wl0[k * SK4_WS + wlw];
#pragma unroll
      for (int k = 0; k < WPk; ++k) { whr[k] = Wlh[k * SK4_WS + wlw]; wor[k] = Wlo[k * SK4_WS + wll]; }
      first = false;
    }

    // ---- the wave's four rows through all layers, no barrier in between: its tile is its own
    float* X = Xt + 4 * pass * XS;
    float sl0 = slope0, slh = slopeh;   // (opaque here: derived loop invariants would otherwise be made - and every read of the
    asm volatile("" : "+v"(sl0), "+v"(slh));   // prologue waited for - in front of the loop)
    {
      const f32x4 acc = sk4_layer<LPk / 4>(w0r, X, lane);
      if (wcols) {
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(B0s + l4 * XS + c4);
        f32x4 p, v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p[j] = acc[j] + b0v[j]; v[j] = prelu_f(p[j], sl0); }
        *reinterpret_cast<f32x4*>(a.pre + (prow0 + l4) * a.WPs + c4) = p;
        *reinterpret_cast<f32x4*>(X + l4 * XS + c4) = v;
      }
    }
    for (int h = 1; h <= a.H; ++h) {   // the shared hidden layer, H applications (Q1)
      const f32x4 acc = sk4_layer<WPk / 4>(whr, X, lane);
      if (wcols) {
        f32x4 p, v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p[j] = acc[j] + bh4[j]; v[j] = prelu_f(p[j], slh); }
        *reinterpret_cast<f32x4*>(a.pre + (size_t)h * a.pre_stride + (prow0 + l4) * a.WPs + c4) = p;
        *reinterpret_cast<f32x4*>(X + l4 * XS + c4) = v;
      }
    }
    {
      const f32x4 acc = sk4_layer<WPk / 4>(wor, X, lane);
      if (lcols) {
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = tanh_fast(acc[j] + bo4[j]);
        *reinterpret_cast<f32x4*>(a.Y + (prow0 + l4) * a.LPs + c4) = y;
        *reinterpret_cast<f32x4*>(Yt + (4 * pass + l4) * XS + c4) = y;
      }
    }
    lds_barrier();
    // ---- loss partial sums (:196-198): R = P - x0, D = (Q - S) / mu^2 - R, over the group's users and the real columns
    {
      double sD = 0, sC = 0, sR = 0, sR2 = 0;
      for (int f = tid; f < SK4_USERS * LPk; f += SK4_THREADS) {
        const int uq = f / LPk, c = f - uq * LPk;
        if (u0 + uq < a.B && c < a.L) {
          const float P = Yt[uq * XS + c], S = Yt[(4 + uq) * XS + c], Q = Yt[(8 + uq) * XS + c];
          const float R = P - x0s[uq * XS + c];
          const float D = (Q - S) / MU2 - R;
          const float RS = R - S;
          sD += (double)(D * D); sC += (double)(RS * RS); sR += (double)R; sR2 += (double)(R * R);
        }
      }
      double v4[4] = {sD, sC, sR, sR2};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v4[j] += __shfl_down(v4[j], off, 64);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[4 * pass + j] = v4[j];
      }
    }
    lds_barrier();
    if (tid < 4) a.loss_part[4 * (size_t)g + tid] = (red[tid] + red[4 + tid]) + red[8 + tid];
    lds_barrier();   // the next group's staging overwrites the tiles and trow
  }
}

}  // namespace sdrm
