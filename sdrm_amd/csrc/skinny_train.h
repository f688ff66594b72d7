// Train-step kernels for narrow eps-nets (padded widths LP, WP <= 64; the ADM/NeuMF configuration L = W = 40, H = 5,
// T = 93 of BASELINE.json).  A train step of such a net is ~0.3 GFLOP spread over ~20 launches of a few microseconds
// each; the rows of a batch are independent through every layer, and all weights fit in one CU's LDS, so
//   k_skinny_train_fwd : staging (q_sample + the three dropout masks, train_SDRM.py:326-331 / :100) and ALL H+2 layers of
//                        the three stacked passes in one launch;
//   k_skinny_train_bwd : the whole dgrad chain (out layer -> layer 0, PReLU' and the slope-gradient partials) in one
//                        launch.
// Ownership: a work-group owns 16 stacked rows, wave w of it owns the 16-column tile w of EVERY layer's output.  Its
// slices of the weight matrices (16 x 64 each) are MFMA B fragments held in registers for the whole kernel (12 VGPR
// quads, loaded once from L2), the 16 x 64 activation tile goes from layer to layer through two alternating LDS
// tiles with one barrier per layer.  A layer is then 16 MFMAs per wave instead of 64: the serial chain a row's
// H+2 layers form is what bounds these launches (a first version with one wave per 16 rows and all columns took
// 26 us per launch on 40 work-groups; global loads in its epilogues were exposed round trips on top).
// Loss, weight gradients (one batched launch), slab reduction, embedding backward and Adam are the general kernels.
// The time-embedding term of layer 0 comes from the per-timestep bias table B0tab[t] = b0 + C0[t] (what the sampler
// uses), so layer 0 contracts over the LP latent columns only; the one-hot columns of U are still written for the
// layer-0 weight gradient.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"
#include "skinny.h"

namespace sdrm {

struct SkinnyTrainArgs {
  // weights (padded compute copies) and tables
  const float* W0c; int K0; const float* Whc; const float* Woc; const float* bh; const float* bo;
  const float* WhcT; const float* WocT;   // [in][out] copies (backward)
  const float* B0tab;                     // [T+1][WP] = b0 + C0[t]
  const float* slope0; const float* slopeh;
  const float* sqrt_ab; const float* one_minus_ab;
  // step inputs
  const float* x0; const float* noise; const int64_t* t; const uint8_t* keep;   // EXPLICIT mode: [B,L], [B], [3,B,L]
  int B, L, W, T, H, MP;
  int LPs, WPs;   // padded widths = row strides (multiples of 32); the kernels' NL, NW count the 16-column tiles in use
  int mode; uint32_t seed_lo, seed_hi, step; int64_t row0; float nd;
  // activations
  float* U; int* tdev; float* pre; size_t pre_stride; float* Y;       // pre[k] = pre + k*pre_stride, [MP][WP]; Y [MP][LP]
  // backward
  const float* dY; float* dpre; float* alpha_part; int alpha_part_stride;   // dpre[k] = dpre + k*pre_stride
};

// NL, NW: 16-column tiles that hold real columns (ceil(L/16), ceil(W/16)); padding tiles beyond them stay zero.
template <int NL, int NW>
__global__ __launch_bounds__(64 * (NL > NW ? NL : NW)) void k_skinny_train_fwd(const SkinnyTrainArgs a) {
  constexpr int LP = 16 * NL, WP = 16 * NW, NV = NL > NW ? NL : NW;
  constexpr int SCR = (LP > WP ? LP : WP) + 4;
  __shared__ __attribute__((aligned(16))) float tile[2][16 * SCR];
  __shared__ int trow[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int s0 = blockIdx.x * 16;
  const int TP = a.K0 - a.LPs;

  f32x4 w0f[NL], whf[NW], wof[NW];
  float bhv = 0.f, bov = 0.f;
  if (wave < NW) {
    load_bfrags<NL>(a.W0c, (size_t)a.K0, wave, li, lq, w0f);
    load_bfrags<NW>(a.Whc, (size_t)a.WPs, wave, li, lq, whf);
    bhv = a.bh[wave * 16 + li];
  }
  if (wave < NL) {
    load_bfrags<NW>(a.Woc, (size_t)a.WPs, wave, li, lq, wof);
    bov = a.bo[wave * 16 + li];
  }
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;

  // staging: U = 2 * keep * {q_sample(x) | x | x + mu*eps}  (k_prep_train's arithmetic).  Lane -> (row, column quad):
  // one Philox call serves the four columns of a quad, stores are 16 bytes.
  if (wave < NL) {
    const int row = lane >> 2, c0 = wave * 16 + 4 * (lane & 3);
    const int s = s0 + row;
    const int pass = s / a.B, usr = s - pass * a.B;   // pass 0 P, 1 S, 2 Q; >= 3: padding rows
    int t0 = 0;
    f32x4 uv = {0.f, 0.f, 0.f, 0.f};
    if (pass < 3) {
      if (a.mode == 0) {
        t0 = (int)a.t[usr];
      } else {
        const U4 w = philox4x32_10((uint32_t)(a.row0 + usr), 0u, PURPOSE_TRAIN_T, a.step, a.seed_lo, a.seed_hi);
        t0 = 1 + (int)bounded(w.x, (uint32_t)a.T);
      }
      t0 = min(max(t0, 0), a.T);
      float ee[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t bits[4] = {0u, 0u, 0u, 0u};
      if (a.mode != 0 && c0 < a.L) {
        const U4 w = philox4x32_10((uint32_t)(a.row0 + usr), (uint32_t)(c0 >> 2), PURPOSE_TRAIN_ELEM, a.step, a.seed_lo, a.seed_hi);
        box_muller(w.x, w.y, ee[0], ee[1]);
        box_muller(w.z, w.w, ee[2], ee[3]);
        bits[0] = w.x; bits[1] = w.y; bits[2] = w.z; bits[3] = w.w;
      }
      const float sa = a.sqrt_ab[t0], sb = a.one_minus_ab[t0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = c0 + j;
        if (col < a.L) {
          const size_t idx = (size_t)usr * a.L + col;
          const float x = a.x0[idx];
          float e1; bool kp;
          if (a.mode == 0) {
            e1 = a.noise[idx];
            kp = a.keep[(size_t)pass * a.B * a.L + idx] != 0;
          } else {
            e1 = ee[j] * a.nd;
            kp = (bits[j] >> pass) & 1u;
          }
          const float v = pass == 0 ? sa * x + sb * e1 : (pass == 1 ? x : x + MU * e1);
          uv[j] = kp ? 2.f * v : 0.f;
        }
      }
      if (wave == 0 && pass == 0 && (lane & 3) == 0) a.tdev[usr] = t0;
    }
    if (wave == 0 && (lane & 3) == 0) trow[row] = pass < 3 ? t0 : -1;
    *reinterpret_cast<f32x4*>(&tile[0][row * SCR + c0]) = uv;
    *reinterpret_cast<f32x4*>(a.U + (size_t)s * a.K0 + c0) = uv;
  }
  __syncthreads();
  // one-hot(t) columns of U (read by the layer-0 weight gradient only)
  for (int f = tid; f < 16 * TP; f += 64 * NV) {
    const int row = f / TP, h = f - row * TP;
    a.U[(size_t)(s0 + row) * a.K0 + a.LPs + h] = h == trow[row] ? 1.f : 0.f;
  }

  const int col = wave * 16 + li;
  // layer 0: latent part by MFMA, time-embedding part + bias from the table row of the row's own timestep
  if (wave < NW) {
    float b0v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b0v[r] = a.B0tab[(size_t)max(trow[4 * lq + r], 0) * a.WPs + col];   // in flight under the MFMAs
    f32x4 af[NL];
    read_frags<NL, SCR>(tile[0], li, lq, af);
    const f32x4 acc = skinny_tile<NL>(af, w0f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = acc[r] + b0v[r];
      a.pre[(size_t)(s0 + 4 * lq + r) * a.WPs + col] = p;
      tile[1][(4 * lq + r) * SCR + col] = prelu_f(p, slope0);
    }
  }
  __syncthreads();
  int cur = 1;
  for (int h = 1; h <= a.H; ++h) {   // the shared hidden layer, H applications (Q1)
    if (wave < NW) {
      f32x4 af[NW];
      read_frags<NW, SCR>(tile[cur], li, lq, af);
      const f32x4 acc = skinny_tile<NW>(af, whf);
      float* ph = a.pre + (size_t)h * a.pre_stride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = acc[r] + bhv;
        ph[(size_t)(s0 + 4 * lq + r) * a.WPs + col] = p;
        tile[cur ^ 1][(4 * lq + r) * SCR + col] = prelu_f(p, slopeh);
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  if (wave < NL) {
    f32x4 af[NW];
    read_frags<NW, SCR>(tile[cur], li, lq, af);
    const f32x4 acc = skinny_tile<NW>(af, wof);
#pragma unroll
    for (int r = 0; r < 4; ++r) a.Y[(size_t)(s0 + 4 * lq + r) * a.LPs + col] = tanh_fast(acc[r] + bov);
  }
}

// The dgrad chain: dpre[H] = (dY * Wo) . prelu'(pre[H]);  dpre[k-1] = (dpre[k] * Wh) . prelu'(pre[k-1]), k = H..1.
// Products against the transposed weight copies ([in][out], so the contraction index is contiguous like in the forward).
// Slope-gradient partials: one float per (application, work-group), summed in a fixed order.
template <int NL, int NW>
__global__ __launch_bounds__(64 * (NL > NW ? NL : NW)) void k_skinny_train_bwd(const SkinnyTrainArgs a) {
  constexpr int LP = 16 * NL, WP = 16 * NW, NV = NL > NW ? NL : NW;
  constexpr int SCR = (LP > WP ? LP : WP) + 4;
  constexpr int MAXAPP = 32;
  __shared__ __attribute__((aligned(16))) float tile[2][16 * SCR];
  __shared__ float red[MAXAPP][NV];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int s0 = blockIdx.x * 16;
  const int col = wave * 16 + li;

  f32x4 woT[NL], whT[NW];
  if (wave < NW) {
    load_bfrags<NL>(a.WocT, (size_t)a.LPs, wave, li, lq, woT);
    load_bfrags<NW>(a.WhcT, (size_t)a.WPs, wave, li, lq, whT);
  }
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;
  if (wave < NL) {   // dY rows -> LDS, 16 bytes per lane
    const int row = lane >> 2, c0 = wave * 16 + 4 * (lane & 3);
    *reinterpret_cast<f32x4*>(&tile[0][row * SCR + c0]) = *reinterpret_cast<const f32x4*>(a.dY + (size_t)(s0 + row) * a.LPs + c0);
  }
  __syncthreads();
  int cur = 0;
  for (int k = a.H; k >= 0; --k) {   // produces dpre[k]
    if (wave < NW) {
      const float* pk = a.pre + (size_t)k * a.pre_stride;
      float* dk = a.dpre + (size_t)k * a.pre_stride;
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[r] = pk[(size_t)(s0 + 4 * lq + r) * a.WPs + col];   // in flight under the MFMAs
      f32x4 acc;
      if (k == a.H) {
        f32x4 af[NL];
        read_frags<NL, SCR>(tile[cur], li, lq, af);
        acc = skinny_tile<NL>(af, woT);
      } else {
        f32x4 af[NW];
        read_frags<NW, SCR>(tile[cur], li, lq, af);
        acc = skinny_tile<NW>(af, whT);
      }
      const float sl = k > 0 ? slopeh : slope0;
      float ssum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = pv[r], v = acc[r];
        const bool pos = p > 0.f;
        const float d = pos ? v : sl * v;
        ssum += pos ? 0.f : v * p;
        dk[(size_t)(s0 + 4 * lq + r) * a.WPs + col] = d;
        tile[cur ^ 1][(4 * lq + r) * SCR + col] = d;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ssum += __shfl_down(ssum, off, 64);
      if (lane == 0) red[k][wave] = ssum;
    }
    cur ^= 1;
    __syncthreads();
  }
  // per-application, per-work-group slope partials (the layout k_grad_finalize sums: [application][block])
  if (tid <= a.H) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[tid][w];
    a.alpha_part[(size_t)tid * a.alpha_part_stride + blockIdx.x] = sum;
  }
}

}  // namespace sdrm
