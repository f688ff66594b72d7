// The train step of a narrow eps-net (padded widths <= 64: the ADM / NeuMF configuration L = W = 40, H = 5, T = 93 of
// BASELINE.json) in TWO launches in front of the tail (tail.h), round 4.  Rounds 1-3 ran six here (tables, forward, loss
// partial sums, loss seeds, dgrad chain, batched weight gradients), each at the 5-9 us floor of a small dependent kernel:
// the step was the length of its launch chain (70 us for 2 us of arithmetic).
//
// Ownership: a work-group owns 16 USERS - the P, S and Q rows of each (train_SDRM.py:331-333), 48 stacked rows - and runs
// 3 x NV waves: wave (pass, column tile) owns the 16 x 16 tile `column tile` of every layer's output for the 16 rows of
// `pass`, with its slices of the weight matrices held in registers as MFMA B fragments (skinny.h).  A row's chain through the
// layers is therefore as short as in the 16-row kernels of round 1 (12-16 MFMAs per link), but the three rows of a user
// meet in ONE work-group:
//   k_skinny_fwd  : staging (q_sample + three dropout masks, :326-331 / :100), all H + 2 layers, AND the loss partial sums
//                   (:196-198) out of the work-group's own Y tile;
//   k_skinny_bwd  : the five sums folded from those partials (or given, sharded step), the loss value and the gradient seeds
//                   (App. A.5), the whole dgrad chain, AND every weight gradient of the work-group's 48 rows
//                   (dW = dpre^T * input: the rows are the contraction axis, 12 MFMAs per 16 x 16 tile out of the LDS tiles
//                   the dgrad chain leaves behind; the shared hidden layer's applications accumulate in registers, Q1), the bias
//                   gradients (column sums in the dgrad epilogue) and the slope partials - written as ONE slab set per
//                   work-group, which the tail reduces.  Neither dpre nor the embedding columns of U ever reach memory.
// Stacked row order ("grouped by 16"): row(pass, user) = 48 * (user / 16) + 16 * pass + user % 16.
// The step's time-embedding table B0tab = b0 + C0[t]: with T <= 128 the forward makes its users' rows itself from the padded copies
// WeP / W0eP the tail keeps current (`intab`, no k_emb_tables launch); longer embeddings do not fit that LDS image and take the
// table from a k_emb_tables launch in front of the forward (sdrm_hip.hip: ensure_tables - one more launch per step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"
#include "skinny.h"

namespace sdrm {

constexpr int SK_USERS = 16;
constexpr int SK_ROWS = 48;

struct SkStepArgs {
  // net: padded compute copies, the per-timestep layer-0 bias table, slopes
  const float* W0c; int K0; const float* Whc; const float* Woc; const float* bh; const float* bo;
  const float* WhcT; const float* WocT;   // [in][out] copies (backward)
  const float* B0tab;                     // [T+1][WPs] = b0 + C0[t] (used when intab == 0)
  // intab: the forward makes its 16 users' rows of that table itself (no tables launch in front of the step): E = temb[t] * We^T
  // + be, then b0 + E * W0e^T, from the padded copies WeP [TPe][TPe], W0eP [WPs][TPe] (TPe = T rounded up to 16, at most 128)
  const float* WeP; const float* W0eP; const float* be; const float* b0; int TPe, intab;
  const float* slope0; const float* slopeh;
  const float* sqrt_ab; const float* one_minus_ab;
  const float* tembP;                     // [T+1][TPs] time-embedding table, rows padded with zeros
  // step inputs (EXPLICIT mode: noise [B,L], t [B], keep [3,B,L])
  const float* x0; const float* noise; const int64_t* t; const uint8_t* keep;
  int B, L, W, T, H, G;                   // G = ceil(B / 16) groups of users
  int LPs, WPs, TPs;                      // padded widths = row strides (multiples of 32)
  int mode; uint32_t seed_lo, seed_hi, step; int64_t row0; float nd;
  // activations, grouped-by-16 stacked rows
  float* U; int* tdev; float* pre; size_t pre_stride; float* Y;   // U [MP][K0] (latent columns only), pre[k] [MP][WPs], Y [MP][LPs]
  double* loss_part;                      // [NP][4]
  int NP;                                 // loss partials the forward leaves: G (k_skinny_fwd) or 4 G (k_skinny_fwd4, one per 4 users)
  // backward
  const double* sums; double count; float* loss;
  float* slab0; float* slabH; float* slabO;        // [S][WPs][K0], [S][WPs][WPs], [S][LPs][WPs]: S = gridDim.x slab sets
  float* db0s; float* dbHs; float* dbOs;           // [S][WPs], [S][WPs], [S][LPs]
  float* alpha_part; int alpha_part_stride;        // [application][alpha_part_stride], entry = work-group
};

template <int NL, int NW>
struct SkCfg {
  static constexpr int NV = NL > NW ? NL : NW;
  static constexpr int NWAVES = 3 * NV, NTHR = 64 * NWAVES;
  static constexpr int LPk = 16 * NL, WPk = 16 * NW;
  static constexpr int SCR = 16 * NV + 4;           // LDS row stride of a 48-row tile (floats): 16-byte rows, odd multiple of 4 banks
  static constexpr int TILE = SK_ROWS * SCR;
};

// dynamic LDS (floats): forward: two tiles; backward: two gradient tiles, two input tiles, the pass-summed dpre0 tile [16][SCR],
// the users' time-embedding rows [16][TPs + 4], bias column sums [3 kinds][3 passes][64], slope partials [32 applications][12 waves]
template <int NL, int NW>
__host__ __device__ inline size_t sk_fwd_lds_floats(int TPe) {   // + trow [16], loss sums [12 waves][4] doubles, x0 [16][SCR], B0 rows [16][SCR];
  // intab (TPe > 0): E rows and temb rows [16][TPe + 4] each, emb_layer.weight [TPe][TPe + 4], W0e [16 NW][TPe + 4]
  return 2 * (size_t)SkCfg<NL, NW>::TILE + 16 + 2 * 4 * 12 + 2 * 16 * (size_t)SkCfg<NL, NW>::SCR +
         (TPe > 0 ? (size_t)(32 + TPe + SkCfg<NL, NW>::WPk) * (TPe + 4) : 0);
}
template <int NL, int NW>
__host__ __device__ inline size_t sk_bwd_lds_floats(int TPs) {
  return 4 * (size_t)SkCfg<NL, NW>::TILE + SK_USERS * (size_t)SkCfg<NL, NW>::SCR + SK_USERS * (size_t)(TPs + 4) + 3 * 3 * 64 + 32 * 12 + 64;
}

// ================================================================================================ forward
template <int NL, int NW>
__global__ __launch_bounds__(64 * 3 * (NL > NW ? NL : NW)) void k_skinny_fwd(const SkStepArgs a) {
  typedef SkCfg<NL, NW> C;
  constexpr int SCR = C::SCR, NTHR = C::NTHR, NV = C::NV;
  extern __shared__ __attribute__((aligned(16))) float sksh[];
  float* tile0 = sksh;
  float* tile1 = sksh + C::TILE;
  int* trow = reinterpret_cast<int*>(sksh + 2 * C::TILE);          // [16]
  double* red = reinterpret_cast<double*>(sksh + 2 * C::TILE + 16);   // [12 waves][4] (8-byte aligned: TILE is a multiple of 4 floats)
  float* x0s = sksh + 2 * C::TILE + 16 + 2 * 4 * 12;                   // [16][SCR]: the group's x0 rows (loss sums)
  float* B0s = x0s + 16 * SCR;                                         // [16][SCR]: b0 + C0[t_user] (intab)
  float* Es = B0s + 16 * SCR;                                          // [16][TPe + 4]: E[t_user] (intab)
  float* Ts = Es + 16 * (a.TPe + 4);                                   // [16][TPe + 4]: temb[t_user] (intab)
  float* WeS = Ts + 16 * (a.TPe + 4);                                  // [TPe][TPe + 4]: emb_layer.weight (intab; loaded once per work-group)
  float* W0eS = WeS + a.TPe * (a.TPe + 4);                             // [16 NW][TPe + 4]: W0e (intab)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pass = wave / NV, ct = wave - pass * NV;
  const int li = lane & 15, lq = lane >> 4;
  const int col = ct * 16 + li;

  // this wave's slices of the three weight matrices, as MFMA B fragments, for every group it works on
  f32x4 w0f[NL], whf[NW], wof[NW];
  float bhv = 0.f, bov = 0.f;
  if (ct < NW) {
    load_bfrags<NL>(a.W0c, (size_t)a.K0, ct, li, lq, w0f);
    load_bfrags<NW>(a.Whc, (size_t)a.WPs, ct, li, lq, whf);
    bhv = a.bh[col];
  }
  if (ct < NL) {
    load_bfrags<NW>(a.Woc, (size_t)a.WPs, ct, li, lq, wof);
    bov = a.bo[col];
  }
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;
  // intab: emb_layer.weight and W0e into LDS, once per work-group (B operands of the two embedding products: out of registers
  // they pushed the kernel past its 168-register budget, 240 bytes of scratch per lane and 10 us)
  const int nke = a.TPe >> 4, lde = a.TPe + 4;
  float b0c_ = 0.f;
  if (a.intab) {
    const int qpr = a.TPe >> 2;
    for (int f = tid; f < (a.TPe + C::WPk) * qpr; f += NTHR) {
      const int r = f / qpr, q = f - r * qpr;
      const float* src = r < a.TPe ? a.WeP + (size_t)r * a.TPe + 4 * q : a.W0eP + (size_t)(r - a.TPe) * a.TPe + 4 * q;
      *reinterpret_cast<f32x4*>(WeS + r * lde + 4 * q) = *reinterpret_cast<const f32x4*>(src);   // (W0eS follows WeS with the same row stride)
    }
    if (pass == 0 && ct < NW) b0c_ = a.b0[col];
  }
  for (int g = blockIdx.x; g < a.G; g += gridDim.x) {
    const int u0 = SK_USERS * g;
    const size_t grow0 = (size_t)SK_ROWS * g;
    // ---- requests first: a work-group is a chain of dependent memory round trips (about 2 us each), so everything the staging
    // needs is asked for before anything is computed.  Staging lane -> (user lane / 4, column quad ct * 4 + lane % 4).
    const int ur = lane >> 2, c0 = ct * 16 + 4 * (lane & 3);
    const int usr = u0 + ur;
    const bool stg = ct < NL && usr < a.B;
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
    if (tid < SK_USERS) {   // the 16 users' timesteps (train_SDRM.py:327), once per work-group
      const int uu = u0 + tid;
      int t0 = -1;
      if (uu < a.B) {
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
    // layer 0's time-embedding term + bias: the table row of each accumulator row's timestep - in flight under the staging.
    // intab: the operands of E = temb[t_user] * We^T instead (wave w < TPe / 16 makes column tile w of it): the users' temb rows
    // as A fragments, the wave's rows of We as B fragments, all requested now
    float b0v[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 tq = {0.f, 0.f, 0.f, 0.f};   // intab: this thread's 16 bytes of the users' temb rows (requested now, to LDS behind the staging)
    float bev = 0.f;
    if (!a.intab) {
      if (ct < NW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b0v[r] = a.B0tab[(size_t)max(trow[4 * lq + r], 0) * a.WPs + col];
      }
    } else {
      const int qpr = a.TPe >> 2;   // at most 32 quads per row: 16 rows are at most 512 threads' worth (NTHR >= 192: a loop)
      if (tid < 16 * qpr && trow[tid / qpr] >= 0) tq = *reinterpret_cast<const f32x4*>(a.tembP + (size_t)trow[tid / qpr] * a.TPs + 4 * (tid % qpr));
      if (wave < nke) bev = wave * 16 + li < a.T ? a.be[wave * 16 + li] : 0.f;
    }
    // ---- staging: the three pass waves of a column tile draw the same Philox words (the noise is shared by the passes, the keep
    // bits are bits 0..2 of the same words)
    if (ct < NL) {
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
      *reinterpret_cast<f32x4*>(&tile0[(16 * pass + ur) * SCR + c0]) = uv;
      *reinterpret_cast<f32x4*>(a.U + (grow0 + 16 * pass + ur) * a.K0 + c0) = uv;
      if (pass == 1) *reinterpret_cast<f32x4*>(&x0s[ur * SCR + c0]) = f32x4{xs[0], xs[1], xs[2], xs[3]};   // x0 for the loss sums
    }
    if (a.intab) {
      // the users' temb rows -> LDS; E rows of the 16 users (wave w: column tile w); b0 + E * W0e^T (pass-0 waves: column tile ct)
      const int qpr = a.TPe >> 2;
      if (tid < 16 * qpr) *reinterpret_cast<f32x4*>(Ts + (tid / qpr) * lde + 4 * (tid % qpr)) = tq;
      for (int f = tid + NTHR; f < 16 * qpr; f += NTHR) {   // (work-groups of fewer than 16 * qpr threads)
        const int r = f / qpr, q = f - r * qpr;
        *reinterpret_cast<f32x4*>(Ts + r * lde + 4 * q) =
            trow[r] >= 0 ? *reinterpret_cast<const f32x4*>(a.tembP + (size_t)trow[r] * a.TPs + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      lds_barrier();   // (also: the staging's tile is complete)
      for (int tl = wave; tl < nke; tl += C::NWAVES) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        for (int u = 0; u < nke; ++u) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(Ts + li * lde + 16 * u + 4 * lq);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(WeS + (tl * 16 + li) * lde + 16 * u + 4 * lq);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
        }
        const float bej = tl == wave ? bev : (tl * 16 + li < a.T ? a.be[tl * 16 + li] : 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) Es[(4 * lq + r) * lde + tl * 16 + li] = tl * 16 + li < a.T ? acc0[r] + acc1[r] + bej : 0.f;
      }
      lds_barrier();
      if (pass == 0 && ct < NW) {
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
        for (int r = 0; r < 4; ++r) B0s[(4 * lq + r) * SCR + col] = acc0[r] + acc1[r] + b0c_;
      }
    }
    lds_barrier();
    if (a.intab && ct < NW) {
#pragma unroll
      for (int r = 0; r < 4; ++r) b0v[r] = B0s[(4 * lq + r) * SCR + col];
    }

    // ---- layer 0: latent part by MFMA, time-embedding part + bias from the table
    if (ct < NW) {
      f32x4 af[NL];
      read_frags<NL, SCR>(tile0 + 16 * pass * SCR, li, lq, af);
      const f32x4 acc = skinny_tile<NL>(af, w0f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = acc[r] + b0v[r];
        a.pre[(grow0 + 16 * pass + 4 * lq + r) * a.WPs + col] = p;
        tile1[(16 * pass + 4 * lq + r) * SCR + col] = prelu_f(p, slope0);
      }
    }
    lds_barrier();
    float* cur = tile1;
    float* oth = tile0;
    for (int h = 1; h <= a.H; ++h) {   // the shared hidden layer, H applications (Q1)
      if (ct < NW) {
        f32x4 af[NW];
        read_frags<NW, SCR>(cur + 16 * pass * SCR, li, lq, af);
        const f32x4 acc = skinny_tile<NW>(af, whf);
        float* ph = a.pre + (size_t)h * a.pre_stride;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = acc[r] + bhv;
          ph[(grow0 + 16 * pass + 4 * lq + r) * a.WPs + col] = p;
          oth[(16 * pass + 4 * lq + r) * SCR + col] = prelu_f(p, slopeh);
        }
      }
      float* t_ = cur; cur = oth; oth = t_;
      lds_barrier();
    }
    // ---- out layer: tanh; Y to memory (the seeds read it) and into the other tile (the loss sums read it)
    if (ct < NL) {
      f32x4 af[NW];
      read_frags<NW, SCR>(cur + 16 * pass * SCR, li, lq, af);
      const f32x4 acc = skinny_tile<NW>(af, wof);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = tanh_fast(acc[r] + bov);
        a.Y[(grow0 + 16 * pass + 4 * lq + r) * a.LPs + col] = y;
        oth[(16 * pass + 4 * lq + r) * SCR + col] = y;
      }
    }
    lds_barrier();
    // ---- loss partial sums (:196-198): R = P - x0, D = (Q - S) / mu^2 - R, over the group's users and the real columns
    {
      double sD = 0, sC = 0, sR = 0, sR2 = 0;
      for (int f = tid; f < SK_USERS * C::LPk; f += NTHR) {
        const int ur = f / C::LPk, c = f - ur * C::LPk;
        const int usr = u0 + ur;
        if (usr < a.B && c < a.L) {
          const float P = oth[ur * SCR + c], S = oth[(16 + ur) * SCR + c], Q = oth[(32 + ur) * SCR + c];
          const float R = P - x0s[ur * SCR + c];
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
        for (int j = 0; j < 4; ++j) red[4 * wave + j] = v4[j];
      }
    }
    lds_barrier();
    if (tid < 4) {
      double s = 0.0;
      for (int w = 0; w < C::NWAVES; ++w) s += red[4 * w + tid];
      a.loss_part[4 * (size_t)g + tid] = s;
    }
    lds_barrier();   // the next group's staging overwrites the tiles and trow
  }
}

// ================================================================================================ backward
// one 16 x 16 tile of a weight gradient: dW[n0 + .][k0 + .] = sum over `rows` stacked rows of D[row][n0 + .] * X[row][k0 + .]
// (A = D^T: lane (li, lq) of MFMA g holds D[4 g + lq][n0 + li]; B: X[4 g + lq][k0 + li]); result in the C layout:
// acc[r] = dW[n0 + 4 lq + r][k0 + li]
template <int ROWS>
__device__ __forceinline__ f32x4 sk_wgrad_tile(const float* __restrict__ D, int ldd, int n0, const float* __restrict__ X, int ldx, int k0,
                                               int li, int lq, f32x4 acc) {
  // every operand is requested before the first MFMA: read - multiply pairs in program order expose the LDS latency per pair
  float dv[ROWS / 4], xv[ROWS / 4];
#pragma unroll
  for (int g = 0; g < ROWS / 4; ++g) { dv[g] = D[(4 * g + lq) * ldd + n0 + li]; xv[g] = X[(4 * g + lq) * ldx + k0 + li]; }
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < ROWS / 4; g += 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[g], xv[g], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[g + 1], xv[g + 1], acc1, 0, 0, 0);
  }
  return acc + acc1;
}

// C-layout tile into a slab: the first contribution of this work-group stores, later ones (further groups of the same work-group)
// add.  Two code paths behind a uniform branch: written as one select the compiler loads the old value on both (a global round
// trip in front of every tile's stores).
__device__ __forceinline__ void sk_slab_tile(float* __restrict__ dst, int ld, int n0, int k0, int li, int lq, const f32x4& acc, bool first) {
  float* p = dst + (size_t)(n0 + 4 * lq) * ld + k0 + li;
  if (first) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(size_t)r * ld] = acc[r];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(size_t)r * ld] += acc[r];
  }
}

template <int NL, int NW>
__global__ __launch_bounds__(64 * 3 * (NL > NW ? NL : NW)) void k_skinny_bwd(const SkStepArgs a) {
  typedef SkCfg<NL, NW> C;
  constexpr int SCR = C::SCR, NTHR = C::NTHR, NV = C::NV, NWAVES = C::NWAVES;
  constexpr int TH = (NW * NW + NWAVES - 1) / NWAVES;   // hidden-layer weight-gradient tiles per wave
  extern __shared__ __attribute__((aligned(16))) float sksh[];
  auto Dt = [&](int i) __attribute__((always_inline)) { return sksh + i * C::TILE; };         // gradient tiles (ping-pong)
  auto Xt = [&](int i) __attribute__((always_inline)) { return sksh + (2 + i) * C::TILE; };   // layer-input tiles (ping-pong)
  float* D3 = sksh + 4 * C::TILE;                            // [16][SCR]: dpre0 summed over the passes
  float* Te = D3 + SK_USERS * SCR;                           // [16][TPs + 4]: temb rows of the group's users
  float* bsum = Te + SK_USERS * (a.TPs + 4);                 // [3 kinds: out, hidden, layer 0][3 passes][64]
  float* red = bsum + 3 * 3 * 64;                            // [32 applications][12 waves]
  double* shd = reinterpret_cast<double*>(red + 32 * 12);    // [16] (8-byte aligned: every piece above is a multiple of 2 floats)
  const int ldte = a.TPs + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pass = wave / NV, ct = wave - pass * NV;
  const int li = lane & 15, lq = lane >> 4;
  const int col = ct * 16 + li;
  const int s = blockIdx.x;   // this work-group's slab set

  f32x4 woT[NL], whT[NW];
  if (ct < NW) {
    load_bfrags<NL>(a.WocT, (size_t)a.LPs, ct, li, lq, woT);
    load_bfrags<NW>(a.WhcT, (size_t)a.WPs, ct, li, lq, whT);
  }
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;

  // ---- the five sums: given (sharded step, after the all-reduce) or folded from the forward's per-group partials in group
  // order (the reduction tree of k_loss_seed: the same bits whichever work-group folds)
  double s0, s1, s2, s3, N;
  if (a.sums) {
    s0 = a.sums[0]; s1 = a.sums[1]; s2 = a.sums[2]; s3 = a.sums[3]; N = a.sums[4];
  } else {
    // The summation tree of k_loss_sums (elementwise.h), so that this fold and that kernel give the same bits: there 256 threads
    // stride the partials, the sums go down each of the four waves by shuffles, and the waves' results are added in order.  Here
    // wave 0 plays the four waves one after the other (a work-group of this kernel may have only three); waves whose partials
    // would all be absent add exact zeros there and are skipped here.
    double* shs = shd;   // [4] totals
    if (wave == 0) {
      double tot[4] = {0, 0, 0, 0};
      double v[4][4];   // [emulated wave][sum]: the first partial of every emulated wave's lane requested before any is used
#pragma unroll
      for (int vw = 0; vw < 4; ++vw)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[vw][j] = 64 * vw + lane < a.NP ? a.loss_part[4 * (size_t)(64 * vw + lane) + j] : 0.0;
#pragma unroll
      for (int vw = 0; vw < 4; ++vw) {
        if (64 * vw >= a.NP) break;
        for (int i = 64 * vw + lane + 256; i < a.NP; i += 256)
          for (int j = 0; j < 4; ++j) v[vw][j] += a.loss_part[4 * (size_t)i + j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[vw][j] += __shfl_down(v[vw][j], off, 64);
        for (int j = 0; j < 4; ++j) tot[j] += v[vw][j];
      }
      if (lane == 0)
        for (int j = 0; j < 4; ++j) shs[j] = tot[j];
    }
    lds_barrier();
    s0 = shs[0]; s1 = shs[1]; s2 = shs[2]; s3 = shs[3]; N = a.count;
  }
  const double A = s0 / N, Cc = s1 / N, Rbar = s2 / N;
  const double V = (N > 1.0) ? (s3 - N * Rbar * Rbar) / (N - 1.0) : __builtin_nan("");
  const double den = 1e-8 + V;
  const double kk = 0.5 / den;
  const float cD = (float)(2.0 * kk / N);
  const float cV = (float)(-(0.5 * (A + Cc) / (den * den)) * 2.0 / (N - 1.0));
  const float rbar = (float)Rbar;
  if (s == 0 && tid == 0 && a.loss) *a.loss = (float)(0.5 * (A + Cc) / den);
  f32x4 accH[TH];
  bool first = true;
  for (int g = blockIdx.x; g < a.G; g += gridDim.x, first = false) {
    const int u0 = SK_USERS * g;
    const size_t grow0 = (size_t)SK_ROWS * g;
    lds_barrier();   // the previous group is done with every tile
    // ---- requests first (a work-group is a chain of dependent memory round trips): the forward's outputs and x0 for the seeds,
    // the pre-activations of the two topmost layers (this lane's accumulator positions), layer 0's input U, the users' timesteps
    constexpr int NSEED = (SK_USERS * C::LPk + NTHR - 1) / NTHR;
    float sP[NSEED], sS[NSEED], sQ[NSEED], sX[NSEED];
#pragma unroll
    for (int i = 0; i < NSEED; ++i) {
      const int f = tid + i * NTHR;
      const int ur = f / C::LPk, c = f - ur * C::LPk;
      const int usr = u0 + ur;
      const bool in = f < SK_USERS * C::LPk && usr < a.B && c < a.L;
      const size_t y = (grow0 + (in ? ur : 0)) * a.LPs + (in ? c : 0);
      sP[i] = in ? a.Y[y] : 0.f; sS[i] = in ? a.Y[y + (size_t)16 * a.LPs] : 0.f; sQ[i] = in ? a.Y[y + (size_t)32 * a.LPs] : 0.f;
      sX[i] = in ? a.x0[(size_t)usr * a.L + c] : 0.f;
    }
    float pv[4] = {0.f, 0.f, 0.f, 0.f}, pv1[4] = {0.f, 0.f, 0.f, 0.f};   // pre[k], pre[k-1] of the chain's current iteration
    if (ct < NW) {
      const float* pk = a.pre + (size_t)a.H * a.pre_stride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t o = (grow0 + 16 * pass + 4 * lq + r) * a.WPs + col;
        pv[r] = pk[o];
        if (a.H >= 1) pv1[r] = (pk - a.pre_stride)[o];
      }
    }
    f32x4 ureg = {0.f, 0.f, 0.f, 0.f};
    if (ct < NL) ureg = *reinterpret_cast<const f32x4*>(a.U + (grow0 + 16 * pass + (lane >> 2)) * a.K0 + ct * 16 + 4 * (lane & 3));
    constexpr int NTE = 4;   // float4 of the users' temb rows per thread: 16 * TPs / 4 <= NTE * NTHR while TPs <= 192 * NV (else a loop)
    float4 tereg[NTE];
    {
      const int tq = a.TPs / 4;
#pragma unroll
      for (int i = 0; i < NTE; ++i) {
        const int f = tid + i * NTHR;
        const int ur = f / tq, q = f - ur * tq;
        const bool in = f < SK_USERS * tq && u0 + ur < a.B;
        tereg[i] = in ? *reinterpret_cast<const float4*>(a.tembP + (size_t)a.tdev[u0 + ur] * a.TPs + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // ---- seeds (App. A.5) times tanh' into Dt(0); zero the bias accumulators; act[H] into Xt(0)
#pragma unroll
    for (int i = 0; i < NSEED; ++i) {
      const int f = tid + i * NTHR;
      if (f < SK_USERS * C::LPk) {
        const int ur = f / C::LPk, c = f - ur * C::LPk;
        float gP = 0.f, gS = 0.f, gQ = 0.f;
        if (u0 + ur < a.B && c < a.L) {
          const float P = sP[i], S = sS[i], Q = sQ[i];
          const float R = P - sX[i];
          const float D = (Q - S) / MU2 - R;
          const float gD = cD * D;
          const float gC = cD * (R - S);
          const float gV = cV * (R - rbar);
          gP = (-gD + gC + gV) * (1.f - P * P);
          gQ = (gD / MU2) * (1.f - Q * Q);
          gS = (-gD / MU2 - gC) * (1.f - S * S);
        }
        Dt(0)[ur * SCR + c] = gP; Dt(0)[(16 + ur) * SCR + c] = gS; Dt(0)[(32 + ur) * SCR + c] = gQ;
      }
    }
    for (int f = tid; f < 3 * 3 * 64; f += NTHR) bsum[f] = 0.f;
    if (ct < NW) {
      const float sl = a.H > 0 ? slopeh : slope0;
#pragma unroll
      for (int r = 0; r < 4; ++r) Xt(0)[(16 * pass + 4 * lq + r) * SCR + col] = prelu_f(pv[r], sl);
    }
    {
      const int tq = a.TPs / 4;
#pragma unroll
      for (int i = 0; i < NTE; ++i) {
        const int f = tid + i * NTHR;
        if (f < SK_USERS * tq) *reinterpret_cast<float4*>(Te + (f / tq) * ldte + 4 * (f % tq)) = tereg[i];
      }
      for (int f = tid + NTE * NTHR; f < SK_USERS * tq; f += NTHR) {   // very long embeddings only
        const int ur = f / tq, q = f - ur * tq;
        *reinterpret_cast<float4*>(Te + ur * ldte + 4 * q) =
            u0 + ur < a.B ? *reinterpret_cast<const float4*>(a.tembP + (size_t)a.tdev[u0 + ur] * a.TPs + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    lds_barrier();
    // ---- out layer: bias gradient = column sums of the seeds; weight gradient dWo[n < L][k < W] = seeds^T * act[H]
    for (int f = tid; f < 3 * C::LPk; f += NTHR) {
      const int p_ = f / C::LPk, c = f - p_ * C::LPk;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += Dt(0)[(16 * p_ + r) * SCR + c];
      bsum[(0 * 3 + p_) * 64 + c] = t;
    }
    for (int tl = wave; tl < NL * NW; tl += NWAVES) {
      const int nt = tl / NW, kt = tl - nt * NW;
      const f32x4 acc = sk_wgrad_tile<SK_ROWS>(Dt(0), SCR, 16 * nt, Xt(0), SCR, 16 * kt, li, lq, f32x4{0.f, 0.f, 0.f, 0.f});
      sk_slab_tile(a.slabO + (size_t)s * a.LPs * a.WPs, a.WPs, 16 * nt, 16 * kt, li, lq, acc, first);
    }
#pragma unroll
    for (int i = 0; i < TH; ++i) accH[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- the chain: iteration k produces dpre[k] (gradient of pre-activation k) from Dt(cur), then the weight gradient of the
    // layer whose OUTPUT is pre[k] (k >= 1: the shared hidden layer, input act[k-1]; k == 0: layer 0, input U)
    int cur = 0, xc = 0;
    float ss_h = 0.f, ss_0 = 0.f, cs_h = 0.f, cs_0 = 0.f;
    // one iteration; (P0, P1, P2) = (pre[k], pre[k-1], pre[k-2]): P0 is this layer's PReLU', P1 the next layer input, P2 is requested
    // here and first used a whole iteration later.  The three register sets rotate through the roles (three-way unrolled loop below):
    // a register copy at the end of the iteration would wait for P2's loads right here
    auto chain_iter = [&](int k, float (&pv)[4], float (&pv1)[4], float (&pv2)[4]) __attribute__((always_inline)) {
      if (ct < NW) {
        // two layers ahead: pre[k-2] is requested now, used as the layer input of the next iteration (loads never sit on the chain)
        if (k >= 2) {
          const float* pk = a.pre + (size_t)(k - 2) * a.pre_stride;
#pragma unroll
          for (int r = 0; r < 4; ++r) pv2[r] = pk[(grow0 + 16 * pass + 4 * lq + r) * a.WPs + col];
        }
        f32x4 acc;
        if (k == a.H) {
          f32x4 af[NL];
          read_frags<NL, SCR>(Dt(cur) + 16 * pass * SCR, li, lq, af);
          acc = skinny_tile<NL>(af, woT);
        } else {
          f32x4 af[NW];
          read_frags<NW, SCR>(Dt(cur) + 16 * pass * SCR, li, lq, af);
          acc = skinny_tile<NW>(af, whT);
        }
        const float sl = k > 0 ? slopeh : slope0;
        // slope gradient (sum of v * min(pre, 0)) and bias gradient (column sums of d): per-lane sums here - the shared hidden
        // layer's over all its applications - reduced through the wave ONCE behind the chain (six dependent cross-lane steps per
        // iteration were a third of it)
        float ssum = 0.f, csum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = pv[r], v = acc[r];
          const bool pos = p > 0.f;
          const float d = pos ? v : sl * v;
          ssum += pos ? 0.f : v * p;
          csum += d;
          Dt(cur ^ 1)[(16 * pass + 4 * lq + r) * SCR + col] = d;
        }
        if (k > 0) { ss_h += ssum; cs_h += csum; } else { ss_0 = ssum; cs_0 = csum; }
        if (k >= 1) {   // the next layer input: act[k-1] = prelu(pre[k-1])
          const float sln = k - 1 > 0 ? slopeh : slope0;
#pragma unroll
          for (int r = 0; r < 4; ++r) Xt(xc ^ 1)[(16 * pass + 4 * lq + r) * SCR + col] = prelu_f(pv1[r], sln);
        }
      }
      if (k == 0 && ct < NL)   // layer 0's input: the dropped-out latents the forward stored (requested at the top)
        *reinterpret_cast<f32x4*>(&Xt(xc ^ 1)[(16 * pass + (lane >> 2)) * SCR + ct * 16 + 4 * (lane & 3)]) = ureg;
      lds_barrier();
      if (k >= 1) {
        for (int i = 0; i < TH; ++i) {
          const int tl = wave + i * NWAVES;
          if (tl < NW * NW) {
            const int nt = tl / NW, kt = tl - nt * NW;
            accH[i] = sk_wgrad_tile<SK_ROWS>(Dt(cur ^ 1), SCR, 16 * nt, Xt(xc ^ 1), SCR, 16 * kt, li, lq, accH[i]);
          }
        }
      } else {
        for (int tl = wave; tl < NW * NL; tl += NWAVES) {
          const int nt = tl / NL, kt = tl - nt * NL;
          const f32x4 acc = sk_wgrad_tile<SK_ROWS>(Dt(cur ^ 1), SCR, 16 * nt, Xt(xc ^ 1), SCR, 16 * kt, li, lq, f32x4{0.f, 0.f, 0.f, 0.f});
          sk_slab_tile(a.slab0 + (size_t)s * a.WPs * a.K0, a.K0, 16 * nt, 16 * kt, li, lq, acc, first);
        }
        // dpre0 summed over the three passes (they share the user's timestep): the operand of M = dpre0^T * temb (tail.h)
        for (int f = tid; f < SK_USERS * C::WPk; f += NTHR) {
          const int ur = f / C::WPk, c = f - ur * C::WPk;
          D3[ur * SCR + c] = (Dt(cur ^ 1)[ur * SCR + c] + Dt(cur ^ 1)[(16 + ur) * SCR + c]) + Dt(cur ^ 1)[(32 + ur) * SCR + c];
        }
      }
      cur ^= 1; xc ^= 1;
    };
    {
      float pvc[4] = {0.f, 0.f, 0.f, 0.f};
      int k = a.H;
      while (true) {
        chain_iter(k, pv, pv1, pvc); if (--k < 0) break;
        chain_iter(k, pv1, pvc, pv); if (--k < 0) break;
        chain_iter(k, pvc, pv, pv1); if (--k < 0) break;
      }
    }
    // the chain's per-lane slope / bias sums, through the wave: bias: the four lane groups of a column (four rows each); slope:
    // the whole wave; red: [0] layer 0, [1] the shared hidden layer (all applications), per wave
    {
      float c0_ = cs_0, ch_ = cs_h, s0_ = ss_0, sh_ = ss_h;
      c0_ += __shfl_xor(c0_, 16, 64); ch_ += __shfl_xor(ch_, 16, 64);
      c0_ += __shfl_xor(c0_, 32, 64); ch_ += __shfl_xor(ch_, 32, 64);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { s0_ += __shfl_down(s0_, off, 64); sh_ += __shfl_down(sh_, off, 64); }
      if (ct < NW && lq == 0) { bsum[(2 * 3 + pass) * 64 + col] = c0_; bsum[(1 * 3 + pass) * 64 + col] = ch_; }
      if (lane == 0) { red[0 * 12 + wave] = ct < NW ? s0_ : 0.f; red[1 * 12 + wave] = ct < NW ? sh_ : 0.f; }
    }
    lds_barrier();
    // ---- M[n < W][i < T] = sum over the group's users of D3[u][n] * temb[t_u][i], into the trailing columns of the layer-0 slab
    {
      const int TT = a.TPs / 16;
      for (int tl = wave; tl < NW * TT; tl += NWAVES) {
        const int nt = tl / TT, it = tl - nt * TT;
        const f32x4 acc = sk_wgrad_tile<SK_USERS>(D3, SCR, 16 * nt, Te, ldte, 16 * it, li, lq, f32x4{0.f, 0.f, 0.f, 0.f});
        sk_slab_tile(a.slab0 + (size_t)s * a.WPs * a.K0, a.K0, 16 * nt, a.LPs + 16 * it, li, lq, acc, first);
      }
    }
    // ---- the shared hidden layer's weight gradient (all H applications), bias gradients, slope partials
    if (a.H >= 1) {
      for (int i = 0; i < TH; ++i) {
        const int tl = wave + i * NWAVES;
        if (tl < NW * NW) sk_slab_tile(a.slabH + (size_t)s * a.WPs * a.WPs, a.WPs, 16 * (tl / NW), 16 * (tl % NW), li, lq, accH[i], first);
      }
    }
    for (int f = tid; f < 3 * 64; f += NTHR) {
      const int kind = f / 64, c = f - kind * 64;
      const float t = (bsum[(kind * 3 + 0) * 64 + c] + bsum[(kind * 3 + 1) * 64 + c]) + bsum[(kind * 3 + 2) * 64 + c];
      float* dst = kind == 0 ? (c < a.LPs ? a.dbOs + (size_t)s * a.LPs + c : nullptr)
                             : (c < a.WPs ? (kind == 1 ? a.dbHs : a.db0s) + (size_t)s * a.WPs + c : nullptr);
      if (dst && (kind != 1 || a.H >= 1)) *dst = first ? t : *dst + t;
    }
    if (tid <= a.H) {   // application slots of alpha_part: [0] layer 0, [1] the hidden layer's total, [2 ..] nothing
      float sum = 0.f;
      if (tid < 2)
        for (int w = 0; w < NWAVES; ++w) sum += red[tid * 12 + w];
      float* dst = a.alpha_part + (size_t)tid * a.alpha_part_stride + s;
      if (first) *dst = sum; else *dst += sum;
    }
  }
}

}  // namespace sdrm
