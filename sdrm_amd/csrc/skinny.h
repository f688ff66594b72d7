// Persistent reverse-sampling kernel for narrow eps-nets (padded widths LP, WP <= 64; e.g. the ADM/NeuMF
// configuration L = W = 40, H = 5, T = 93 of BASELINE.json).
//
// All weights of such a net (3 * 64*64 floats) fit in one CU's LDS, and the rows of a sampling call are
// independent for the WHOLE reverse loop (sample_ddpm, train_SDRM.py:37-59), so one launch runs every
// timestep: no per-step kernel boundaries, no HBM round trips of activations, no grid-wide barrier.
//
//   work-group = 4 waves, weights staged once into LDS, then NO further block barriers;
//   each wave owns 16 rows (slots) for all steps i = Tmax..1 of those rows:
//     x (state) lives in registers in the MFMA C layout,
//     every layer = v_mfma_f32_16x16x4_f32 tiles with A = activations (registers), B = weights (LDS, read as
//     ds_read_b128: lane group q supplies k = 16u + 4q + e for MFMA (u,e); A uses the same k permutation),
//     C layout -> A layout of the next layer through a per-wave LDS scratch tile (same-wave LDS ops are
//     ordered, so no barrier),
//     epilogue of the last layer = tanh, DDPM reverse update (denoise_add_noise, :20-25) and the next step's
//     input dropout (F.dropout, :100), with z / keep bits from Philox (one call per column pair, shared by the
//     two lanes of the pair) or from caller arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.h"
#include "philox.h"

namespace sdrm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
  // weights (padded compute copies) and tables
  const float* W0c; int K0;        // [WP][K0], first LP columns used
  const float* Whc;                // [WP][WP]
  const float* Woc;                // [LP][WP]
  const float* bh; const float* bo;
  const float* B0tab;              // [T+1][WP] = b0 + C0[t]
  const float* slope0; const float* slopeh;
  const float* rev;                // [3][T+1]: c1, sqrt(alpha), sqrt(beta)
  // call
  const float* xT; const float* Z; const uint8_t* keep;   // EXPLICIT: [n,L], [T+1,n,L], [T+1,n,L]
  const int64_t* Tj; const int* rowid;                     // per slot (null = full resolution / identity)
  float* out;                                              // [n,L] original row order
  int n, L, W, T, H;
  int mode; uint32_t seed_lo, seed_hi, call_id; int64_t row0; float nd;
};

// One layer on a wave's 16 rows: OUT tiles (16 cols each) = A[16 x 16*NK] * Wl^T, two column tiles at a time
// (two independent accumulators cover the 40-cycle dependent latency of the 32-cycle MFMA).
template <int NK, int NOUT, class Epi>
__device__ __forceinline__ void skinny_layer(const f32x4 (&a)[NK], const float* __restrict__ Wl, int ldw, int li, int lq,
                                             Epi epi) {
#pragma unroll
  for (int ct = 0; ct < NOUT; ct += 2) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* w0 = Wl + (ct * 16 + li) * ldw + 4 * lq;
    const float* w1 = w0 + 16 * ldw;
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(w0 + 16 * u);
      f32x4 b1 = {0.f, 0.f, 0.f, 0.f};
      if (ct + 1 < NOUT) b1 = *reinterpret_cast<const f32x4*>(w1 + 16 * u);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b0[e], acc0, 0, 0, 0);
        if (ct + 1 < NOUT) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b1[e], acc1, 0, 0, 0);
      }
    }
    epi(ct, acc0);
    if (ct + 1 < NOUT) epi(ct + 1, acc1);
  }
}

// scratch tile [16][SCR] (C layout in, A fragments out)
template <int NK, int SCR>
__device__ __forceinline__ void read_frags(const float* __restrict__ scr, int li, int lq, f32x4 (&a)[NK]) {
#pragma unroll
  for (int u = 0; u < NK; ++u) a[u] = *reinterpret_cast<const f32x4*>(scr + li * SCR + 16 * u + 4 * lq);
}

template <int NL, int NW>   // LP = 16*NL, WP = 16*NW
__global__ __launch_bounds__(256) void k_skinny_sample(const SkinnyArgs a) {
  constexpr int LP = 16 * NL, WP = 16 * NW;
  constexpr int LD0 = LP + 4, LDH = WP + 4;              // weight row strides in LDS (floats; 16-byte aligned rows)
  constexpr int SCR = (LP > WP ? LP : WP) + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W0s = lds;                 // [WP][LD0]
  float* Whs = W0s + WP * LD0;      // [WP][LDH]
  float* Wos = Whs + WP * LDH;      // [LP][LDH]
  float* scr_all = Wos + LP * LDH;  // 4 x [16][SCR]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int f = tid; f < WP * LP; f += 256) { const int j = f / LP, k = f - j * LP; W0s[j * LD0 + k] = a.W0c[(size_t)j * a.K0 + k]; }
  for (int f = tid; f < WP * WP; f += 256) { const int j = f / WP, k = f - j * WP; Whs[j * LDH + k] = a.Whc[(size_t)j * WP + k]; }
  for (int f = tid; f < LP * WP; f += 256) { const int j = f / WP, k = f - j * WP; Wos[j * LDH + k] = a.Woc[(size_t)j * WP + k]; }
  __syncthreads();   // the only block-wide barrier: from here on the four waves never meet again

  float* scr = scr_all + wave * 16 * SCR;
  const int li = lane & 15, lq = lane >> 4;
  const int s0 = (blockIdx.x * 4 + wave) * 16;
  if (s0 >= a.n) return;
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;
  const int npair = lane & 1;   // which half of the 4 rows this lane draws Philox for (shared with lane^1)

  // rows of this lane in the C layout: slot s0 + 4*lq + r, r = 0..3; column of tile ct: ct*16 + li
  int row[4], tj[4];
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = s0 + 4 * lq + r;
    ok[r] = s < a.n;
    row[r] = ok[r] ? (a.rowid ? a.rowid[s] : s) : 0;
    tj[r] = ok[r] ? (a.Tj ? (int)a.Tj[s] : a.T) : 0;
  }
  int ihi = max(max(tj[0], tj[1]), max(tj[2], tj[3]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ihi = max(ihi, __shfl_xor(ihi, off, 64));
  ihi = __builtin_amdgcn_readfirstlane(ihi);

  // randoms of "sub-step" sub for this lane's 4 rows of column tile ct: normals zz[r] (already * nd) and the
  // keep bits that ride on the same Philox call (they belong to step sub-1).  Lanes l and l^1 hold the two
  // columns of a pair: the even one draws rows 0,1, the odd one rows 2,3, and they swap halves.
  auto draws = [&](int ct, int sub, float (&zz)[4], bool (&kp)[4]) {
    const int col = ct * 16 + li;
    if (ct * 16 >= a.L) {   // tile of pure padding columns (wave-uniform): nothing to draw
#pragma unroll
      for (int r = 0; r < 4; ++r) { zz[r] = 0.f; kp[r] = false; }
      return;
    }
    if (a.mode == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool v = ok[r] && col < a.L;
        const size_t idx = (size_t)row[r] * a.L + col;
        zz[r] = (v && sub >= 2 && sub <= a.T) ? a.Z[(size_t)sub * a.n * a.L + idx] : 0.f;
        kp[r] = (v && sub >= 2) ? a.keep[(size_t)(sub - 1) * a.n * a.L + idx] != 0 : false;
      }
      return;
    }
    float mine[2], theirs[2];
    uint32_t mb = 0, tb = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = npair ? row[2 + j] : row[j];   // static indices + select: runtime-indexed arrays would go to scratch
      const U4 w = philox4x32_10((uint32_t)(a.row0 + rr), (uint32_t)(col >> 1), PURPOSE_SAMPLE_STEP | ((uint32_t)sub << 8),
                                 a.call_id, a.seed_lo, a.seed_hi);
      float n0, n1;
      box_muller(w.x, w.y, n0, n1);
      mine[j] = npair ? n1 : n0;
      theirs[j] = npair ? n0 : n1;
      mb |= ((w.z >> (npair * 8)) & 1u) << j;
      tb |= ((w.z >> ((npair ^ 1) * 8)) & 1u) << j;
    }
    const uint32_t gb = (uint32_t)__shfl_xor((int)tb, 1, 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float got = __shfl_xor(theirs[j], 1, 64);   // partner's draw for MY column, its row half
      const bool mk = (mb >> j) & 1u, gk = (gb >> j) & 1u;
      zz[j] = (npair ? got : mine[j]) * a.nd;       kp[j] = npair ? gk : mk;        // rows 0,1: drawn by the even lane
      zz[2 + j] = (npair ? mine[j] : got) * a.nd;   kp[2 + j] = npair ? mk : gk;    // rows 2,3: drawn by the odd lane
    }
  };

  // x_T and the dropout of every row's own first step
  float x[NL][4];
#pragma unroll
  for (int ct = 0; ct < NL; ++ct) {
    const int col = ct * 16 + li;
    if (a.mode == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[ct][r] = (ok[r] && col < a.L) ? a.xT[(size_t)row[r] * a.L + col] : 0.f;
    } else {
      float mine[2], theirs[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rr = npair ? row[2 + j] : row[j];
        const U4 w = philox4x32_10((uint32_t)(a.row0 + rr), (uint32_t)(col >> 1), PURPOSE_SAMPLE_XT, a.call_id, a.seed_lo,
                                   a.seed_hi);
        float n0, n1;
        box_muller(w.x, w.y, n0, n1);
        mine[j] = npair ? n1 : n0;
        theirs[j] = npair ? n0 : n1;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float got = __shfl_xor(theirs[j], 1, 64);
        x[ct][j] = npair ? got : mine[j];
        x[ct][2 + j] = npair ? mine[j] : got;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!(ok[r] && col < a.L)) x[ct][r] = 0.f;
    }
  }
  // stage dropout_{ihi}(x) (C layout) -> scratch; rows that start later get their own mask when they start
  auto stage_input = [&](int step) {
#pragma unroll
    for (int ct = 0; ct < NL; ++ct) {
      float zz[4];
      bool kp[4];
      draws(ct, step + 1, zz, kp);
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[(4 * lq + r) * SCR + ct * 16 + li] = kp[r] ? 2.f * x[ct][r] : 0.f;
    }
  };
  stage_input(ihi);

  f32x4 a0[NL], ah[NW];
  for (int i = ihi; i >= 1; --i) {
    read_frags<NL, SCR>(scr, li, lq, a0);
    const float* b0row = a.B0tab + (size_t)i * WP;
    // layer 0 (+ the step's embedding term folded into the bias table), PReLU, to scratch
    skinny_layer<NL, NW>(a0, W0s, LD0, li, lq, [&](int ct, const f32x4& acc) {
      const float b = b0row[ct * 16 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[(4 * lq + r) * SCR + ct * 16 + li] = prelu_f(acc[r] + b, slope0);
    });
    for (int h = 0; h < a.H; ++h) {   // the shared hidden layer, H applications (Q1)
      read_frags<NW, SCR>(scr, li, lq, ah);
      skinny_layer<NW, NW>(ah, Whs, LDH, li, lq, [&](int ct, const f32x4& acc) {
        const float b = a.bh[ct * 16 + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[(4 * lq + r) * SCR + ct * 16 + li] = prelu_f(acc[r] + b, slopeh);
      });
    }
    read_frags<NW, SCR>(scr, li, lq, ah);
    const float c1 = a.rev[i], sa = a.rev[(a.T + 1) + i], sb = a.rev[2 * (a.T + 1) + i];
    skinny_layer<NW, NL>(ah, Wos, LDH, li, lq, [&](int ct, const f32x4& acc) {
      const float b = a.bo[ct * 16 + li];
      float zz[4];
      bool kp[4];
      draws(ct, i, zz, kp);     // z_i and the keep bits of step i-1
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float eps_hat = tanh_fast(acc[r] + b);
        const bool active = tj[r] >= i;
        const float z = (i > 1 && ct * 16 + li < a.L) ? zz[r] : 0.f;   // padding columns stay exactly zero
        const float xn = active ? (x[ct][r] - eps_hat * c1) / sa + sb * z : x[ct][r];
        x[ct][r] = xn;
        scr[(4 * lq + r) * SCR + ct * 16 + li] = (i > 1 && kp[r]) ? 2.f * xn : 0.f;
      }
    });
  }
#pragma unroll
  for (int ct = 0; ct < NL; ++ct) {
    const int col = ct * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r] && col < a.L) a.out[(size_t)row[r] * a.L + col] = x[ct][r];
  }
}

}  // namespace sdrm
