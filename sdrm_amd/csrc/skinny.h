// Persistent reverse-sampling kernel for narrow eps-nets (padded widths LP, WP <= 64; e.g. the ADM/NeuMF
// configuration L = W = 40, H = 5, T = 93 of BASELINE.json).
//
// All weights of such a net (3 * 64*64 floats) fit in one CU's LDS, and the rows of a sampling call are
// independent for the WHOLE reverse loop (sample_ddpm, train_SDRM.py:37-59), so one launch runs every
// timestep: no per-step kernel boundaries, no HBM round trips of activations, no grid-wide barrier.
//
//   work-group = 16 rows (slots) for all steps i = Tmax..1 of those rows; wave w owns the 16-column tile w of every
//   layer's output (a row's H+2 layers are a serial chain: splitting the columns over the waves makes every link
//   16 MFMAs long instead of 64):
//     x (state) lives in registers in the MFMA C layout, the wave's tile of it,
//     every layer = v_mfma_f32_16x16x4_f32 with A = the 16 x 64 activation tile (LDS, ds_read_b128: lane group q
//     supplies k = 16u + 4q + e for MFMA (u,e)), B = the wave's slice of the weights, held in registers for the whole
//     loop (same k permutation),
//     C layout -> A layout of the next layer through two alternating LDS tiles, one barrier per layer,
//     epilogue of the last layer = tanh, DDPM reverse update (denoise_add_noise, :20-25) and the next step's
//     input dropout (F.dropout, :100), with z / keep bits from Philox (one call per row and column quad, dealt to
//     all lanes of the work-group at the top of the step and passed through LDS) or from caller arrays.
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
  int LPs, WPs;                    // padded widths = row strides of the weight copies and tables (multiples of 32)
  int mode; uint32_t seed_lo, seed_hi, call_id; int64_t row0; float nd;
};

// scratch tile [16][SCR] (C layout in, A fragments out)
template <int NK, int SCR>
__device__ __forceinline__ void read_frags(const float* __restrict__ scr, int li, int lq, f32x4 (&a)[NK]) {
#pragma unroll
  for (int u = 0; u < NK; ++u) a[u] = *reinterpret_cast<const f32x4*>(scr + li * SCR + 16 * u + 4 * lq);
}

// One output-column tile of a layer: acc[16 x 16] = A[16 x 16*NK] * B, both operands in registers.
template <int NK>
__device__ __forceinline__ f32x4 skinny_tile(const f32x4 (&a)[NK], const f32x4 (&b)[NK]) {
  // two accumulators (even / odd k-steps): a 16x16x4 MFMA issues in 32 cycles but its result is ready later
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NK; ++u) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u][1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[u][2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[u][3], acc1, 0, 0, 0);
  }
  return acc0 + acc1;
}

// B fragments of output-column tile `tile` of a row-major [out][ld] matrix: lane (li, lq) holds k = 16u + 4lq + e
template <int NK>
__device__ __forceinline__ void load_bfrags(const float* __restrict__ Wm, size_t ld, int tile, int li, int lq, f32x4 (&b)[NK]) {
  const float* p = Wm + (size_t)(tile * 16 + li) * ld + 4 * lq;
#pragma unroll
  for (int u = 0; u < NK; ++u) b[u] = *reinterpret_cast<const f32x4*>(p + 16 * u);
}

// NL, NW: 16-column tiles that hold real columns (ceil(L/16), ceil(W/16)); the padding tiles beyond them (the engine
// pads widths to 32) are all-zero in weights and activations and are skipped.
template <int NL, int NW>
__global__ __launch_bounds__(64 * (NL > NW ? NL : NW)) void k_skinny_sample(const SkinnyArgs a) {
  constexpr int LP = 16 * NL, WP = 16 * NW, NV = NL > NW ? NL : NW;
  constexpr int SCR = (LP > WP ? LP : WP) + 4;
  __shared__ __attribute__((aligned(16))) float tile[2][16 * SCR];
  __shared__ __attribute__((aligned(16))) float zbuf[16 * LP];      // the step's normals, [row][col]
  __shared__ __attribute__((aligned(4))) uint8_t kbuf[16 * LP];    // keep bits of the step below
  __shared__ int rid[16];              // global row id of each slot (Philox counter)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int s0 = blockIdx.x * 16;
  const int col = wave * 16 + li;
  const bool lat = wave < NL, hid = wave < NW;   // this wave owns a latent / a hidden column tile
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;

  // this wave's slices of the three weight matrices, as MFMA B fragments, for the whole reverse loop
  f32x4 w0f[NL], whf[NW], wof[NW];
  float bhv = 0.f, bov = 0.f;
  if (hid) {
    load_bfrags<NL>(a.W0c, (size_t)a.K0, wave, li, lq, w0f);
    load_bfrags<NW>(a.Whc, (size_t)a.WPs, wave, li, lq, whf);
    bhv = a.bh[col];
  }
  if (lat) {
    load_bfrags<NW>(a.Woc, (size_t)a.WPs, wave, li, lq, wof);
    bov = a.bo[col];
  }

  // rows of this lane in the C layout: slot s0 + 4*lq + r, r = 0..3
  int row[4], tj[4];
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = s0 + 4 * lq + r;
    ok[r] = s < a.n;
    row[r] = ok[r] ? (a.rowid ? a.rowid[s] : s) : 0;
    tj[r] = ok[r] ? (a.Tj ? (int)a.Tj[s] : a.T) : 0;
  }
  int ihi = max(max(tj[0], tj[1]), max(tj[2], tj[3]));   // every wave sees all 16 rows: the same value in each
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ihi = max(ihi, __shfl_xor(ihi, off, 64));
  ihi = __builtin_amdgcn_readfirstlane(ihi);

  // PHILOX randoms of a step: one call per (row, column quad) gives the quad's four normals and the keep bits of the
  // step below.  The calls of a 16-row block are dealt to ALL lanes of the work-group (a call is ~1000 cycles of
  // integer multiplies: left to the waves that own latent tiles it doubled the step), results go through LDS in
  // the [row][col] layout the epilogue reads.  Purpose word: x_T, or (step << 8 | SAMPLE_STEP).
  const int P = (a.L + 3) >> 2, ncall = 16 * P, nthr = 64 * NV;
  auto produce = [&](uint32_t purpose) {
    for (int f0 = 0; f0 < ncall; f0 += nthr) {
      // the last, partial round goes to the highest lanes (the wave without a latent tile, if there is one)
      const int left = ncall - f0;
      const int f = left >= nthr ? f0 + tid : f0 + tid - (nthr - left);
      if (f >= f0) {
        const int r = f / P, pr = f - r * P;
        const U4 w = philox4x32_10((uint32_t)(a.row0 + rid[r]), (uint32_t)pr, purpose, a.call_id, a.seed_lo, a.seed_hi);
        f32x4 nn;
        float n0, n1;
        box_muller(w.x, w.y, n0, n1); nn[0] = n0; nn[1] = n1;
        box_muller(w.z, w.w, n0, n1); nn[2] = n0; nn[3] = n1;
        *reinterpret_cast<f32x4*>(&zbuf[r * LP + 4 * pr]) = nn;
        *reinterpret_cast<uint32_t*>(&kbuf[r * LP + 4 * pr]) = (w.x & 1u) | ((w.y & 1u) << 8) | ((w.z & 1u) << 16) | ((w.w & 1u) << 24);
      }
    }
  };
  // this lane's 4 rows of the wave's column tile: normals zz[r] (already * nd) of sub-step sub and the keep bits of
  // step sub-1, from caller arrays (EXPLICIT) or from what produce() left in LDS
  auto draws = [&](int sub, float (&zz)[4], bool (&kp)[4]) {
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool v = col < a.L;   // padding columns are never produced
      zz[r] = v ? zbuf[(4 * lq + r) * LP + col] * a.nd : 0.f;
      kp[r] = v ? kbuf[(4 * lq + r) * LP + col] != 0 : false;
    }
  };

  // x_T of the wave's latent tile (registers, C layout) and the dropout of step ihi -> LDS tile 0
  if (tid < 16) rid[tid] = (s0 + tid < a.n) ? (a.rowid ? a.rowid[s0 + tid] : s0 + tid) : 0;
  __syncthreads();
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.mode != 0) produce(PURPOSE_SAMPLE_XT);
  __syncthreads();
  if (lat) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ok[r] && col < a.L) x[r] = a.mode == 0 ? a.xT[(size_t)row[r] * a.L + col] : zbuf[(4 * lq + r) * LP + col];
    }
  }
  __syncthreads();
  if (a.mode != 0) produce(PURPOSE_SAMPLE_STEP | ((uint32_t)(ihi + 1) << 8));
  __syncthreads();
  if (lat) {
    float zz[4];
    bool kp[4];
    draws(ihi + 1, zz, kp);   // rows that start later get their own mask when they start
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[0][(4 * lq + r) * SCR + col] = kp[r] ? 2.f * x[r] : 0.f;
  }
  __syncthreads();

  // per-step table values are fetched one step ahead (a load issued where it is used is an exposed L2 round trip)
  float b0n = hid ? a.B0tab[(size_t)ihi * a.WPs + col] : 0.f;
  float c1n = a.rev[ihi], san = a.rev[(a.T + 1) + ihi], sbn = a.rev[2 * (a.T + 1) + ihi];
  int cur = 0;
  for (int i = ihi; i >= 1; --i) {
    const float b0 = b0n, c1 = c1n, sa = san, sb = sbn;
    if (i > 1) {
      if (hid) b0n = a.B0tab[(size_t)(i - 1) * a.WPs + col];
      c1n = a.rev[i - 1]; san = a.rev[(a.T + 1) + i - 1]; sbn = a.rev[2 * (a.T + 1) + i - 1];
    }
    if (a.mode != 0 && i > 1) produce(PURPOSE_SAMPLE_STEP | ((uint32_t)i << 8));   // read H+1 barriers further down
    // layer 0 (+ the step's embedding term folded into the bias table), PReLU
    if (hid) {
      f32x4 af[NL];
      read_frags<NL, SCR>(tile[cur], li, lq, af);
      const f32x4 acc = skinny_tile<NL>(af, w0f);
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[cur ^ 1][(4 * lq + r) * SCR + col] = prelu_f(acc[r] + b0, slope0);
    }
    cur ^= 1;
    __syncthreads();
    for (int h = 0; h < a.H; ++h) {   // the shared hidden layer, H applications (Q1)
      if (hid) {
        f32x4 af[NW];
        read_frags<NW, SCR>(tile[cur], li, lq, af);
        const f32x4 acc = skinny_tile<NW>(af, whf);
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[cur ^ 1][(4 * lq + r) * SCR + col] = prelu_f(acc[r] + bhv, slopeh);
      }
      cur ^= 1;
      __syncthreads();
    }
    // out layer, tanh, DDPM reverse update, the next step's input dropout
    if (lat) {
      float zz[4];
      bool kp[4];
      draws(i, zz, kp);     // z_i and the keep bits of step i-1
      f32x4 af[NW];
      read_frags<NW, SCR>(tile[cur], li, lq, af);
      const f32x4 acc = skinny_tile<NW>(af, wof);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float eps_hat = tanh_fast(acc[r] + bov);
        const bool active = tj[r] >= i;
        const float z = (i > 1 && col < a.L) ? zz[r] : 0.f;   // padding columns stay exactly zero
        const float xn = active ? (x[r] - eps_hat * c1) / sa + sb * z : x[r];
        x[r] = xn;
        tile[cur ^ 1][(4 * lq + r) * SCR + col] = (i > 1 && kp[r]) ? 2.f * xn : 0.f;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  if (lat) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r] && col < a.L) a.out[(size_t)row[r] * a.L + col] = x[r];
  }
}

}  // namespace sdrm
