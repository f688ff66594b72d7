// Elementwise / reduction / optimiser kernels of the SDRM denoising path (gfx950).
// Reference lines are into /root/reference/train_SDRM.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.h"

namespace sdrm {

// Stacked row order of a train step's three passes P, S, Q (train_SDRM.py:331-333).  BLOCKED: pass * B + user, zero rows
// behind 3 B.  GROUPED (the row-owned forward, rowchain.h): a work-group's 96 rows are the P | S | Q rows of 32 users,
// row = 96 * (user / 32) + 32 * pass + user % 32; user slots beyond the batch inside the last group, and the rows behind the
// last group, are padding.
constexpr int RC_USERS = 32;            // users per group
constexpr int RC_ROWS = 3 * RC_USERS;   // stacked rows per group
__host__ __device__ inline int rc_row(int pass, int user) { return RC_ROWS * (user / RC_USERS) + RC_USERS * pass + (user % RC_USERS); }
// grouped: 0 BLOCKED, 1 GROUPED by 32 users (rowchain.h), 2 grouped by 16 users (the narrow nets' work-groups, skinny_step.h:
// row = 48 * (user / 16) + 16 * pass + user % 16)
__host__ __device__ inline size_t stacked_row(int grouped, int pass, int user, int B) {
  if (grouped == 2) return (size_t)(48 * (user / 16) + 16 * pass + (user % 16));
  return grouped ? (size_t)rc_row(pass, user) : (size_t)pass * B + user;
}

// Work-group barrier for LDS hand-overs: waits for this wave's LDS operations only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)): every barrier of a layer chain then waits for the layer's global stores (pre-activations, slab
// tiles, the Adam state) and for loads requested ahead on purpose - measured 1.9 k cycles per layer in the forward, 4.6 k per chain iteration in
// the backward, against ~0.5 k of MFMA + LDS work.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A layer's activations prelu(v) stand in for its pre-activations v in the row-owned backward (rowchain.h: skip_pre) when the slope
// is at least this: positive, and far enough from zero that slope * v neither underflows nor loses min(v, 0) to rounding
constexpr float SLOPE_FROM_ACT_MIN = 1e-6f;
constexpr float MU = 0.1f;          // score_matching_loss(..., mu=.1), :333
constexpr float MU2 = 0.01f;        // mu ** 2, :196

// ---------------------------------------------------------------------------------------------
// Per-step time-embedding tables (only T+1 distinct timesteps exist):
//   E[t]  = temb[t] * We^T + be                    (:98-99)
//   C0[t] = E[t] * W0[:, L:]^T                     (the emb part of dnn.0, :101-102)
// written as C0^T into the trailing columns of the padded layer-0 weight W0c[w][LP + t]; for sampling
// (all rows share t) also B0tab[t][w] = b0[w] + C0[t][w].
struct EmbTabArgs {
  const float* temb; const float* We; const float* be; const float* W0; const float* b0;
  float* W0c; float* B0tab;
  int L, W, T, LP, WP, K0;
  int ones_col;   // pad column of B0tab that holds 1.0 in every row (-1: none): the layer-0 pre-activation of that column is then 1
                  // in every stacked row, the "ones column" the strip-owned weight gradients take the bias gradients from (wgrad2.h)
  // k_emb_tables only: `warm_lines` 128-byte lines from `warm` on are touched while the tables are built - the batch x0 the
  // row-owned forward stages next (rowchain.h): its first loads then find the lines in L2 instead of paying a cold HBM round
  // trip with every CU asking at once (3.9 us of that kernel's staging were that wait)
  const float* warm; unsigned warm_lines;
};

// sum_t x[t*sx] * y[t*sy]: the table kernels are pure latency chains, so each batch issues 16 + 16
// independent loads before the first FMA needs one (a handful of memory round trips instead of n/4).
__device__ __forceinline__ float dot_strided(const float* __restrict__ x, long sx, const float* __restrict__ y, long sy,
                                             int n) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int t = 0;
  for (; t + 16 <= n; t += 16) {
    float xv[16], yv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { xv[u] = x[(t + u) * sx]; yv[u] = y[(t + u) * sy]; }
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      s0 = fmaf(xv[u], yv[u], s0); s1 = fmaf(xv[u + 1], yv[u + 1], s1);
      s2 = fmaf(xv[u + 2], yv[u + 2], s2); s3 = fmaf(xv[u + 3], yv[u + 3], s3);
    }
  }
  if (t < n) {   // remainder as one more predicated batch (a scalar tail loop would be n%16 dependent round trips)
    float xv[16], yv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const bool in = t + u < n;
      xv[u] = in ? x[(t + u) * sx] : 0.f;
      yv[u] = in ? y[(t + u) * sy] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      s0 = fmaf(xv[u], yv[u], s0); s1 = fmaf(xv[u + 1], yv[u + 1], s1);
      s2 = fmaf(xv[u + 2], yv[u + 2], s2); s3 = fmaf(xv[u + 3], yv[u + 3], s3);
    }
  }
  return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ float dot_unrolled(const float* __restrict__ x, const float* __restrict__ y, int n) {
  return dot_strided(x, 1, y, 1, n);
}

// sum over t in [lane, n) step 4 of x[t*sx] * y[t*sy]: a quarter of a dot product per lane of a quad, NB elements per
// batch all in flight at once (one memory round trip per batch), then the quarters meet through the wave.
template <int NB>
__device__ __forceinline__ float dot_quad(const float* __restrict__ x, int sx, const float* __restrict__ y, int sy, int n,
                                          int lane4) {
  float s0 = 0.f, s1 = 0.f;
  for (int t0 = lane4; t0 < n; t0 += 4 * NB) {
    float xv[NB], yv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = t0 + 4 * u;
      const bool in = t < n;
      xv[u] = in ? x[(unsigned)t * (unsigned)sx] : 0.f;
      yv[u] = in ? y[(unsigned)t * (unsigned)sy] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NB; u += 2) {
      s0 = fmaf(xv[u], yv[u], s0);
      if (u + 1 < NB) s1 = fmaf(xv[u + 1], yv[u + 1], s1);
    }
  }
  float s = s0 + s1;
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  return s;
}

// The same row with FOUR lanes per output (1024 threads): a dot product's loads are all in flight at once - one memory round
// trip per phase instead of five.  The stand-alone launch sits on the critical path in front of the row-owned forward
// (10.5 -> 5 us at ML-1M).
__device__ __forceinline__ void emb_tables_row4(const EmbTabArgs& a, int t, float* sh /* [2*T] */) {
  float* tr = sh;
  float* er = sh + a.T;
  for (int i = threadIdx.x; i < a.T; i += blockDim.x) tr[i] = a.temb[(size_t)t * a.T + i];
  __syncthreads();
  const int q = threadIdx.x >> 2, l4 = threadIdx.x & 3, nq = blockDim.x >> 2;
  for (int j = q; j < a.T; j += nq) {
    const float s = a.be[j] + dot_quad<20>(a.We + (size_t)j * a.T, 1, tr, 1, a.T, l4);
    if (l4 == 0) er[j] = s;
  }
  __syncthreads();
  const int ldw = a.L + a.T;
  for (int w = q; w < a.WP; w += nq) {
    const float s = (w < a.W) ? dot_quad<20>(a.W0 + (size_t)w * ldw + a.L, 1, er, 1, a.T, l4) : 0.f;
    if (l4 == 0) {
      a.W0c[(size_t)w * a.K0 + a.LP + t] = s;
      if (a.B0tab) a.B0tab[(size_t)t * a.WP + w] = (w < a.W) ? s + a.b0[w] : (w == a.ones_col ? 1.f : 0.f);
    }
  }
}

// One row t of the tables for the 64 output columns of chunk `wc` only, on a 256-thread block (64 quads of lanes): the E row is
// made again by every chunk's block (two round trips), its 64 columns of C0 are one more - k_prep_train's leading blocks, which
// were a chain of fifteen round trips (10 us, the launch's long pole) when one block of single lanes made a whole row.  Same
// arithmetic, lane for lane, as emb_tables_row4.
__device__ __forceinline__ void emb_tables_chunk(const EmbTabArgs& a, int t, int wc, float* sh /* [2*T] */) {
  float* tr = sh;
  float* er = sh + a.T;
  for (int i = threadIdx.x; i < a.T; i += blockDim.x) tr[i] = a.temb[(size_t)t * a.T + i];
  __syncthreads();
  const int q = threadIdx.x >> 2, l4 = threadIdx.x & 3, nq = blockDim.x >> 2;
  for (int j = q; j < a.T; j += nq) {
    const float s = a.be[j] + dot_quad<20>(a.We + (size_t)j * a.T, 1, tr, 1, a.T, l4);
    if (l4 == 0) er[j] = s;
  }
  __syncthreads();
  const int ldw = a.L + a.T;
  const int w = nq * wc + q;
  if (w < a.WP) {
    const float s = (w < a.W) ? dot_quad<20>(a.W0 + (size_t)w * ldw + a.L, 1, er, 1, a.T, l4) : 0.f;
    if (l4 == 0) {
      a.W0c[(size_t)w * a.K0 + a.LP + t] = s;
      if (a.B0tab) a.B0tab[(size_t)t * a.WP + w] = (w < a.W) ? s + a.b0[w] : (w == a.ones_col ? 1.f : 0.f);
    }
  }
}

__global__ __launch_bounds__(1024) void k_emb_tables(const EmbTabArgs a) {
  extern __shared__ float sh[];  // [2*T]: temb row, E row
  float touched = 0.f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < a.warm_lines; i += gridDim.x * blockDim.x) touched += a.warm[(size_t)i * 32];
  emb_tables_row4(a, blockIdx.x, sh);
  if (touched == 1.2345678e-30f) a.W0c[a.LP] = touched;   // (keeps the loads; never true in practice, harmless if it were: one table entry)
}

// four consecutive columns of an unpadded [rows, L] matrix (vector load when rows are 16-B aligned)
__device__ __forceinline__ float4 load4_unpadded(const float* __restrict__ x, int r, int c, int L) {
  const float* p = x + (size_t)r * L + c;
  if ((L & 3) == 0) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < L) v.x = p[0];
  if (c + 1 < L) v.y = p[1];
  if (c + 2 < L) v.z = p[2];
  if (c + 3 < L) v.w = p[3];
  return v;
}

// ---------------------------------------------------------------------------------------------
// Train-step staging: q_sample (:203) + the three dropout-ed inputs (:100) + one-hot(t) columns.
//   U[0*B + r] = 2*keep1 * (sqrt(abar[t]) x0 + (1-abar[t]) eps)      pass P (:328,:331)
//   U[1*B + r] = 2*keep2 * x0                                         pass S (:193)
//   U[2*B + r] = 2*keep3 * (x0 + 0.1 eps)                             pass Q (:194-195)
// columns [L,LP) zero, columns LP + i = temb[t][i] (the row's sinusoidal time embedding, train_SDRM.py:105-112: the layer-0
// weight gradient then delivers M = dpre0^T * temb beside d dnn.0.weight[:, :L], and every gradient of the embedding path
// is a product of M with the parameters - tail.h).  Rows [3B, MP) are zero-filled.
struct PrepTrainArgs {
  const float* x0; const float* noise; const int64_t* t; const uint8_t* keep;
  const float* sqrt_ab; const float* one_minus_ab;
  const float* tembP;   // [T+1][K0 - LP]: the time-embedding table, rows padded with zeros
  float* U; int* tdev;
  int B, L, LP, K0, T, MP;
  int mode; uint32_t seed_lo, seed_hi, step; int64_t row0; float nd;
  EmbTabArgs emb; int emb_row0;   // staged + zero-pad rows; the first `emb_blocks` blocks build the step's embedding
  int emb_blocks, emb_chunks;     // tables (independent work, merged here so it runs beside the staging): (T + 1) rows x chunks of 64 columns
};

__global__ __launch_bounds__(256) void k_prep_train(const PrepTrainArgs a) {
  extern __shared__ float sh[];
  if ((int)blockIdx.x < a.emb_blocks) {   // leading blocks (dispatched first; the staging blocks behind them fill the rest of
    const int t = (int)blockIdx.x / a.emb_chunks;   // the chip meanwhile): row t of the tables, 64 output columns each
    emb_tables_chunk(a.emb, t, (int)blockIdx.x - t * a.emb_chunks, sh);
    return;
  }
  // one thread per group of four columns of one staged row triple (16-byte stores; the row's t is drawn once
  // per thread instead of once per column pair)
  const int QP = a.K0 >> 2;
  // (staged rows x column quads: below 2^32 for every shape sdrm_create accepts - 32-bit divisions: a 64-bit one is ~100 instructions)
  const uint32_t i0 = (uint32_t)((int)blockIdx.x - a.emb_blocks) * 256u;
  const uint32_t i = i0 + threadIdx.x;
  const int r = (int)(i / (uint32_t)QP);
  const int c = 4 * (int)(i - (uint32_t)r * (uint32_t)QP);
  // the timesteps of the (at most 18: K0 >= 64) rows this block touches, drawn once per row
  __shared__ int tts[20];
  const int r_first = (int)(i0 / (uint32_t)QP), r_last = (int)((i0 + 255u) / (uint32_t)QP);
  if ((int)threadIdx.x <= r_last - r_first) {
    const int rr = r_first + (int)threadIdx.x;
    int t0 = 0;
    if (rr < a.B) {
      if (a.mode == 0) {
        t0 = (int)a.t[rr];
      } else {
        const U4 w = philox4x32_10((uint32_t)(a.row0 + rr), 0u, PURPOSE_TRAIN_T, a.step, a.seed_lo, a.seed_hi);
        t0 = 1 + (int)bounded(w.x, (uint32_t)a.T);
      }
      t0 = min(max(t0, 0), a.T);
      a.tdev[rr] = t0;   // (rows shared by two blocks are written twice with the same value)
    }
    tts[threadIdx.x] = t0;
  }
  __syncthreads();
  if (r >= a.emb_row0) return;
  if (r >= a.B) {  // zero pad rows
    const int row = 3 * a.B + (r - a.B);
    if (row < a.MP) *reinterpret_cast<float4*>(a.U + (size_t)row * a.K0 + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int tt = tts[r - r_first];
  float vP[4] = {0.f, 0.f, 0.f, 0.f}, vS[4] = {0.f, 0.f, 0.f, 0.f}, vQ[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < a.LP) {
    if (c < a.L) {
      const float sa = a.sqrt_ab[tt], om = a.one_minus_ab[tt];
      const float4 X = load4_unpadded(a.x0, r, c, a.L);
      const float x_[4] = {X.x, X.y, X.z, X.w};
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t bits[4] = {0u, 0u, 0u, 0u};
      if (a.mode != 0) {
        // ONE Philox call per group of four columns: (x, y) and (z, w) give two normal pairs (the upper 24 bits of
        // each word), the low byte of word j carries the three keep bits of column j (integer multiplies are what
        // this kernel is bound by; the counter is the column quad)
        const U4 w = philox4x32_10((uint32_t)(a.row0 + r), (uint32_t)(c >> 2), PURPOSE_TRAIN_ELEM, a.step, a.seed_lo, a.seed_hi);
        box_muller(w.x, w.y, e[0], e[1]);
        box_muller(w.z, w.w, e[2], e[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] *= a.nd;
        bits[0] = w.x; bits[1] = w.y; bits[2] = w.z; bits[3] = w.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = c + j;
        if (cc < a.L) {
          const size_t idx = (size_t)r * a.L + cc;
          const float x = x_[j];
          bool k1, k2, k3;
          float ee;
          if (a.mode == 0) {
            ee = a.noise[idx];
            const size_t BL = (size_t)a.B * a.L;
            k1 = a.keep[idx] != 0; k2 = a.keep[BL + idx] != 0; k3 = a.keep[2 * BL + idx] != 0;
          } else {
            ee = e[j];
            const uint32_t bb = bits[j];
            k1 = bb & 1u; k2 = (bb >> 1) & 1u; k3 = (bb >> 2) & 1u;
          }
          const float xp = sa * x + om * ee;
          const float xq = x + MU * ee;
          vP[j] = k1 ? 2.f * xp : 0.f;
          vS[j] = k2 ? 2.f * x : 0.f;
          vQ[j] = k3 ? 2.f * xq : 0.f;
        }
      }
    }
  } else {
    const int h = c - a.LP;  // the row's time embedding (the three passes share t)
    const float4 te = *reinterpret_cast<const float4*>(a.tembP + (size_t)tt * (a.K0 - a.LP) + h);
    vP[0] = vS[0] = vQ[0] = te.x; vP[1] = vS[1] = vQ[1] = te.y; vP[2] = vS[2] = vQ[2] = te.z; vP[3] = vS[3] = vQ[3] = te.w;
  }
  *reinterpret_cast<float4*>(a.U + (size_t)r * a.K0 + c) = make_float4(vP[0], vP[1], vP[2], vP[3]);
  *reinterpret_cast<float4*>(a.U + (size_t)(a.B + r) * a.K0 + c) = make_float4(vS[0], vS[1], vS[2], vS[3]);
  *reinterpret_cast<float4*>(a.U + (size_t)(2 * a.B + r) * a.K0 + c) = make_float4(vQ[0], vQ[1], vQ[2], vQ[3]);
}

// Staging for a plain forward / reverse step on caller rows: U = 2*keep*x | one-hot(t).
struct PrepFwdArgs {
  const float* x; const int64_t* t; int t_uniform; const uint8_t* keep;
  float* U; int n, L, LP, K0, T, MP;
  int mode; uint32_t seed_lo, seed_hi, step; int64_t row0;
  int bpr;   // work-groups per row: the row is folded into grid.x (grid.y stops at 65535, max_rows does not)
};

__global__ __launch_bounds__(256) void k_prep_forward(const PrepFwdArgs a) {
  const int r = blockIdx.x / a.bpr;
  const int q = (blockIdx.x - r * a.bpr) * 256 + threadIdx.x;
  const int c = 2 * q;
  if (c >= a.K0 || r >= a.MP) return;
  float2 o = make_float2(0.f, 0.f);
  if (r < a.n) {
    int tt = a.t ? (int)a.t[r] : a.t_uniform;
    tt = min(max(tt, 0), a.T);
    if (c < a.LP) {
      uint32_t bits = 0;
      if (a.mode != 0 && c < a.L) {
        const U4 w = philox4x32_10((uint32_t)(a.row0 + r), (uint32_t)q, PURPOSE_FORWARD, a.step, a.seed_lo, a.seed_hi);
        bits = w.z;
      }
      float v[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cc = c + j;
        if (cc < a.L) {
          const size_t idx = (size_t)r * a.L + cc;
          const bool k = (a.mode == 0) ? (a.keep[idx] != 0) : (((bits >> (8 * j)) & 1u) != 0);
          v[j] = k ? 2.f * a.x[idx] : 0.f;
        }
      }
      o = make_float2(v[0], v[1]);
    } else {
      const int h = c - a.LP;
      o = make_float2(h == tt ? 1.f : 0.f, (h + 1) == tt ? 1.f : 0.f);
    }
  }
  *reinterpret_cast<float2*>(a.U + (size_t)r * a.K0 + c) = o;
}

// ---------------------------------------------------------------------------------------------
// Loss partial sums (:196-198): R = P - x0, D = (Q-S)/mu^2 - R.
struct LossArgs {
  const float* Y; const float* x0; int B, L, LP;
  double* part;   // [gridDim.x][4]
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

// four sums at once, each by block_sum's own tree (shuffles down the wave, then the waves' results added in wave order): the same
// bits, one barrier pair instead of four.  sh: [4 * waves].  The totals are valid in thread 0 only.
__device__ __forceinline__ void block_sum4_seq(double (&v)[4], double* sh) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_down(v[j], off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[4 * wave + j] = v[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double t = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[4 * w + j];
      v[j] = t;
    }
  }
}

// 1024 threads per work-group: an iteration is one dependent round trip (four independent 16-byte loads, then the
// sums), so the launch takes as long as its iterations per thread - 3 instead of 11 at B = 8192, L = 340.
__global__ __launch_bounds__(1024) void k_loss_partials(const LossArgs a) {
  __shared__ double sh[4 * 16];
  double sD = 0, sC = 0, sR = 0, sR2 = 0;
  const int QP = a.LP >> 2;
  const uint32_t total = (uint32_t)a.B * (uint32_t)QP;   // (rows x column quads: below 2^32 for every shape sdrm_create accepts)
  const size_t BLP = (size_t)a.B * a.LP;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = (int)(i / (uint32_t)QP), c = 4 * (int)(i - (uint32_t)r * (uint32_t)QP);   // (32-bit: a 64-bit division is ~100 instructions)
    if (c >= a.L) continue;
    const size_t yi = (size_t)r * a.LP + c;
    const float4 P = *reinterpret_cast<const float4*>(a.Y + yi);
    const float4 S = *reinterpret_cast<const float4*>(a.Y + BLP + yi);
    const float4 Q = *reinterpret_cast<const float4*>(a.Y + 2 * BLP + yi);
    const float4 X = load4_unpadded(a.x0, r, c, a.L);
    const float p_[4] = {P.x, P.y, P.z, P.w}, s_[4] = {S.x, S.y, S.z, S.w}, q_[4] = {Q.x, Q.y, Q.z, Q.w},
                x_[4] = {X.x, X.y, X.z, X.w};
    float fD = 0.f, fC = 0.f, fR = 0.f, fR2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c + j < a.L) {
        const float R = p_[j] - x_[j];
        const float D = (q_[j] - s_[j]) / MU2 - R;
        const float RS = R - s_[j];
        fD += D * D; fC += RS * RS; fR += R; fR2 += R * R;
      }
    }
    sD += fD; sC += fC; sR += fR; sR2 += fR2;
  }
  double t4[4] = {sD, sC, sR, sR2};
  block_sum4_seq(t4, sh);
  if (threadIdx.x == 0) {
    double* o = a.part + 4 * (size_t)blockIdx.x;
    o[0] = t4[0]; o[1] = t4[1]; o[2] = t4[2]; o[3] = t4[3];
  }
}

// The five global sums, in block order (deterministic).  (Folding this into the last block of k_loss_partials
// through a ticket was measured: 6.4 us of fences and a serial tail against 4.8 us for this launch.)
__global__ __launch_bounds__(256) void k_loss_sums(const double* part, int nblk, double count, double* sums) {
  __shared__ double sh[4];
  double v[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblk; i += blockDim.x)
    for (int j = 0; j < 4; ++j) v[j] += part[4 * (size_t)i + j];
  for (int j = 0; j < 4; ++j) {
    const double s = block_sum(v[j], sh);
    if (threadIdx.x == 0) sums[j] = s;
  }
  if (threadIdx.x == 0) sums[4] = count;
}

// Loss value and closed-form gradient seeds, times tanh' = 1 - y^2.
struct SeedArgs {
  const double* sums; const float* Y; const float* x0; float* dY; float* loss;
  int B, L, LP, MP;
  int grouped;   // stacked row order (stacked_row above)
  // single-GPU fused step: sums == null and every block folds the k_loss_partials output itself (same reduction tree
  // as k_loss_sums, so the same bits) - one launch less per train step
  const double* part; int nblk; double count;
};

__global__ __launch_bounds__(256) void k_loss_seed(const SeedArgs a) {
  // the five sums first: the fold below has block barriers, so it sits ahead of every early exit
  double s0, s1, s2, s3, N;
  if (a.sums) {
    s0 = a.sums[0]; s1 = a.sums[1]; s2 = a.sums[2]; s3 = a.sums[3]; N = a.sums[4];
  } else {
    __shared__ double shs[4 * 4], tot[4];
    double v[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < a.nblk; i += blockDim.x)
      for (int j = 0; j < 4; ++j) v[j] += a.part[4 * (size_t)i + j];
    block_sum4_seq(v, shs);   // (k_loss_sums' tree per sum, one barrier pair for the four)
    if (threadIdx.x == 0)
      for (int j = 0; j < 4; ++j) tot[j] = v[j];
    __syncthreads();
    s0 = tot[0]; s1 = tot[1]; s2 = tot[2]; s3 = tot[3]; N = a.count;
  }
  const int QP = a.LP >> 2;
  const double A = s0 / N, C = s1 / N, Rbar = s2 / N;
  // unbiased variance; a single element gives 0/0 = NaN exactly as torch.var does (train_SDRM.py:198)
  const double V = (N > 1.0) ? (s3 - N * Rbar * Rbar) / (N - 1.0) : __builtin_nan("");
  const double den = 1e-8 + V;
  const double k = 0.5 / den;
  const float cD = (float)(2.0 * k / N);
  const float cV = (float)(-(0.5 * (A + C) / (den * den)) * 2.0 / (N - 1.0));
  const float rbar = (float)Rbar;
  // items: user slots (three rows each), then the padding rows behind them (one row each)
  const int gu = a.grouped == 2 ? 16 : RC_USERS;   // users per group of the grouped orders
  const int nslots = a.grouped ? gu * ((a.B + gu - 1) / gu) : a.B;
  const int ntail = a.MP - 3 * nslots;
  const size_t total = (size_t)(nslots + ntail) * QP;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // grid-stride (the host caps the grid at 2048 work-groups): the fold above - eight block barriers - is paid once for the
  // two or three column quads a thread then handles (B = 8192: train step 580.5 -> 578.3 us on one box)
  for (size_t flat = (size_t)blockIdx.x * 256 + threadIdx.x; flat < total; flat += (size_t)gridDim.x * 256) {
    const int r = (int)((uint32_t)flat / (uint32_t)QP);   // (rows x column quads: below 2^32; a 64-bit division is ~100 instructions)
    const int c = 4 * (int)((uint32_t)flat - (uint32_t)r * (uint32_t)QP);
    if (r >= nslots) {
      *reinterpret_cast<float4*>(a.dY + (size_t)(3 * nslots + (r - nslots)) * a.LP + c) = zero;
      continue;
    }
    const size_t yP = stacked_row(a.grouped, 0, r, a.B) * a.LP + c, yS = stacked_row(a.grouped, 1, r, a.B) * a.LP + c,
                 yQ = stacked_row(a.grouped, 2, r, a.B) * a.LP + c;
    if (r >= a.B) {   // grouped order: a user slot of the last group beyond the batch
      *reinterpret_cast<float4*>(a.dY + yP) = zero;
      *reinterpret_cast<float4*>(a.dY + yS) = zero;
      *reinterpret_cast<float4*>(a.dY + yQ) = zero;
      continue;
    }
    if (r == 0 && c == 0 && a.loss) *a.loss = (float)(0.5 * (A + C) / den);
    float gP[4] = {0.f, 0.f, 0.f, 0.f}, gS[4] = {0.f, 0.f, 0.f, 0.f}, gQ[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < a.L) {
      const float4 P4 = *reinterpret_cast<const float4*>(a.Y + yP);
      const float4 S4 = *reinterpret_cast<const float4*>(a.Y + yS);
      const float4 Q4 = *reinterpret_cast<const float4*>(a.Y + yQ);
      const float4 X4 = load4_unpadded(a.x0, r, c, a.L);
      const float p_[4] = {P4.x, P4.y, P4.z, P4.w}, s_[4] = {S4.x, S4.y, S4.z, S4.w}, q_[4] = {Q4.x, Q4.y, Q4.z, Q4.w},
                  x_[4] = {X4.x, X4.y, X4.z, X4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c + j < a.L) {
          const float P = p_[j], S = s_[j], Q = q_[j];
          const float R = P - x_[j];
          const float D = (Q - S) / MU2 - R;
          const float gD = cD * D;
          const float gC = cD * (R - S);
          const float gV = cV * (R - rbar);
          gP[j] = (-gD + gC + gV) * (1.f - P * P);
          gQ[j] = (gD / MU2) * (1.f - Q * Q);
          gS[j] = (-gD / MU2 - gC) * (1.f - S * S);
        }
      }
    }
    *reinterpret_cast<float4*>(a.dY + yP) = make_float4(gP[0], gP[1], gP[2], gP[3]);
    *reinterpret_cast<float4*>(a.dY + yS) = make_float4(gS[0], gS[1], gS[2], gS[3]);
    *reinterpret_cast<float4*>(a.dY + yQ) = make_float4(gQ[0], gQ[1], gQ[2], gQ[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// Adam on a given flat gradient + re-pack of the padded compute copies (sdrm_adam_step, sdrm_set_params; the single-GPU step
// applies Adam straight from the weight-gradient slabs: tail.h).
struct Job {
  int64_t flat_off;   // offset of the tensor in the flat vectors
  int rows, cols;     // logical shape; flat row stride = flat_ld
  int flat_ld;
  int ncols;          // columns [0,ncols) of each row have a compute copy (dnn.0.weight: L of L+T)
  float* dst; int dst_ld;                                          // compute copy (may be null)
  float* dst2; int dst2_ld;                                        // a second padded copy, of columns [ncols, cols) (the embedding columns of
                                                                   // dnn.0.weight for the narrow nets' forward; may be null)
  float* dstT; int dstT_ld;                                        // transposed compute copy [col][row] (may be null)
  float* dstF; float* dstFT; int fnct;                             // fragment-packed copies of the matrix / of its transpose for the
                                                                   // row-owned kernels (rowchain.h; may be null), fnct column tiles
  int fklast, fklastT;                                             // their compact last K-step (wfrag_index), -1: none
};

// element (n, k) of a padded row-major [NP][KP] matrix in its fragment-packed copy (rowchain.h): [k / 16][n / 16][lane][k % 4],
// lane = 16 * ((k / 4) % 4) + n % 16 - one contiguous 1 KiB wave-load per 16-column tile and 16-deep K-step
// klast >= 0: K-step `klast` - the last one, when at most four of its sixteen k are real - is stored COMPACT: its k = 16 klast + j
// sits in lane group j, component 0, so that ONE MFMA (the components 0 of the four lane groups) covers the four real k and the
// K-step's other three MFMAs are not issued (rc_light_klast): 340 = 21 * 16 + 4 costs 85 k-groups of four instead of 88.
__host__ __device__ inline size_t wfrag_index(int n, int k, int NCT, int klast = -1) {
  const int ct = n >> 4, nn = n & 15, ks = k >> 4;
  int kg = (k >> 2) & 3, e = k & 3;
  if (ks == klast) { kg = k & 3; e = (k >> 2) & 3; }
  return ((((size_t)ks * NCT + ct) * 64) + (size_t)(kg * 16 + nn)) * 4 + e;
}
// the compact last K-step of a reduction axis of K real entries padded to KP: its index KP / 16 - 1, or -1 when that K-step
// holds more than four real k (or none: a width whose padding spans a whole K-step keeps the plain layout)
__host__ __device__ inline int rc_light_klast(int K, int KP) {
  const int r = K - (KP - 16);
  return (KP >= 32 && r >= 1 && r <= 4) ? KP / 16 - 1 : -1;
}
constexpr int MAX_JOBS = 12;
struct JobTable { Job j[MAX_JOBS]; int n; };

// torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-4), coupled L2 (:309, Q8):
//   g += wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
struct AdamArgs {
  float* p; float* m; float* v; const float* g;
  float step_size, bc2_sqrt, b1, b2, eps, wd;
  int update;   // 0: only re-pack compute copies from p (set_params)
};

// One element's update, written once for every kernel that applies Adam (k_adam here, k_tail / k_tail_emb in tail.h), with
// contraction off and the one fused multiply-add spelled out: the single-GPU step (Adam straight from the slab sums) and the
// three-phase step (the same sums through the flat gradient, then k_adam) must leave the same bits in p, m and v whatever the
// compiler would fuse in either context (tests/test_rccl_exchange.py).
// sqrt and the two divisions through the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp each; the IEEE-exact expansions are ~10
// instructions apiece, and with four elements per thread they were 2 us of a tail work-group): the update term lr * m / (..)
// moves by 1e-7 of itself, far inside the 1e-4 parity bar and below the reference's own reorder noise.
__device__ __forceinline__ float adam_math(float w, float g, float& m, float& v, float step_size, float bc2_sqrt, float b1, float b2,
                                           float eps, float wd) {
#pragma clang fp contract(off)
  const float gg = fmaf(wd, w, g);
  const float mm = b1 * m + (1.f - b1) * gg;
  const float vv = b2 * v + ((1.f - b2) * gg) * gg;
  m = mm; v = vv;
  const float denom = __builtin_amdgcn_sqrtf(vv) * __builtin_amdgcn_rcpf(bc2_sqrt) + eps;
  return w - step_size * (mm * __builtin_amdgcn_rcpf(denom));
}

__device__ __forceinline__ float adam_element(const AdamArgs& a, int64_t fi) {
  float w = a.p[fi];
  if (a.update) {
    float mm = a.m[fi], vv = a.v[fi];
    w = adam_math(w, a.g[fi], mm, vv, a.step_size, a.bc2_sqrt, a.b1, a.b2, a.eps, a.wd);
    a.m[fi] = mm; a.v[fi] = vv;
    a.p[fi] = w;
  }
  return w;
}

// One grid row (blockIdx.y) per parameter tensor.  Tensors that keep a transposed compute copy go tile by tile
// (32 x 32 through LDS, so both copies are written along their contiguous index); the rest element by element.
__global__ __launch_bounds__(256) void k_adam(const JobTable tab, const AdamArgs a) {
  const Job& jb = tab.j[blockIdx.y];
  if (jb.dstT != nullptr) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int tr = (jb.rows + 31) >> 5, tc = (jb.cols + 31) >> 5;
    for (int tix = blockIdx.x; tix < tr * tc; tix += gridDim.x) {
      const int r0 = (tix / tc) * 32, c0 = (tix % tc) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float w = 0.f;
        if (r < jb.rows && c < jb.cols) {
          w = adam_element(a, jb.flat_off + (int64_t)r * jb.flat_ld + c);
          if (jb.dst != nullptr && c < jb.ncols) jb.dst[(size_t)r * jb.dst_ld + c] = w;
          if (jb.dstF != nullptr && c < jb.ncols) jb.dstF[wfrag_index(r, c, jb.fnct, jb.fklast)] = w;
        }
        tile[ty + 8 * k][tx] = w;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < jb.rows && c < jb.ncols) {
          jb.dstT[(size_t)c * jb.dstT_ld + r] = tile[tx][ty + 8 * k];
          if (jb.dstFT != nullptr) jb.dstFT[wfrag_index(c, r, jb.fnct, jb.fklastT)] = tile[tx][ty + 8 * k];
        }
      }
      __syncthreads();
    }
    return;
  }
  const int64_t total = (int64_t)jb.rows * jb.cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / jb.cols), c = (int)(i - (int64_t)r * jb.cols);
    const float w = adam_element(a, jb.flat_off + (int64_t)r * jb.flat_ld + c);
    if (jb.dst != nullptr && c < jb.ncols) jb.dst[(size_t)r * jb.dst_ld + c] = w;
    if (jb.dst2 != nullptr && c >= jb.ncols) jb.dst2[(size_t)r * jb.dst2_ld + c - jb.ncols] = w;
    if (jb.dstF != nullptr && c < jb.ncols) jb.dstF[wfrag_index(r, c, jb.fnct, jb.fklast)] = w;
  }
}

// ---------------------------------------------------------------------------------------------
// Sampling helpers.
// Sampler state: slot s of X/U holds original row rowid[s] (identity when rowid == null).  For the
// multi-resolution branch the host orders slots by DESCENDING start step Tj, so the rows active at step i
// (Tj >= i) are always the prefix [0, n_act(i)) and every step's GEMMs run on exactly the active rows —
// the same row-steps as the reference's per-user batch-1 loops (train_SDRM.py:40-48), batched.
// U[s] is pre-loaded with the dropout of the row's OWN first step (keep mask of step Tj[s]).
struct SampleInitArgs {
  const float* xT; const uint8_t* keep; const int64_t* Tj; const int* rowid;
  float* X; float* U; int n, L, LP, K0, MP, T;
  int mode; uint32_t seed_lo, seed_hi, call_id; int64_t row0;
  int bpr;   // work-groups per row (row folded into grid.x)
};

__global__ __launch_bounds__(256) void k_sample_init(const SampleInitArgs a) {
  const int s = blockIdx.x / a.bpr;
  const int q = (blockIdx.x - s * a.bpr) * 256 + threadIdx.x;   // column quad: one Philox call, 16-byte stores
  const int c = 4 * q;
  if (c >= a.LP || s >= a.MP) return;
  float x[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
  if (s < a.n) {
    const int r = a.rowid ? a.rowid[s] : s;
    const int first = a.Tj ? (int)a.Tj[s] : a.T;   // the step this row starts at
    if (c < a.L) {
      uint32_t bits[4] = {0u, 0u, 0u, 0u};
      float nrm[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.mode != 0) {
        const uint32_t grow = (uint32_t)(a.row0 + r);
        const U4 w = philox4x32_10(grow, (uint32_t)q, PURPOSE_SAMPLE_XT, a.call_id, a.seed_lo, a.seed_hi);
        box_muller(w.x, w.y, nrm[0], nrm[1]);
        box_muller(w.z, w.w, nrm[2], nrm[3]);
        // the keep bits of step i ride on the Philox call that draws z_{i+1} (see k_reverse_update)
        const U4 w2 = philox4x32_10(grow, (uint32_t)q, PURPOSE_SAMPLE_STEP | ((uint32_t)(first + 1) << 8), a.call_id,
                                    a.seed_lo, a.seed_hi);
        bits[0] = w2.x; bits[1] = w2.y; bits[2] = w2.z; bits[3] = w2.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = c + j;
        if (cc < a.L) {
          const size_t idx = (size_t)r * a.L + cc;
          x[j] = (a.mode == 0) ? a.xT[idx] : nrm[j];
          const bool k = (a.mode == 0) ? (a.keep[(size_t)first * a.n * a.L + idx] != 0) : ((bits[j] & 1u) != 0);
          u[j] = k ? 2.f * x[j] : 0.f;
        }
      }
    }
  }
  *reinterpret_cast<float4*>(a.X + (size_t)s * a.LP + c) = make_float4(x[0], x[1], x[2], x[3]);
  *reinterpret_cast<float4*>(a.U + (size_t)s * a.K0 + c) = make_float4(u[0], u[1], u[2], u[3]);
}

// One DDPM reverse update on the sampler's padded state (denoise_add_noise, train_SDRM.py:20-25) fused
// with the next step's input dropout (F.dropout, :100):
//   x <- (x - eps_hat*c1)/sqrt(alpha_i) + sqrt(beta_i)*z   for rows with Tj >= i (all rows if Tj == null)
//   U_next = 2*keep_{i-1}*x
// One thread per column quad: in PHILOX mode one Philox call yields the quad's four normals z_i and its four
// keep bits for step i-1.
struct ReverseArgs {
  float* X; const float* Y; float* U; const float* Z; const uint8_t* keep_next; const int64_t* Tj; const int* rowid;
  int s0, n, L, LP, K0, step_i; float c1, sqrt_alpha, sqrt_beta, nd;
  int mode; uint32_t seed_lo, seed_hi, call_id; int64_t row0;
  int bpr;   // work-groups per row (row folded into grid.x)
  int rpb;   // rows per work-group when a row has at most 128 column quads (then bpr == 1): 340 columns are 85 quads, three
             // rows fill 255 of the 256 lanes where one row left two thirds of them idle
};

__global__ __launch_bounds__(256) void k_reverse_update(const ReverseArgs a) {
  int rr, q;
  if (a.rpb > 1) {
    const int qpr = (a.L + 3) >> 2;
    const int sub = (int)threadIdx.x / qpr;
    if (sub >= a.rpb) return;
    rr = blockIdx.x * a.rpb + sub;
    q = (int)threadIdx.x - sub * qpr;
  } else {
    rr = blockIdx.x / a.bpr;
    q = (blockIdx.x - rr * a.bpr) * 256 + threadIdx.x;
  }
  const int s = a.s0 + rr;                       // slot; this launch covers the active slots [s0, n) of one row chain
  const int c = 4 * q;
  if (c >= a.L || s >= a.n) return;
  const int r = a.rowid ? a.rowid[s] : s;       // original row: indexes explicit randoms and keys Philox
  const size_t xi = (size_t)s * a.LP + c;
  const float4 xo4 = *reinterpret_cast<const float4*>(a.X + xi);
  const float4 e4 = *reinterpret_cast<const float4*>(a.Y + xi);
  const float xo[4] = {xo4.x, xo4.y, xo4.z, xo4.w}, e[4] = {e4.x, e4.y, e4.z, e4.w};
  const bool active = (a.Tj == nullptr) || (a.Tj[s] >= (int64_t)a.step_i);
  float z[4] = {0.f, 0.f, 0.f, 0.f};
  bool kp[4] = {false, false, false, false};
  if (a.step_i > 1) {
    if (a.mode == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < a.L) {
          const size_t idx = (size_t)r * a.L + c + j;
          z[j] = a.Z[idx];
          kp[j] = a.keep_next[idx] != 0;
        }
    } else {
      const U4 w = philox4x32_10((uint32_t)(a.row0 + r), (uint32_t)q, PURPOSE_SAMPLE_STEP | ((uint32_t)a.step_i << 8),
                                 a.call_id, a.seed_lo, a.seed_hi);
      box_muller(w.x, w.y, z[0], z[1]);
      box_muller(w.z, w.w, z[2], z[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] *= a.nd;
      kp[0] = w.x & 1u; kp[1] = w.y & 1u; kp[2] = w.z & 1u; kp[3] = w.w & 1u;
    }
  }
  float xn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    xn[j] = (c + j < a.L) ? (active ? (xo[j] - e[j] * a.c1) / a.sqrt_alpha + a.sqrt_beta * z[j] : xo[j]) : 0.f;
  *reinterpret_cast<float4*>(a.X + xi) = make_float4(xn[0], xn[1], xn[2], xn[3]);
  if (a.step_i > 1)
    *reinterpret_cast<float4*>(a.U + (size_t)s * a.K0 + c) =
        make_float4(kp[0] ? 2.f * xn[0] : 0.f, kp[1] ? 2.f * xn[1] : 0.f, kp[2] ? 2.f * xn[2] : 0.f, kp[3] ? 2.f * xn[3] : 0.f);
}

// Up to eight device-to-device copies in one launch (the sampler's snapshot of the net: eight hipMemcpyAsync calls were eight
// 5-us nodes on the stream at every sdrm_sample_begin).  Segment = blockIdx.y; float4 body, scalar tail.
struct CopySegs {
  const float* src[8]; float* dst[8]; unsigned n[8];
};

__global__ __launch_bounds__(256) void k_copy_segments(const CopySegs a) {
  const float* __restrict__ s = a.src[blockIdx.y];
  float* __restrict__ d = a.dst[blockIdx.y];
  const unsigned n = a.n[blockIdx.y];
  if (s == nullptr || n == 0) return;
  const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
  const unsigned n4 = vec ? n >> 2 : 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
  for (unsigned i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ __launch_bounds__(256) void k_unpad_rows(const float* src, int ld, float* dst, int n, int L, const int* rowid) {
  const size_t total = (size_t)n * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(i / L), c = (int)(i - (size_t)s * L);
    const int r = rowid ? rowid[s] : s;
    dst[(size_t)r * L + c] = src[(size_t)s * ld + c];
  }
}

// stacked padded rows (either row order) -> [3][B][L]
__global__ __launch_bounds__(256) void k_unpad_psq(const float* Y, int B, int L, int LP, int grouped, float* dst) {
  const size_t total = (size_t)3 * B * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / L; const int c = (int)(i - r * L);
    const int pass = (int)(r / B), user = (int)(r - (size_t)pass * B);
    dst[i] = Y[stacked_row(grouped, pass, user, B) * LP + c];
  }
}

// ... of a layer whose pre-activations the forward did not store (rowchain.h: skip_pre): v = prelu(v) where that is positive,
// prelu(v) / slope below (the slope is positive whenever the forward skipped the store; else `pre` holds them)
__global__ __launch_bounds__(256) void k_unpad_pre(const float* pre, const float* act, const float* slope, int B, int W, int WP, int grouped, float* dst) {
  const float sl = *slope;
  const bool from_act = act != nullptr && sl >= SLOPE_FROM_ACT_MIN;
  const size_t total = (size_t)3 * B * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / W; const int c = (int)(i - r * W);
    const int pass = (int)(r / B), user = (int)(r - (size_t)pass * B);
    const size_t at = stacked_row(grouped, pass, user, B) * WP + c;
    if (from_act) { const float h = act[at]; dst[i] = h > 0.f ? h : h / sl; }
    else dst[i] = pre[at];
  }
}

// Test hook (sdrm_debug_philox_draws): the generator's output exactly as the staging kernels consume it - one Philox call per
// (row, column quad): four normals (two Box-Muller pairs) and the low bits of the four words (bit b of word j = keep bit of
// pass b for column j in the train step).
__global__ __launch_bounds__(256) void k_philox_draws(uint32_t seed_lo, uint32_t seed_hi, uint32_t purpose, uint32_t step, int64_t row0, int rows,
                                                      int quads, float* __restrict__ normals, uint8_t* __restrict__ lowbits) {
  const size_t total = (size_t)rows * quads;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / quads), q = (int)(i - (size_t)r * quads);
    const U4 w = philox4x32_10((uint32_t)(row0 + r), (uint32_t)q, purpose, step, seed_lo, seed_hi);
    float n[4];
    box_muller(w.x, w.y, n[0], n[1]);
    box_muller(w.z, w.w, n[2], n[3]);
    *reinterpret_cast<float4*>(normals + 4 * i) = make_float4(n[0], n[1], n[2], n[3]);
    if (lowbits) *reinterpret_cast<uint32_t*>(lowbits + 4 * i) = (w.x & 7u) | ((w.y & 7u) << 8) | ((w.z & 7u) << 16) | ((w.w & 7u) << 24);
  }
}

// x <- (x - eps_hat * c1) / sqrt(alpha_i) + sqrt(beta_i) * z   (denoise_add_noise, :20-25)
__global__ __launch_bounds__(256) void k_reverse_apply(float* x, const float* Y, int ldy, const float* z, int n, int L,
                                                       float c1, float sqrt_alpha, float sqrt_beta) {
  const size_t total = (size_t)n * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / L), c = (int)(i - (size_t)r * L);
    const float e = Y[(size_t)r * ldy + c];
    const float zz = z ? z[i] : 0.f;
    x[i] = (x[i] - e * c1) / sqrt_alpha + sqrt_beta * zz;
  }
}

// perturb_input (:202-203)
__global__ __launch_bounds__(256) void k_perturb(const float* x, const int64_t* t, const float* noise,
                                                 const float* sqrt_ab, const float* one_minus_ab, int n, int L, int T,
                                                 float* out) {
  const size_t total = (size_t)n * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / L);
    const int tt = min(max((int)t[r], 0), T);
    out[i] = sqrt_ab[tt] * x[i] + one_minus_ab[tt] * noise[i];
  }
}

}  // namespace sdrm
