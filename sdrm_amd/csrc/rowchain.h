// Row-owned train forward for mid-width eps-nets (64 < padded width <= 352, L == W) on gfx950: the north-star's
// "persistent-threadblock MLP with LDS-resident activations", at TRAIN size.
//
// Ownership: ONE work-group per CU owns 96 stacked rows - the P, S and Q rows of 32 users (train_SDRM.py:331-333: the three
// forwards of a train step) - through staging (q_sample + three dropout masks, :326-331 / :100) and ALL H+2 layers (:97-103).
//   * the 96 x NP activation tile lives in LDS (135 KB at NP = 352) and is overwritten in place after each layer's barrier;
//     it holds the layer INPUT as it is stored for the backward (dropped-out latents for layer 0, pre-activations above:
//     PReLU is applied when a fragment is read), so the same tile is streamed out to HBM - U, pre[k], coalesced 16-byte
//     stores - a few rows per K-step while the next layer multiplies out of it: no store burst, no separate staging launch;
//   * the weights never touch LDS: every wave fetches its MFMA B fragments straight from L2 out of a FRAGMENT-PACKED copy
//     ([k-step][column tile][lane][4 floats]: one contiguous 1 KiB wave-load per 16x16 tile and 16-deep K-step; k_adam
//     writes these copies beside the padded ones), double-buffered in registers one whole K-step (132 MFMAs = 4224 cycles)
//     ahead: an L2 round trip is a tenth of that;
//   * 4 waves, one per SIMD, as 2 x 2: a wave owns 3 row tiles x CT column tiles of v_mfma_f32_16x16x4_f32 (33 accumulator
//     quads at NP = 352) - per K-step 3 ds_read_b128 + CT global loads feed 12 * CT MFMAs, and the only barriers are the two
//     around each layer's in-place epilogue (the per-layer path pays one per 8 MFMAs and a launch ramp + tail per layer);
//     the wave's three row tiles are the P, S and Q rows of the SAME 16 users, so the out layer needs no LDS at all: tanh,
//     the Y stores and the loss partial sums come straight out of the accumulators;
//   * every other instruction of a K-step (B loads of the next step, A fragment reads + PReLU, the tile stream) is cut into
//     pieces that sit in the shadows of 3-MFMA groups (sched_barrier pins the order), as in gemm.h.
// 24576 stacked rows (B = 8192) = 256 work-groups = exactly one round of the chip, no tail.
// The loss partial sums (score_matching_loss, :191-199) fall out of the out-layer epilogue: P, S, Q of a user sit in the
// same LDS tile, so k_loss_partials is not launched.
//
// Stacked row order of this path ("grouped"): row(pass, user) = 96 * (user / 32) + 32 * pass + user % 32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"
#include "gemm.h"
#include "philox.h"
#include "skinny.h"

namespace sdrm {

// RC_USERS, RC_ROWS, rc_row() and wfrag_index() live in elementwise.h (k_adam writes the fragment-packed copies, the loss
// seeds and the unpadding kernel know both stacked row orders).

// LDS row stride (floats) of the activation tile: a multiple of 4 with stride * 4 B = 32 B x odd (mod 256 B), which makes the
// ds_read_b128 fragment reads (lane -> row lane & 15, 16-byte k group lane >> 4) conflict-free for the 16-lane service groups
// of a b128 access (MI355X guide, LDS table: each group holds all 16 rows, half of them one k group further)
__host__ __device__ constexpr int rc_lda(int NP) {
  int p = 4;
  while (((NP + p) % 64) != 8 && ((NP + p) % 64) != 24 && ((NP + p) % 64) != 40 && ((NP + p) % 64) != 56) p += 4;
  return NP + p;
}

struct RowChainArgs {
  // step inputs (EXPLICIT mode: noise [B,L], t [B], keep [3,B,L]; PHILOX mode: null)
  const float* x0; const float* noise; const int64_t* t; const uint8_t* keep;
  const float* sqrt_ab; const float* one_minus_ab;
  int B, L, T, H;
  int mode; uint32_t seed_lo, seed_hi, step; int64_t row0; float nd;
  // net: fragment-packed weights (layer 0: the latent columns only), biases, per-timestep bias table of layer 0
  const float* W0f; const float* Whf; const float* Wof;
  const float* bh; const float* bo; const float* B0tab; int ldtab;
  const float* slope0; const float* slopeh;
  // outputs, grouped stacked rows
  float* U; int K0, LPs; int* tdev;
  int ones_col;                             // pad column of U set to 1.0 in every row (-1: none), see wgrad2.h
  float* pre; size_t pre_stride; int ldp;   // pre[k] = pre + k * pre_stride, [MP][ldp]
  float* act;                               // activations prelu(pre[k]) in the same layout (null: not stored)
  float* Y; int ldy;
  double* loss_part;                        // [gridDim.x][4]
  unsigned long long* stamps;               // diagnostic builds only (-DRC_STAMPS): 16 s_memtime slots per work-group
};

#ifdef RC_STAMPS
#define RC_STAMP(i) do { if (tid == 0) st_[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif

template <int CT>
struct RowChainCfg {
  static constexpr int NP = 32 * CT, NCT = 2 * CT, KS = NP / 16, QP = NP / 4, LDA = rc_lda(NP);
  static constexpr int NCHUNK = RC_ROWS * QP / NTHREADS;   // float4 per thread in one sweep over the tile (= 3 * CT)
  static constexpr size_t LDS_BYTES = (size_t)RC_ROWS * LDA * 4 + 256;
  static_assert(RC_ROWS * QP % NTHREADS == 0 && RC_USERS * QP % NTHREADS == 0, "tile sweeps must divide over the work-group");
  static_assert(LDS_BYTES <= 160 * 1024, "activation tile does not fit LDS");
};

// one K-step of a wave: 12 * CT MFMAs out of (ac, bc); in their shadows the next step's B fragments (global) and A fragments
// (LDS, PReLU on read) land in (an, bn), and two chunks of the activation tile are streamed LDS -> HBM
// diagnostic builds only (-DRC_DIAG=mask): drop a piece of the main loop to see what it costs - bit0 the tile stream, bit1 the B
// fragment loads, bit2 the A fragment reads, bit3 PReLU on read
#ifndef RC_DIAG
#define RC_DIAG 0
#endif

// fp32 MFMA runs on the SIMD's fp32 vector lanes (that is why its peak equals the vector peak): a VALU instruction of the same
// wave is NOT hidden behind it - measured (tools/mfma_filler_probe.hip, one wave per SIMD): a group of n plain VALU
// instructions between two v_mfma_f32_16x16x4_f32 costs ~6 + 2.2 n cycles of a 32-cycle MFMA slot, LDS reads and global loads
// cost nothing.  So the main loop is written for FEW VALU instructions, in FEW groups: B loads and tile-stream stores address
// through scalar bases (advanced on the scalar unit) + a constant lane offset, and PReLU on read is two instructions per value.
//
// PReLU for a slope in [0, 1] (the initial 0.25 and every trained value seen): max(v, slope * v).  The generic form
// (max(v, 0) + slope * min(v, 0), three instructions) is taken when the slope is outside that range (wave-uniform choice).
__device__ __forceinline__ float prelu_01(float v, float slope) {
  float r;
  asm("v_mul_f32 %0, %1, %2\n\tv_max_f32 %0, %0, %1" : "=&v"(r) : "v"(v), "v"(slope));
  return r;
}
__device__ __forceinline__ float prelu_any(float v, float slope) {
  float lo, r;
  asm("v_min_f32 %0, 0, %2\n\tv_max_f32 %1, 0, %2\n\tv_fmac_f32 %1, %3, %0" : "=&v"(lo), "=&v"(r) : "v"(v), "v"(slope));
  return r;
}

// a GLOBAL-address-space pointer the compiler keeps in scalar registers: loads / stores through it take the
// global_load / global_store saddr + lane-offset form (through an integer cast alone it would decay to a flat pointer:
// flat_load, which also counts in lgkmcnt)
typedef __attribute__((address_space(1))) char gchar;
__device__ __forceinline__ gchar* uniform_gptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gchar*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ f32x4 gload4(const gchar* base, uint32_t off) {
  return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + off);
}
__device__ __forceinline__ void gstore4(gchar* base, uint32_t off, float4 v) {
  f32x4 w = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<__attribute__((address_space(1))) f32x4*>(base + off) = w;
}
// the same with the non-temporal hint: bytes nobody reads before the weight gradients, a whole backward chain later (U, the
// activations) - they should not push the pre-activations and Y, which the loss seeds and the dgrads read next, out of the caches
__device__ __forceinline__ void gstore4_nt(gchar* base, uint32_t off, float4 v) {
  f32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<__attribute__((address_space(1))) f32x4*>(base + off));
}

// position of a thread in the sweep of the activation tile (one float4 per thread and chunk): chunk j covers the flat quad
// indices j * 256 + tid, row = f / QP, quad = f % QP; stepping a chunk adds 256 = (256 / QP) rows + (256 % QP) quads, with one
// wrap at most - branch-free: the K-step must stay ONE basic block (see rc_kstep)
template <int QP, int LDA>
struct RcSweep {
  static constexpr int DR = NTHREADS / QP, DQ = NTHREADS % QP;
  int cq; uint32_t lds_off, g_off;   // quad in its row; byte offsets of the quad in the LDS tile / in the HBM rows
  uint32_t gd0, gd1;                 // HBM byte step of a chunk without / with a row wrap
  __device__ __forceinline__ void init(int tid, int sld) {
    const int row = tid / QP;
    cq = tid - row * QP;
    lds_off = (uint32_t)(row * LDA + 4 * cq) * 4u;
    g_off = (uint32_t)(row * sld + 4 * cq) * 4u;
    gd0 = (uint32_t)((DR * sld + 4 * DQ) * 4);
    gd1 = (uint32_t)(((DR + 1) * sld + 4 * (DQ - QP)) * 4);
  }
  __device__ __forceinline__ void next() {
    const bool wrap = cq >= QP - DQ;
    cq += wrap ? DQ - QP : DQ;
    lds_off += wrap ? (uint32_t)(((DR + 1) * LDA + 4 * (DQ - QP)) * 4) : (uint32_t)((DR * LDA + 4 * DQ) * 4);
    g_off += wrap ? gd1 : gd0;
  }
};

// One K-step of a wave: 12 * CT MFMAs out of (ac, bc), and between them the pieces that prepare the next step: its B fragments
// (global, CT wave-loads of 1 KiB), its A fragments (3 ds_read_b128) with their PReLU as ONE group of VALU instructions, and
// NS chunks of the tile stream LDS -> HBM.  The body must stay ONE basic block with every piece where it is written: a
// branch inside it, or a load the compiler is free to hoist (loads of read-only memory are not ordered against sched_barrier),
// and all pieces end up in front of the MFMAs.  Hence compile-time piece counts, and every load's offset is passed through an
// empty asm volatile at its slot, which pins it there.
// ACT: the streamed tile holds pre-activations and their ACTIVATIONS go to `adst` as well (same offsets): the weight gradients
// then read their operand as it is - PReLU on operand load costs the split-K kernel a tenth of its time, here it is 12 VALU
// instructions per chunk.
template <int CT, int LDA, int NS, bool ACT>
__device__ __forceinline__ void rc_kstep(f32x4 (&acc)[3][CT], const f32x4 (&ac)[3], const f32x4 (&bc)[CT], f32x4 (&an)[3],
                                         f32x4 (&bn)[CT], const gchar* wnext, uint32_t lane16, uint32_t anext, float slope,
                                         const float* __restrict__ Act, gchar* sdst, gchar* adst, RcSweep<8 * CT, LDA>& sw) {
  constexpr int NSLOT = 12 * CT;
  constexpr int NPH = ACT ? 3 : 2;   // phases of a chunk: LDS read, store, (activation store)
  constexpr int P_A = CT, P_S = CT + 3, P_X = P_S + NPH * NS, NPIECE = P_X + 1;
  constexpr int STRIDE = NSLOT / NPIECE >= 1 ? NSLOT / NPIECE : 1;
  static_assert(NPIECE <= NSLOT && NS <= 2, "not enough MFMA slots for the pipeline pieces");
  float4 sv0 = make_float4(0.f, 0.f, 0.f, 0.f), sv1 = sv0;
  uint32_t so0 = 0, so1 = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int s = (e * CT + ct) * 3 + rt;
    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt][e], bc[ct][e], acc[rt][ct], 0, 0, 0);
    if (s % STRIDE == 0 && s / STRIDE < NPIECE) {
      const int p = s / STRIDE;
      if (p < P_A) {
        if (!(RC_DIAG & 2)) {
          // a scalar base per 4 KiB (the immediate offset field covers the rest), opaque at this slot: pins the load here
          const gchar* wb = wnext + (p / 4) * 4096;
          asm volatile("" : "+s"(wb));
          bn[p] = gload4(wb + (p % 4) * 1024, lane16);
        }
      } else if (p < P_S) {
        if (!(RC_DIAG & 4)) {
          uint32_t ao = anext;
          asm volatile("" : "+v"(ao));
          an[p - P_A] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Act) + ao + (p - P_A) * RC_USERS * LDA * 4);
        }
      } else if (p < P_X) {
        // tile stream, chunk q: LDS read, then (NS pieces later) the HBM store and the step to the next chunk
        const int k = p - P_S, ph = k / NS, q = k % NS;
        if (!(RC_DIAG & 1)) {
          float4& v = q == 0 ? sv0 : sv1;
          uint32_t& so = q == 0 ? so0 : so1;
          if (ph == 0) {
            uint32_t lo = sw.lds_off;
            asm volatile("" : "+v"(lo));
            v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Act) + lo);
            so = sw.g_off;
            sw.next();
          } else if (ph == 1) {
            if (ACT) gstore4(sdst, so, v);   // pre-activations: read again by the first dgrad
            else gstore4_nt(sdst, so, v);    // U: read again by the layer-0 weight gradient only
          } else {
            gstore4_nt(adst, so, make_float4(prelu_any(v.x, slope), prelu_any(v.y, slope), prelu_any(v.z, slope), prelu_any(v.w, slope)));
          }
        }
      } else {
        if (!(RC_DIAG & 8)) {
#pragma unroll
          for (int f = 0; f < 3; ++f) {
            an[f].x = prelu_any(an[f].x, slope); an[f].y = prelu_any(an[f].y, slope);
            an[f].z = prelu_any(an[f].z, slope); an[f].w = prelu_any(an[f].w, slope);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// four doubles summed over the work-group, every thread gets the totals (one barrier pair instead of four)
__device__ __forceinline__ void block_sum4(double (&v)[4], double* sh /* [16] */) {
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
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (sh[j] + sh[4 + j]) + (sh[8 + j] + sh[12 + j]);
}

template <int CT>
__global__ __launch_bounds__(NTHREADS, 1) void k_row_fwd(const RowChainArgs a) {
  typedef RowChainCfg<CT> C;
  constexpr int NP = C::NP, NCT = C::NCT, KS = C::KS, QP = C::QP, LDA = C::LDA, RT = 3;
  static_assert(KS % 2 == 0, "K-steps are taken in pairs");
  __shared__ __attribute__((aligned(16))) float Act[RC_ROWS * LDA];
  __shared__ int trow[RC_USERS];
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lq = lane >> 4;
  const int g = blockIdx.x, u0 = RC_USERS * g;
  const size_t grow0 = (size_t)RC_ROWS * g;   // first stacked row of this work-group
#ifdef RC_STAMPS
  unsigned long long st_[16] = {0};
#endif
  RC_STAMP(0);

  // ---------------------------------------------------------------- staging
  if (tid < RC_USERS) {
    const int usr = u0 + tid;
    int t0 = 0;
    if (usr < a.B) {
      if (a.mode == 0) {
        t0 = (int)a.t[usr];
      } else {
        const U4 w = philox4x32_10((uint32_t)(a.row0 + usr), 0u, PURPOSE_TRAIN_T, a.step, a.seed_lo, a.seed_hi);
        t0 = 1 + (int)bounded(w.x, (uint32_t)a.T);
      }
      t0 = min(max(t0, 0), a.T);
      a.tdev[usr] = t0;
    }
    trow[tid] = t0;
  }
  // thread -> (user tid >> 3, column quads (tid & 7) + 8 j): every x0 quad of the thread is requested before the first is used
  constexpr int NQ = QP / 8;   // = CT
  const int su = tid >> 3, sq = tid & 7;
  const int susr = u0 + su;
  float4 xs[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (sq + 8 * j);
    xs[j] = (susr < a.B && c < a.L) ? load4_unpadded(a.x0, susr, c, a.L) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  {
    const int tt = trow[su];
    const float sa = a.sqrt_ab[tt], om = a.one_minus_ab[tt];
    // PHILOX mode: the thread's NQ calls (one per column quad: two normal pairs and, in the low bits of word j, the three keep
    // bits of column j) in ONE straight-line block - a call is a serial chain of 10 rounds, and with one wave per SIMD nothing
    // but the other calls of the same thread can fill its latencies
    U4 rw[NQ];
    if (a.mode != 0) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        rw[j] = philox4x32_10((uint32_t)(a.row0 + susr), (uint32_t)(sq + 8 * j), PURPOSE_TRAIN_ELEM, a.step, a.seed_lo, a.seed_hi);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (sq + 8 * j);
      float vP[4] = {0.f, 0.f, 0.f, 0.f}, vS[4] = {0.f, 0.f, 0.f, 0.f}, vQ[4] = {0.f, 0.f, 0.f, 0.f};
      if (susr < a.B && c < a.L) {
        const float x_[4] = {xs[j].x, xs[j].y, xs[j].z, xs[j].w};
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t bits[4] = {0u, 0u, 0u, 0u};
        if (a.mode != 0) {
          const U4 w = rw[j];
          box_muller(w.x, w.y, e[0], e[1]);
          box_muller(w.z, w.w, e[2], e[3]);
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] *= a.nd;
          bits[0] = w.x; bits[1] = w.y; bits[2] = w.z; bits[3] = w.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (c + k < a.L) {
            const size_t idx = (size_t)susr * a.L + c + k;
            const float x = x_[k];
            bool k1, k2, k3;
            float ee;
            if (a.mode == 0) {
              ee = a.noise[idx];
              const size_t BL = (size_t)a.B * a.L;
              k1 = a.keep[idx] != 0; k2 = a.keep[BL + idx] != 0; k3 = a.keep[2 * BL + idx] != 0;
            } else {
              ee = e[k];
              k1 = bits[k] & 1u; k2 = (bits[k] >> 1) & 1u; k3 = (bits[k] >> 2) & 1u;
            }
            vP[k] = k1 ? 2.f * (sa * x + om * ee) : 0.f;
            vS[k] = k2 ? 2.f * x : 0.f;
            vQ[k] = k3 ? 2.f * (x + MU * ee) : 0.f;
          }
        }
      }
      if (c == (a.ones_col & ~3)) {   // the ones column (a pad column: layer 0's weights are zero there)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k == (a.ones_col & 3)) vP[k] = vS[k] = vQ[k] = 1.f;
      }
      *reinterpret_cast<float4*>(Act + su * LDA + c) = make_float4(vP[0], vP[1], vP[2], vP[3]);
      *reinterpret_cast<float4*>(Act + (RC_USERS + su) * LDA + c) = make_float4(vS[0], vS[1], vS[2], vS[3]);
      *reinterpret_cast<float4*>(Act + (2 * RC_USERS + su) * LDA + c) = make_float4(vQ[0], vQ[1], vQ[2], vQ[3]);
    }
  }
  {
    // one-hot(t) columns of U (read by the layer-0 weight gradient only); rows of users beyond the batch stay all-zero
    const int TQ = (a.K0 - a.LPs) >> 2;
    for (int f = tid; f < RC_ROWS * TQ; f += NTHREADS) {
      const int row = f / TQ, h = 4 * (f - row * TQ);
      const int u = row & (RC_USERS - 1);
      const int tt = (u0 + u < a.B) ? trow[u] : -1;
      *reinterpret_cast<float4*>(a.U + (grow0 + row) * a.K0 + a.LPs + h) =
          make_float4(h == tt ? 1.f : 0.f, h + 1 == tt ? 1.f : 0.f, h + 2 == tt ? 1.f : 0.f, h + 3 == tt ? 1.f : 0.f);
    }
  }
  __syncthreads();
  RC_STAMP(1);

  // ---------------------------------------------------------------- layers
  // wave (wr, wc): row tile rt = pass rt of users 16 wr .. 16 wr + 15 (tile rows 32 rt + 16 wr ..), column tiles CT wc ..
  const float* abase = Act + (16 * wr + li) * LDA + 4 * lq;     // + 32 * rt * LDA + 16 * ks
  const uint32_t aoff = (uint32_t)(((16 * wr + li) * LDA + 4 * lq) * 4);   // the same as a byte offset into the tile
  const uint32_t lane16 = 16u * (uint32_t)lane;   // byte offset of a lane's float4 in a 1 KiB wave-load / wave-store
  const int myusr = u0 + 16 * wr + 4 * lq;                      // + r: the user of accumulator register r (every rt, ct)
  f32x4 acc[RT][CT];
  f32x4 b0[CT], b1[CT];
  f32x4 a0[RT], a1[RT];
  float xq[CT][4];   // x0 at this lane's accumulator positions (loss sums), requested before the out layer's loop

  const int nlayers = a.H + 2;
  for (int layer = 0; layer < nlayers; ++layer) {
    const bool last = layer == nlayers - 1;
    const gchar* Wf = uniform_gptr((layer == 0 ? a.W0f : (last ? a.Wof : a.Whf)) + (size_t)(CT * wc) * 256);
    // PReLU on fragment read; layer 0's input is not a pre-activation: slope 1 leaves it as it is, bit for bit
    const float slope = layer == 0 ? 1.f : (layer == 1 ? *a.slope0 : *a.slopeh);
    // this layer streams its own input tile out to HBM while it multiplies: U (layer 0) or pre[layer - 1]
    float* __restrict__ sdst = layer == 0 ? a.U + grow0 * a.K0 : a.pre + (size_t)(layer - 1) * a.pre_stride + grow0 * a.ldp;
    const int sld = layer == 0 ? a.K0 : a.ldp;

    // accumulators start at the bias (layer 0: the row's own row of b0 + C0[t])
    if (layer == 0) {
      int tr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) tr[r] = trow[16 * wr + 4 * lq + r];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = 16 * (CT * wc + ct) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bv = a.B0tab[(size_t)tr[r] * a.ldtab + col];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][ct][r] = bv;
        }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float bv = (last ? a.bo : a.bh)[16 * (CT * wc + ct) + li];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rt][ct][r] = bv;
      }
    }
    if (last) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = 16 * (CT * wc + ct) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) xq[ct][r] = (myusr + r < a.B && col < a.L) ? a.x0[(size_t)(myusr + r) * a.L + col] : 0.f;
      }
    }
    // prologue: K-step 0's fragments
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b0[ct] = gload4(Wf, lane16 + ct * 1024u);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      a0[rt] = *reinterpret_cast<const f32x4*>(abase + rt * RC_USERS * LDA);
#pragma unroll
      for (int k = 0; k < 4; ++k) a0[rt][k] = prelu_any(a0[rt][k], slope);
    }
    // the tile stream: 3 CT chunks over the CT pairs of K-steps (two chunks, then one)
    gchar* sdw = uniform_gptr(sdst);
    gchar* adw = uniform_gptr(layer == 0 ? nullptr : a.act + (size_t)(layer - 1) * a.pre_stride + grow0 * a.ldp);
    RcSweep<QP, LDA> sw;
    sw.init(tid, sld);
    auto kloop = [&](auto act_tag) {
      constexpr bool ACT = decltype(act_tag)::value;
      for (int ks = 0; ks < KS; ks += 2) {
        const int k2 = ks + 2 < KS ? ks + 2 : ks;   // past the end: a harmless re-read
        rc_kstep<CT, LDA, 2, ACT>(acc, a0, b0, a1, b1, Wf + (size_t)(ks + 1) * (NCT * 1024), lane16, aoff + 64u * (ks + 1), slope, Act, sdw, adw, sw);
        rc_kstep<CT, LDA, 1, ACT>(acc, a1, b1, a0, b0, Wf + (size_t)k2 * (NCT * 1024), lane16, aoff + 64u * k2, slope, Act, sdw, adw, sw);
      }
    };
    if (layer == 0 || a.act == nullptr) kloop(std::false_type{});
    else kloop(std::true_type{});
    if (layer < 3) RC_STAMP(2 + 3 * layer);
    if (last) break;

    // in-place epilogue: every wave is done reading the tile, then it takes this layer's outputs (pre-activations)
    __syncthreads();
    if (layer < 3) RC_STAMP(3 + 3 * layer);
    float* __restrict__ obase = Act + (16 * wr + 4 * lq) * LDA + 16 * CT * wc + li;   // + (32 * rt + r) * LDA + 16 * ct
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) obase[(RC_USERS * rt + r) * LDA + 16 * ct] = acc[rt][ct][r];
    __syncthreads();
    if (layer < 3) RC_STAMP(4 + 3 * layer);
  }

  // ---------------------------------------------------------------- out layer: tanh, Y, loss partial sums (:196-198), from registers
  float* __restrict__ ydst = a.Y + (grow0 + 16 * wr + 4 * lq) * a.ldy + 16 * CT * wc + li;   // + (32 * rt + r) * ldy + 16 * ct
  double sD = 0, sC = 0, sR = 0, sR2 = 0;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = 16 * (CT * wc + ct) + li;
    float fD = 0.f, fC = 0.f, fR = 0.f, fR2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float P = tanh_fast(acc[0][ct][r]), S = tanh_fast(acc[1][ct][r]), Q = tanh_fast(acc[2][ct][r]);
      ydst[(size_t)r * a.ldy + 16 * ct] = P;
      ydst[(size_t)(RC_USERS + r) * a.ldy + 16 * ct] = S;
      ydst[(size_t)(2 * RC_USERS + r) * a.ldy + 16 * ct] = Q;
      if (myusr + r < a.B && col < a.L) {
        const float R = P - xq[ct][r];
        const float D = (Q - S) * (1.f / MU2) - R;   // a multiply: an IEEE division is ten instructions, 132 times per lane
        const float RS = R - S;
        fD += D * D; fC += RS * RS; fR += R; fR2 += R * R;
      }
    }
    sD += fD; sC += fC; sR += fR; sR2 += fR2;
  }
  RC_STAMP(9);
  double tot[4] = {sD, sC, sR, sR2};
  block_sum4(tot, red);
  if (tid == 0) {
    double* o = a.loss_part + 4 * (size_t)g;
    o[0] = tot[0]; o[1] = tot[1]; o[2] = tot[2]; o[3] = tot[3];
  }
#ifdef RC_STAMPS
  RC_STAMP(10);
  if (tid == 0 && a.stamps) {
    st_[12] = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 16; ++i) a.stamps[16 * (size_t)g + i] = st_[i];
  }
#endif
}

}  // namespace sdrm
