// Row-owned train forward for mid-width eps-nets (64 < padded width <= 352, L == W) on gfx950: the north-star's
// "persistent-threadblock MLP with LDS-resident activations", at TRAIN size.
//
// Ownership: ONE work-group per CU owns 96 stacked rows - the P, S and Q rows of 32 users (train_SDRM.py:331-333: the three
// forwards of a train step) - through staging (q_sample + three dropout masks, :326-331 / :100) and ALL H+2 layers (:97-103).
//   * the 96 x NP activation tile lives in LDS (135 KB at NP = 352) and is overwritten in place after each layer's barrier;
//     it holds the layer INPUT exactly as the MFMAs want it and as the weight gradients read it later (dropped-out latents
//     for layer 0, ACTIVATIONS above: PReLU is applied ONCE per element, in the epilogue that writes the tile), so the same
//     tile is streamed out to HBM - U, act[k], coalesced 16-byte stores - a few rows per K-step while the next layer
//     multiplies out of it; the pre-activations the dgrads want go to HBM straight from the accumulators (the MFMA operands
//     are swapped, the tile comes out transposed: a lane holds four consecutive COLUMNS of one row, a 16-byte store);
//   * the weights never touch LDS: every wave fetches its MFMA B fragments straight from L2 out of a FRAGMENT-PACKED copy
//     ([k-step][column tile][lane][4 floats]: one contiguous 1 KiB wave-load per 16x16 tile and 16-deep K-step; k_adam
//     writes these copies beside the padded ones), double-buffered in registers one whole K-step (132 MFMAs = 4224 cycles)
//     ahead, by raw buffer loads (resource + lane offset + scalar offset: no VALU address arithmetic);
//   * NO VALU instruction in a K-step except the LDS address of the two stream chunks: fp32 MFMA runs on the vector lanes,
//     a VALU instruction beside it is paid in full (tools/mfma_shadow_asm_probe.py), memory instructions are not;
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

#include "bufres.h"
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
  const float* tembP;   // [T+1][K0 - LPs]: the time-embedding table, rows padded with zeros
  int B, L, T, H;
  int mode; uint32_t seed_lo, seed_hi, step; int64_t row0; float nd;
  // net: fragment-packed weights (layer 0: the latent columns only), biases, per-timestep bias table of layer 0
  const float* W0f; const float* Whf; const float* Wof;
  const float* bh; const float* bo; const float* B0tab; int ldtab;
  const float* slope0; const float* slopeh;
  // outputs, grouped stacked rows
  float* U; int K0, LPs; int* tdev;
  int ones_col;                             // pad column of U set to 1.0 in every row (-1: none), see wgrad2.h
  int light;                                // the weight copies' last K-step is compact (wfrag_index): issue a quarter of its MFMAs
  float* pre; size_t pre_stride; int ldp;   // pre[k] = pre + k * pre_stride, [MP][ldp]
  float* act;                               // activations prelu(pre[k]) in the same layout (null: not stored)
  float* Y; int ldy;
  double* loss_part;                        // [gridDim.x][4]
  unsigned long long* stamps;               // diagnostic builds only (-DRC_STAMPS): 16 s_memtime slots per work-group
  // column-split row groups only (rows48.h): the groups' hand-shake counters, and the number of row groups
  unsigned* xcnt; unsigned xbase; unsigned* xabort; int ngroups;
  int skip_pre;   // the backward of this step reads a layer's ACTIVATIONS where it used to read pre-activations (dgrad_rows.h: PReLU' and
                  // the slope gradient follow from prelu(v) when the slope is positive), so the forward need not store pre[k]:
                  // 35 MB per layer at B = 8192 that every epilogue burst out at once; a slope <= 0 keeps the stores (decided on the device)
  int sweep;   // rows48.h, shared-tile form: a layer's outputs leave for HBM behind the next layer's MFMAs (R48Sweep); 0: stored by its epilogue
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
// (LDS) land in (an, bn), and two chunks of the activation tile are streamed LDS -> HBM
// diagnostic builds only (-DRC_DIAG=mask): drop a piece of the main loop to see what it costs - bit0 the tile stream, bit1 the B
// fragment loads, bit2 the A fragment reads, bit3 the pre-activation stores of the epilogues
#ifndef RC_DIAG
#define RC_DIAG 0
#endif
#ifndef RC_PIECE_STRIDE   // MFMAs between two pipeline pieces of a K-step (the pieces sit at its front)
#define RC_PIECE_STRIDE 3
#endif

// fp32 MFMA runs on the SIMD's fp32 vector lanes (that is why its peak equals the vector peak): a VALU instruction of the same
// wave is NOT hidden behind it (tools/mfma_filler_probe.hip, tools/mfma_shadow_asm_probe.py: 4 - 9 cycles each), LDS reads,
// global loads and stores cost nothing - unless their ADDRESS is a VALU result of the same K-step: the in-order wave then
// stalls for the add and for the load that waits for it (~32 cycles per load, csrc/dgrad_rows.h).  So the K-step has no VALU
// work: weights by raw buffer loads (scalar offsets), the tile stream by buffer stores, PReLU in the epilogue.
//
// PReLU, the generic form: max(v, 0) + slope * min(v, 0), three instructions without a compare.
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

// The sweep of the activation tile (LDS -> HBM, one float4 per thread and chunk): thread -> (row tid / 8, quad tid % 8) of a
// 32-row x 8-quad block, chunk c = block (c / CT, c % CT) - every chunk is the same lane pattern at a UNIFORM offset, so the
// stepping lives on the scalar unit (and it is the pattern the staging writes the tile in)
struct RcStream {
  uint32_t lds0, g0;     // this lane's byte offsets in chunk 0: LDS tile / HBM rows
  uint32_t sl, sg;       // uniform byte offsets of the current chunk
  int jj;                // its column block
  uint32_t gwrap;        // HBM step from the last column block of a row block to the first of the next
  template <int CT, int LDA>
  __device__ __forceinline__ void init(int tid, int sld) {
    lds0 = (uint32_t)(((tid >> 3) * LDA + 4 * (tid & 7)) * 4);
    g0 = (uint32_t)(((tid >> 3) * sld + 4 * (tid & 7)) * 4);
    sl = 0u; sg = 0u; jj = 0;
    gwrap = (uint32_t)(RC_USERS * sld * 4 - (CT - 1) * 128);
  }
  template <int CT, int LDA>
  __device__ __forceinline__ void next() {
    const bool wrap = jj == CT - 1;
    jj = wrap ? 0 : jj + 1;
    sl += wrap ? (uint32_t)(RC_USERS * LDA * 4 - (CT - 1) * 128) : 128u;
    sg += wrap ? gwrap : 128u;
  }
};

// before the first K-step of a layer: every accumulator's initial value is materialised in ITS register here (an opaque
// read-modify of each: the compiler can no longer write it lazily, right in front of the first asm MFMA that reads it), then the
// wait states a VALU write needs before an MFMA may read it
template <int CT>
__device__ __forceinline__ void rc_acc_begin(f32x4 (&acc)[3][CT]) {
#pragma unroll
  for (int rt = 0; rt < 3; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+a"(acc[rt][ct]));
  asm volatile("s_nop 7");
}
// after the last K-step: the wait states an MFMA result needs before a VALU instruction may read it, as a dependence of every
// accumulator: volatile asm statements keep their order, so every read follows the s_nops
template <int CT>
__device__ __forceinline__ void rc_acc_settle(f32x4 (&acc)[3][CT]) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][0]));
#pragma unroll
  for (int rt = 0; rt < 3; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      if (rt + ct > 0) asm volatile("" : "+a"(acc[rt][ct]));
}

// One K-step of a wave: 12 * CT MFMAs out of (ac, bc), and between them the pieces that prepare the next step: its B fragments
// (CT raw buffer wave-loads of 1 KiB), its A fragments (3 ds_read_b128) and NS chunks of the tile stream LDS -> HBM (read, then
// NS pieces later the store).  The body must stay ONE basic block with every piece where it is written: a branch inside it, or a
// load the compiler is free to hoist (loads of read-only memory are not ordered against sched_barrier), and all pieces end up in
// front of the MFMAs.  Hence compile-time piece counts, and every access's offset is passed through an empty asm volatile at its
// slot, which pins it there.  STREAM = false: the tile is not stored (a layer whose input nobody reads again).
// The MFMA operands are SWAPPED (weights as srcA, activations as srcB): the 16 x 16 tile comes out transposed, lane (li, lq)
// holds row li, columns 4 lq .. 4 lq + 3 of it - every epilogue access is a 16-byte one.
// MODE (the compact last K-step of a layer, elementwise.h: wfrag_index): RC_NEXT_LIGHT - the A pieces fetch the NEXT K-step's
// compact fragments (one float per lane: k = 16 ks + lane group; `anext` is that address); RC_LIGHT - this IS the compact K-step:
// only the first MFMA of every tile is issued (its four lane groups hold the four real k), and nothing is fetched behind it.
enum { RC_PLAIN = 0, RC_NEXT_LIGHT = 1, RC_LIGHT = 2 };
// RTS: rows between two of the wave's row tiles in the LDS tile (the users of a group: 32 here, 16 in rows48.h)
template <int CT, int LDA, int NS, bool STREAM, bool NT, int MODE = RC_PLAIN, int RTS = RC_USERS>
__device__ __forceinline__ void rc_kstep(f32x4 (&acc)[3][CT], const f32x4 (&ac)[3], const f32x4 (&bc)[CT], f32x4 (&an)[3],
                                         f32x4 (&bn)[CT], brsrc wres, uint32_t wnext, uint32_t lane16, uint32_t anext,
                                         const float* __restrict__ Act, brsrc sres, RcStream& sw) {
  constexpr int NE = MODE == RC_LIGHT ? 1 : 4;
  constexpr int NSLOT = 3 * NE * CT;
  constexpr int P_A = MODE == RC_LIGHT ? 0 : CT, P_S = MODE == RC_LIGHT ? 0 : CT + 3, NPIECE = P_S + (STREAM ? 2 * NS : 0);
  constexpr int STRIDE = NPIECE == 0 ? NSLOT : (NSLOT / NPIECE >= RC_PIECE_STRIDE ? RC_PIECE_STRIDE : (NSLOT / NPIECE >= 1 ? NSLOT / NPIECE : 1));
  static_assert(NPIECE <= NSLOT && NS <= 2, "not enough MFMA slots for the pipeline pieces");
  f32x4 sv0 = {0.f, 0.f, 0.f, 0.f}, sv1 = sv0;
  uint32_t so0 = 0, so1 = 0;
#pragma unroll
  for (int e = 0; e < NE; ++e)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int s = (e * CT + ct) * 3 + rt;
    // asm with the accumulator tied in place: through the builtin the register allocator renames accumulators inside the loop
    // body and copies them back (v_accvgpr_mov by the hundred).  The price: the compiler's hazard recogniser does not see an
    // MFMA here - rc_acc_begin / rc_acc_settle guard the two places where compiler-made VALU code meets the accumulators
    // (their initial values, their first read); tests/test_isa_lint.py checks the generated code for any other
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[rt][ct]) : "v"(bc[ct][e]), "v"(ac[rt][e]));
    if (s % STRIDE == 0 && s / STRIDE < NPIECE) {
      const int p = s / STRIDE;
      if (p < P_A) {
        if (!(RC_DIAG & 2)) {
          uint32_t so = wnext + (p / 4) * 4096;   // opaque at this slot: pins the load here
          asm volatile("" : "+s"(so));
          bn[p] = bload4(wres, lane16 + (p % 4) * 1024, so);
        }
      } else if (p < P_S) {
        if (!(RC_DIAG & 4)) {
          uint32_t ao = anext;
          asm volatile("" : "+v"(ao));
          if (MODE == RC_NEXT_LIGHT) an[p - P_A][0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Act) + ao + (p - P_A) * RTS * LDA * 4);
          else an[p - P_A] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Act) + ao + (p - P_A) * RTS * LDA * 4);
        }
      } else {
        // tile stream, chunk q: LDS read, then (NS pieces later) the HBM store
        constexpr int NSD = NS > 0 ? NS : 1;   // (NS = 0: no stream pieces; the branch is dead but compiled)
        const int k = p - P_S, ph = k / NSD, q = k % NSD;
        if (!(RC_DIAG & 1)) {
          f32x4& v = q == 0 ? sv0 : sv1;
          uint32_t& so = q == 0 ? so0 : so1;
          if (ph == 0) {
            uint32_t lo = sw.lds0 + sw.sl;
            asm volatile("" : "+v"(lo));
            v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Act) + lo);
            so = sw.sg;
            sw.template next<CT, LDA>();
          } else {
            asm volatile("" : "+s"(so));
            bstore4<NT>(sres, sw.g0, so, v);
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

// LIGHT: the weight copies' last K-step is compact (a.light; the host picks the instantiation)
template <int CT, bool LIGHT = false>
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
    // the time-embedding columns of U, temb[t] of the row's timestep (read by the layer-0 weight gradient only: they deliver
    // M = dpre0^T * temb, tail.h; rows of users beyond the batch stay all-zero): the P, S and Q row of this thread's user, column
    // quads sq + 8 j - issued first, they drain while the randoms are drawn
    const int TPc = a.K0 - a.LPs;          // a multiple of 32 columns
    const int TQ8 = TPc >> 5;              // groups of eight quads
    const bool uin = susr < a.B;
    const float* trow_p = a.tembP + (size_t)(uin ? trow[su] : 0) * TPc + 4 * sq;
    const brsrc ures = make_brsrc(a.U + grow0 * a.K0, (uint32_t)(RC_ROWS * a.K0 * 4));
    const uint32_t uvo = (uint32_t)((su * a.K0 + a.LPs + 4 * sq) * 4);
    for (int jq = 0; jq < TQ8; ++jq) {
      const float4 te = uin ? *reinterpret_cast<const float4*>(trow_p + 32 * jq) : make_float4(0.f, 0.f, 0.f, 0.f);
      const f32x4 oh = {te.x, te.y, te.z, te.w};
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) bstore4<true>(ures, uvo, (uint32_t)((pass * RC_USERS * a.K0 + 32 * jq) * 4), oh);
    }
  }
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
  __syncthreads();
  RC_STAMP(1);

  // ---------------------------------------------------------------- layers
  // wave (wr, wc): row tile rt = pass rt of users 16 wr .. 16 wr + 15 (tile rows 32 rt + 16 wr ..), column tiles CT wc ..;
  // lane (li, lq) holds of tile (rt, ct) row 32 rt + 16 wr + li, columns 16 (CT wc + ct) + 4 lq .. + 3 (the transposed MFMA tile)
  const float* abase = Act + (16 * wr + li) * LDA + 4 * lq;     // + 32 * rt * LDA + 16 * ks: A fragment reads
  const uint32_t aoff = (uint32_t)(((16 * wr + li) * LDA + 4 * lq) * 4);   // the same as a byte offset into the tile
  const uint32_t aoffl = (uint32_t)(((16 * wr + li) * LDA + lq) * 4);      // ... of the compact K-step's fragment (k = 16 ks + lq)
  const uint32_t lane16 = 16u * (uint32_t)lane;   // byte offset of a lane's float4 in a 1 KiB wave-load
  const int myrow = 16 * wr + li;                 // + 32 rt: the lane's row of the tile; its user is u0 + myrow
  const int mycol = 16 * CT * wc + 4 * lq;        // + 16 ct: the first of its four columns
  float* __restrict__ otile = Act + myrow * LDA + mycol;   // + 32 rt * LDA + 16 ct: the lane's quad of the tile
  f32x4 acc[RT][CT];
  f32x4 b0[CT], b1[CT];
  f32x4 a0[RT], a1[RT];
  f32x4 xq[CT];   // x0 at this lane's accumulator positions (loss sums), requested before the out layer's loop

  const int nlayers = a.H + 2;
  for (int layer = 0; layer < nlayers; ++layer) {
    const bool last = layer == nlayers - 1;
    const brsrc Wf = make_brsrc((layer == 0 ? a.W0f : (last ? a.Wof : a.Whf)) + (size_t)(CT * wc) * 256, (uint32_t)((KS * NCT - CT * wc) * 1024));
    // this layer streams its own input tile out to HBM while it multiplies: U (layer 0) or the activations act[layer - 1]
    // (nobody reads them when the weight gradients take PReLU(pre) themselves: a.act == null)
    const bool stream = layer == 0 || a.act != nullptr;
    float* sdst = layer == 0 ? a.U + grow0 * a.K0 : a.act + (size_t)(layer - 1) * a.pre_stride + grow0 * a.ldp;
    const int sld = layer == 0 ? a.K0 : a.ldp;
    const brsrc sres = make_brsrc(stream ? sdst : a.U, stream ? (uint32_t)(RC_ROWS * sld * 4) : 0u);

    // accumulators start at the bias (layer 0: the row's own row of b0 + C0[t])
    {
      const float* bsrc = layer == 0 ? a.B0tab + (size_t)trow[myrow] * a.ldtab : (last ? a.bo : a.bh);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 bv = *reinterpret_cast<const float4*>(bsrc + mycol + 16 * ct);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = f32x4{bv.x, bv.y, bv.z, bv.w};
      }
    }
    if (last) {
      const int usr = u0 + myrow;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = mycol + 16 * ct;
        const float4 x = (usr < a.B && col < a.L) ? load4_unpadded(a.x0, usr, col, a.L) : make_float4(0.f, 0.f, 0.f, 0.f);
        xq[ct] = f32x4{x.x, x.y, x.z, x.w};
      }
    }
    // prologue: K-step 0's fragments
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b0[ct] = bload4(Wf, lane16 + ct * 1024u, 0u);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a0[rt] = *reinterpret_cast<const f32x4*>(abase + rt * RC_USERS * LDA);
    // the tile stream: 3 CT chunks over the CT pairs of K-steps (two chunks, then one)
    RcStream sw;
    sw.template init<CT, LDA>(tid, sld);
    auto kloop = [&](auto stream_tag, auto nt_tag) {
      constexpr bool STREAM = decltype(stream_tag)::value, NT = decltype(nt_tag)::value;
      constexpr uint32_t WS = NCT * 1024;
      for (uint32_t ks = 0; ks < (uint32_t)KS - 2; ks += 2) {
        rc_kstep<CT, LDA, 2, STREAM, NT>(acc, a0, b0, a1, b1, Wf, (ks + 1) * WS, lane16, aoff + 64u * (ks + 1), Act, sres, sw);
        rc_kstep<CT, LDA, 1, STREAM, NT>(acc, a1, b1, a0, b0, Wf, (ks + 2) * WS, lane16, aoff + 64u * (ks + 2), Act, sres, sw);
      }
      // the last pair: K-step KS - 1 may be the compact one (a.light: four real k in its sixteen)
      if constexpr (LIGHT) {
        rc_kstep<CT, LDA, 2, STREAM, NT, RC_NEXT_LIGHT>(acc, a0, b0, a1, b1, Wf, (KS - 1) * WS, lane16, aoffl + 64u * (KS - 1), Act, sres, sw);
        rc_kstep<CT, LDA, 1, STREAM, NT, RC_LIGHT>(acc, a1, b1, a0, b0, Wf, 0u, lane16, aoff, Act, sres, sw);
      } else {
        rc_kstep<CT, LDA, 2, STREAM, NT>(acc, a0, b0, a1, b1, Wf, (KS - 1) * WS, lane16, aoff + 64u * (KS - 1), Act, sres, sw);
        rc_kstep<CT, LDA, 1, STREAM, NT>(acc, a1, b1, a0, b0, Wf, (KS - 2) * WS, lane16, aoff + 64u * (KS - 2), Act, sres, sw);   // past the end: a harmless re-read
      }
    };
    rc_acc_begin<CT>(acc);
    if (stream) kloop(std::true_type{}, std::true_type{});
    else kloop(std::false_type{}, std::false_type{});
    rc_acc_settle<CT>(acc);
    if (layer < 3) RC_STAMP(2 + 3 * layer);
    if (last) break;

    // in-place epilogue: every wave is done reading the tile; then the pre-activations go to HBM as they are (read next by the
    // dgrads) and their PReLU into the tile - the next layer's input, and what the stream stores for the weight gradients
    __syncthreads();
    if (layer < 3) RC_STAMP(3 + 3 * layer);
    {
      const float slope = layer == 0 ? *a.slope0 : *a.slopeh;
      gchar* pw = uniform_gptr(a.pre + (size_t)layer * a.pre_stride + grow0 * a.ldp);
      const uint32_t pbase = (uint32_t)((myrow * a.ldp + mycol) * 4);
      const uint32_t prt = (uint32_t)(RC_USERS * a.ldp * 4);
      if (!(a.skip_pre && slope >= SLOPE_FROM_ACT_MIN)) {   // (uniform) somebody will read the pre-activations themselves
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const f32x4 v = acc[rt][ct];
            if (!(RC_DIAG & 8)) gstore4(pw + ct * 64, pbase + rt * prt, make_float4(v[0], v[1], v[2], v[3]));
          }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const f32x4 v = acc[rt][ct];
          *reinterpret_cast<float4*>(otile + rt * RC_USERS * LDA + 16 * ct) =
              make_float4(prelu_any(v[0], slope), prelu_any(v[1], slope), prelu_any(v[2], slope), prelu_any(v[3], slope));
        }
    }
    __syncthreads();
    if (layer < 3) RC_STAMP(4 + 3 * layer);
  }

  // ---------------------------------------------------------------- out layer: tanh, Y, loss partial sums (:196-198), from registers
  gchar* yw = uniform_gptr(a.Y + grow0 * a.ldy);
  const uint32_t ybase = (uint32_t)((myrow * a.ldy + mycol) * 4), yrt = (uint32_t)(RC_USERS * a.ldy * 4);
  const bool uok = u0 + myrow < a.B;
  double sD = 0, sC = 0, sR = 0, sR2 = 0;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = mycol + 16 * ct;
    float fD = 0.f, fC = 0.f, fR = 0.f, fR2 = 0.f;
    float P[4], S[4], Q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { P[i] = tanh_fast(acc[0][ct][i]); S[i] = tanh_fast(acc[1][ct][i]); Q[i] = tanh_fast(acc[2][ct][i]); }
    gstore4(yw + ct * 64, ybase, make_float4(P[0], P[1], P[2], P[3]));
    gstore4(yw + ct * 64, ybase + yrt, make_float4(S[0], S[1], S[2], S[3]));
    gstore4(yw + ct * 64, ybase + 2 * yrt, make_float4(Q[0], Q[1], Q[2], Q[3]));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (uok && col + i < a.L) {
        const float R = P[i] - xq[ct][i];
        const float D = (Q[i] - S[i]) * (1.f / MU2) - R;   // a multiply: an IEEE division is ten instructions, 132 times per lane
        const float RS = R - S[i];
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
