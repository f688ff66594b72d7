// The row-owned train step for batches that do NOT fill the chip with 96-row work-groups (rowchain.h, dgrad_rows.h): the same
// ownership at half the height.  ONE work-group owns 48 stacked rows - the P, S and Q rows of 16 users (train_SDRM.py:331-333) -
// through staging (q_sample + three dropout masks, :326-331 / :100), ALL H + 2 layers (:97-103) and the loss partial sums
// (:191-199) in the forward, and through the loss value, the gradient seeds and every layer's input gradient in the backward
// (:336).  A batch of 4096 users (the 2-GPU shard of the 8192 batch) is 256 such work-groups - one round of the chip - where the
// 96-row kernels would leave half the CUs idle and the per-layer path runs eleven launches.
//
// What changes against the 96-row kernels:
//   * the four waves stand SIDE BY SIDE along the columns: wave w owns all three row tiles (P, S, Q of the same 16 users: the out
//     layer's loss sums still come straight out of one lane's accumulators) and the column tiles [CW w, CW w + CW), CW =
//     ceil(NCT / 4) of the layer's NCT = NP / 16 tiles (6 at NP = 352, where the last wave has four real tiles: its two tiles
//     beyond the layer are multiplied - operands out of range read as zero - and dropped; the other waves set the time anyway);
//   * every wave fetches DIFFERENT weight fragments (in the 96-row kernel the two row halves fetch the same ones): per CU the same
//     bytes per cycle from L1, twice the bytes per flop from L2 (10.4 B per cycle and CU, inside what every kernel here sustains);
//   * the 48 x NP tile is 69 KB of LDS (two work-groups fit a CU: batches between 4096 and 6912 users run two rounds' worth of
//     work-groups side by side) and is NOT streamed to HBM while a layer multiplies: the layer input (U, then the activations) is
//     stored straight from registers by the staging / the epilogue that produces it - half the rows make half the burst;
//   * stacked row order "grouped by 16" (elementwise.h, stacked_row(2, ..)): row = 48 (user / 16) + 16 pass + user % 16, the order
//     the narrow nets' kernels use.
// K-steps, pipeline pieces, asm MFMAs with tied accumulators, swapped operands (a lane holds four consecutive COLUMNS of a row),
// fragment-packed weights with the compact last K-step: rowchain.h's, through the same rc_kstep / dr_kstep.
//
// COLUMN-SPLIT row groups (G = 2 or 4 work-groups per 48-row group; round 5): below 4096 users even 48-row work-groups leave CUs
// idle - a batch of 1024 users (the 8-GPU shard of the 8192 batch) is 64 groups - and the per-layer path pays eleven launches of
// 6 - 15 us for 40 us of matrix work.  Here G work-groups share a row group: each stages the whole input tile (the randoms are
// counter-based: every one draws the same), owns 1 / G of every layer's output columns (its four waves side by side inside that
// range), stores its columns of the pre-activations and activations, and then PULLS the other work-groups' columns of the
// activations - which the weight gradients want in HBM anyway - into its LDS tile for the next layer.  What makes that cheap is the
// chip's topology: work-group b runs on XCD b & 7 (stamps, profiles/r02_wgrad_one_round_and_strips.txt; checked by sdrm_create with
// a probe launch), so the G work-groups of a group are given block indices on ONE XCD, whose L2 is coherent for its own CUs:
// stores are acknowledged by that L2 (s_waitcnt vmcnt(0)), a counter per group is bumped by a work-group-scope atomic (executed in
// that L2) and polled with sc1 loads, the columns are pulled with sc1 loads (served by the L2, never by the CU's L1) - 1.05 us per
// hand-shake and ~15 B per cycle and CU measured (tools/xcd_local_probe.hip, profiles/r04_xcd_local_probe.txt) against ~4.5 us for
// a kernel boundary.  No device-scope fence, no L2 write-back.  Every work-group of the launch is resident at once (grids of at most
// 256 work-groups, one per CU), every wait is bounded by the wall clock (s_memrealtime) and a timeout raises a flag in host-visible
// memory that the next call on the handle reports (sdrm_hip.hip: split_status).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgrad_rows.h"
#include "rowchain.h"

// diagnostic builds only (-DR48_DIAG=mask, timing: results are wrong): bit0 drops the epilogues' pre-activation stores, bit1 their
// activation stores, bit2 the staging's U stores
#ifndef R48_DIAG
#define R48_DIAG 0
#endif

namespace sdrm {

constexpr int R48_USERS = 16;
constexpr int R48_ROWS = 3 * R48_USERS;

template <int CT, int G = 1>
struct Rows48Cfg {
  static constexpr int NP = 32 * CT, NCT = 2 * CT, KS = NP / 16, QP = NP / 4, LDA = rc_lda(NP);
  static constexpr int NCG = (NCT + G - 1) / G;    // column tiles of one work-group of a group
  static constexpr int CW = (NCG + 3) / 4;         // ... of one of its waves
  static constexpr int NQ = (QP + 15) / 16;        // column quads per staging thread (16 quad lanes per user)
  static constexpr int NPULL = (R48_ROWS * QP + NTHREADS - 1) / NTHREADS;   // quads of the tile per thread in a sweep over it
  static constexpr size_t LDS_BYTES = (size_t)R48_ROWS * LDA * 4 + 512;
  static_assert(G == 1 || G == 2 || G == 4, "work-groups per row group");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two work-groups share a CU's LDS");
};

// The hand-shake of a column-split row group (see the head of this file).  cnt: the group's counter (one per 128-byte line), never
// reset: launch number `epoch` counts from base = epoch * phases * G.
struct SplitSync {
  unsigned* cnt;        // [groups][32]
  unsigned base;        // first target of this launch minus G
  unsigned* abort_;     // host-visible word: != 0 once any wait of any launch timed out
};

constexpr unsigned long long R48_SPIN_LIMIT = 3000000ull;   // s_memrealtime ticks (100 MHz): 30 ms

__device__ __forceinline__ unsigned r48_poll_l2(const unsigned* p) {   // the counter as the XCD's L2 holds it
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// every thread's stores of this phase are acknowledged by the L2, then the work-groups of the group meet; false: timed out
__device__ __forceinline__ bool r48_group_barrier(unsigned* my, unsigned target, unsigned* abort_, int* go) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int ok = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 0;; ++spin) {
      if ((int)(r48_poll_l2(my) - target) >= 0) { ok = 1; break; }
      if ((spin & 63u) == 63u) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > R48_SPIN_LIMIT) break;
        if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
      }
    }
    if (!ok) __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *go = ok;
  }
  __syncthreads();
  return *go != 0;
}

// where work-group b of a launch runs: HW_REG_XCC_ID (id 20) per block; sdrm_create checks "block b on XCD b & 7" with it before the
// column-split path may be taken
__global__ void k_xcc_probe(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11));
}

__device__ __forceinline__ f32x4 r48_load_l2(const float* p) {   // 16 bytes as the XCD's L2 holds them (issued, not waited for)
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// One K-step over a WINDOW of a wave's column tiles (the shared-tile form of k_rows48_fwd, below): MFMAs for the tiles [C0, C1) of
// the window of NW, and in their shadows the next K-step's B fragments for the tiles [L0, L1) (raw buffer wave-loads of 1 KiB) and
// its A fragments (3 ds_read_b128).  rc_kstep (rowchain.h) with tile ranges instead of a tile count and without the tile stream:
// the same slot discipline (one basic block, every piece's offset opaque at its slot), the same MODEs of the compact last K-step.
// The sweep that stores a layer's outputs BEHIND the next layer's MFMAs (shared-tile form): a layer's epilogue leaves its
// activations in the LDS tile (the next layer's input anyway) and its pre-activations in a second LDS image of the same shape;
// each K-step of the next layer then moves ONE 4 KB chunk of either to HBM - two ds_read_b128, two 16-byte buffer stores a few
// MFMAs later - instead of both going out in one burst that the first weight fragment behind it has to wait for (vmcnt counts loads
// and stores in order; measured: the two bursts of a layer cost the B = 4096 forward 9.4 of its 99 us).  Chunk c = the 16-row x
// 16-quad block (c / NCB, c % NCB), thread -> (row tid / 16, quad tid % 16): the same lane pattern at uniform offsets, stepped on
// the scalar unit; lanes beyond the tile's last column quad (the last column block of an odd CT) and every lane of a chunk beyond
// the tile take a byte offset outside the buffer resource - the hardware drops such stores.
struct R48Sweep {
  uint32_t lds0, g0, g0m;   // this lane's byte offsets in chunk 0: LDS image / HBM rows; g0m: the same, out of range where the last column block overhangs
  uint32_t sl, sg;          // uniform byte offsets of the current chunk
  int cb, left;             // its column block; chunks still inside the tile
  uint32_t lwrap, gwrap;    // steps from a row block's last column block to the next row block's first
  template <int QP, int LDA>
  __device__ __forceinline__ void init(int tid, int ld) {
    constexpr int NCB = (QP + 15) / 16;
    const int r16 = tid >> 4, q16 = tid & 15;
    lds0 = (uint32_t)((r16 * LDA + 4 * q16) * 4);
    g0 = (uint32_t)((r16 * ld + 4 * q16) * 4);
    g0m = (16 * (NCB - 1) + q16 < QP) ? g0 : 0x7ffffff0u;
    sl = 0u; sg = 0u; cb = 0; left = 3 * NCB;
    lwrap = (uint32_t)(16 * LDA * 4 - (NCB - 1) * 256);
    gwrap = (uint32_t)(16 * ld * 4 - (NCB - 1) * 256);
  }
  // the current chunk's per-lane HBM offset (out of range: dropped), then on to the next chunk
  template <int QP>
  __device__ __forceinline__ uint32_t voff_and_next() {
    constexpr int NCB = (QP + 15) / 16;
    const bool lastcb = cb == NCB - 1;
    const uint32_t vo = left > 0 ? (lastcb ? g0m : g0) : 0x7ffffff0u;
    cb = lastcb ? 0 : cb + 1;
    sl += lastcb ? lwrap : 256u;
    sg += lastcb ? gwrap : 256u;
    --left;
    return vo;
  }
};

template <int NW, int C0, int C1, int L0, int L1, int LDA, int MODE = RC_PLAIN, bool STREAM = false, int QP = 0>
__device__ __forceinline__ void r48_kstep(f32x4 (&acc)[3][NW], const f32x4 (&ac)[3], const f32x4 (&bc)[NW], f32x4 (&an)[3], f32x4 (&bn)[NW],
                                          brsrc wres, uint32_t wnext, uint32_t lane16, uint32_t anext, const float* __restrict__ Act,
                                          R48Sweep* sw, brsrc ares, brsrc pres, uint32_t stg) {
  constexpr int NE = MODE == RC_LIGHT ? 1 : 4;
  constexpr int NT_ = C1 - C0, NSLOT = 3 * NE * NT_;
  constexpr bool ST = STREAM && MODE != RC_LIGHT;
  constexpr int P_A = MODE == RC_LIGHT ? 0 : L1 - L0, P_S = MODE == RC_LIGHT ? 0 : P_A + 3, NPIECE = P_S + (ST ? 4 : 0);
  constexpr int STRIDE = NPIECE == 0 ? NSLOT : (NSLOT / NPIECE >= RC_PIECE_STRIDE ? RC_PIECE_STRIDE : (NSLOT / NPIECE >= 1 ? NSLOT / NPIECE : 1));
  static_assert(0 <= C0 && C0 < C1 && C1 <= NW && 0 <= L0 && L0 <= L1 && L1 <= NW && NPIECE <= NSLOT, "tile ranges / pipeline pieces");
  f32x4 sva = {0.f, 0.f, 0.f, 0.f}, svp = sva;
  uint32_t sso = 0u, svo = 0u;
#pragma unroll
  for (int e = 0; e < NE; ++e)
#pragma unroll
  for (int ct = C0; ct < C1; ++ct)
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int s = (e * NT_ + (ct - C0)) * 3 + rt;
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[rt][ct]) : "v"(bc[ct][e]), "v"(ac[rt][e]));
    if (s % STRIDE == 0 && s / STRIDE < NPIECE) {
      const int p = s / STRIDE;
      if (p < P_A) {
        const int tl = L0 + p;                          // tile of the window
        uint32_t so = wnext + (tl / 4) * 4096;          // opaque at this slot: pins the load here
        asm volatile("" : "+s"(so));
        bn[tl] = bload4(wres, lane16 + (tl % 4) * 1024, so);
      } else if (p < P_S) {
        uint32_t ao = anext;
        asm volatile("" : "+v"(ao));
        if (MODE == RC_NEXT_LIGHT) an[p - P_A][0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Act) + ao + (p - P_A) * R48_USERS * LDA * 4);
        else an[p - P_A] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Act) + ao + (p - P_A) * R48_USERS * LDA * 4);
      } else if constexpr (ST) {
        // the sweep: chunk reads (activations out of the tile, pre-activations out of the image `stg` bytes behind it), then the stores
        const int k = p - P_S;
        if (k == 0) {
          uint32_t lo = sw->lds0 + sw->sl;
          asm volatile("" : "+v"(lo));
          sva = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Act) + lo);
        } else if (k == 1) {
          uint32_t lo = sw->lds0 + sw->sl + stg;
          asm volatile("" : "+v"(lo));
          svp = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Act) + lo);
          sso = sw->sg;
          svo = sw->template voff_and_next<QP>();
        } else if (k == 2) {
          asm volatile("" : "+s"(sso));
          bstore4<true>(ares, svo, sso, sva);    // (nobody reads the activations before the weight gradients: non-temporal)
        } else {
          asm volatile("" : "+s"(sso));
          bstore4<false>(pres, svo, sso, svp);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LIGHT: the weight copies' last K-step is compact (a.light; the host picks the instantiation).  Arguments: rowchain.h's.
// G: work-groups per row group (see the head of this file); grid: G = 1 one work-group per group; G > 1: 8 * G * ceil(groups / 8)
// work-groups, block b = XCD x = b & 7, slot b >> 3 = (group x + 8 * (slot / G), part slot % G).
// SHARE (G == 1, NCT = 4 q + 2 column tiles - NP = 352: 22): no wave multiplies tiles beyond the layer.  Wave w owns q tiles of
// its own and HALF of one more: the two waves of a pair (0, 1) / (2, 3) have windows of q + 1 consecutive tiles that overlap in one
// tile, which the lower wave multiplies on the even K-steps and the upper wave on the odd ones (B fragments are fetched only for
// the K-steps a wave multiplies); behind the K loop the upper wave hands its partial sums of that tile (3 accumulator quads) to the
// lower wave through LDS, which owns the tile's epilogue.  5.5 tiles per wave and K-step instead of 6: the 8 % of the MFMAs that the
// plain form spends on two tiles of zeros.
template <int CT, bool LIGHT = false, int G = 1, bool SHARE = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_rows48_fwd(const RowChainArgs a) {
  typedef Rows48Cfg<CT, G> C;
  constexpr int NP = C::NP, NCT = C::NCT, NCG = C::NCG, KS = C::KS, QP = C::QP, LDA = C::LDA, NQ = C::NQ, RT = 3;
  constexpr int Q4 = NCT / 4;
  constexpr int CW = SHARE ? Q4 + 1 : C::CW;               // tiles a wave holds accumulators for (SHARE: its window)
  constexpr bool ALLV = SHARE || (4 * CW == NCG && NCG * G == NCT);   // every tile of every wave is a real one
  static_assert(KS % 2 == 0, "K-steps are taken in pairs");
  static_assert(!SHARE || (G == 1 && NCT % 4 == 2), "the shared-tile form: one work-group per row group, 4 q + 2 column tiles");
  __shared__ f32x4 xsh[SHARE ? 2 * 3 * 64 : 1];            // SHARE: the upper waves' partial sums of the pairs' shared tiles
  // (SHARE: a second image of the tile's shape behind it takes a layer's pre-activations until the next layer's K-steps have
  // moved them to HBM - R48Sweep)
  __shared__ __attribute__((aligned(16))) float Act[(SHARE ? 2 : 1) * R48_ROWS * LDA];
  constexpr uint32_t STG = (uint32_t)(R48_ROWS * LDA * 4);
  __shared__ int trow[R48_USERS];
  __shared__ double red[16];
  __shared__ int go;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave = its column block inside the work-group's range
  const int li = lane & 15, lq = lane >> 4;
  int g = blockIdx.x, part = 0;
  if constexpr (G > 1) {
    const int slot = (int)blockIdx.x >> 3;
    part = slot % G;
    g = ((int)blockIdx.x & 7) + 8 * (slot / G);
    if (g >= a.ngroups) return;   // (the whole group: its work-groups share g)
  }
  const int u0 = R48_USERS * g;
  const size_t grow0 = (size_t)R48_ROWS * g;   // first stacked row of this work-group's group
  const int t0 = part * NCG, t1 = min(t0 + NCG, NCT);   // this work-group's column tiles
  unsigned* const xmy = G > 1 ? a.xcnt + 32 * (size_t)g : nullptr;
  unsigned xphase = 0;

  // ---------------------------------------------------------------- staging
  if (tid < R48_USERS) {
    const int usr = u0 + tid;
    int ts = 0;
    if (usr < a.B) {
      if (a.mode == 0) {
        ts = (int)a.t[usr];
      } else {
        const U4 w = philox4x32_10((uint32_t)(a.row0 + usr), 0u, PURPOSE_TRAIN_T, a.step, a.seed_lo, a.seed_hi);
        ts = 1 + (int)bounded(w.x, (uint32_t)a.T);
      }
      ts = min(max(ts, 0), a.T);
      if (part == 0) a.tdev[usr] = ts;
    }
    trow[tid] = ts;
  }
  // thread -> (user tid >> 4, column quads (tid & 15) + 16 j): every x0 quad of the thread is requested before the first is used
  const int su = tid >> 4, sq = tid & 15;
  const int susr = u0 + su;
  float4 xs[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (sq + 16 * j);
    xs[j] = (susr < a.B && c < a.L) ? load4_unpadded(a.x0, susr, c, a.L) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // (the layer-0 operand of the weight gradients is written by ONE work-group of the group: a zero-sized resource drops the others' stores)
  const brsrc ures = make_brsrc(a.U + grow0 * a.K0, part == 0 ? (uint32_t)(R48_ROWS * a.K0 * 4) : 0u);
  {
    // the time-embedding columns of U, temb[t] of the row's timestep (read by the layer-0 weight gradient only: they deliver
    // M = dpre0^T * temb, tail.h; rows of users beyond the batch stay all-zero): the P, S and Q row of this thread's user
    const int TPc = a.K0 - a.LPs;          // a multiple of 32 columns
    const bool uin = susr < a.B;
    const float* trow_p = a.tembP + (size_t)(uin ? trow[su] : 0) * TPc;
    for (int c = 4 * sq; c < TPc; c += 64) {
      const float4 te = uin ? *reinterpret_cast<const float4*>(trow_p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const f32x4 oh = {te.x, te.y, te.z, te.w};
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
        bstore4<true>(ures, (uint32_t)(((pass * R48_USERS + su) * a.K0 + a.LPs + c) * 4), 0u, oh);
    }
  }
  {
    const int tt = trow[su];
    const float sa = a.sqrt_ab[tt], om = a.one_minus_ab[tt];
    // PHILOX mode: the thread's NQ calls (one per column quad: two normal pairs and, in the low bits of word j, the three keep
    // bits of column j) in ONE straight-line block (rowchain.h)
    U4 rw[NQ];
    if (a.mode != 0) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        rw[j] = philox4x32_10((uint32_t)(a.row0 + susr), (uint32_t)(sq + 16 * j), PURPOSE_TRAIN_ELEM, a.step, a.seed_lo, a.seed_hi);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (sq + 16 * j);
      if (c >= NP) continue;   // (the sixteen quad lanes overhang the tile's last columns)
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
      const f32x4 fP = {vP[0], vP[1], vP[2], vP[3]}, fS = {vS[0], vS[1], vS[2], vS[3]}, fQ = {vQ[0], vQ[1], vQ[2], vQ[3]};
      *reinterpret_cast<f32x4*>(Act + su * LDA + c) = fP;
      *reinterpret_cast<f32x4*>(Act + (R48_USERS + su) * LDA + c) = fS;
      *reinterpret_cast<f32x4*>(Act + (2 * R48_USERS + su) * LDA + c) = fQ;
      // the layer-0 operand of the weight gradients, from the same registers (nobody reads it before them: non-temporal)
      const uint32_t uo = (uint32_t)((su * a.K0 + c) * 4), up = (uint32_t)(R48_USERS * a.K0 * 4);
      if (!(R48_DIAG & 4)) {
        bstore4<true>(ures, uo, 0u, fP);
        bstore4<true>(ures, uo + up, 0u, fS);
        bstore4<true>(ures, uo + 2 * up, 0u, fQ);
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- layers
  // wave wc: row tile rt = pass rt of the 16 users (tile rows 16 rt ..), column tiles CW wc ..; lane (li, lq) holds of tile (rt, ct)
  // row 16 rt + li, columns 16 (CW wc + ct) + 4 lq .. + 3 (the transposed MFMA tile)
  const float* abase = Act + li * LDA + 4 * lq;                  // + 16 rt LDA + 16 ks: A fragment reads
  const uint32_t aoff = (uint32_t)((li * LDA + 4 * lq) * 4);     // the same as a byte offset into the tile
  const uint32_t aoffl = (uint32_t)((li * LDA + lq) * 4);        // ... of the compact K-step's fragment (k = 16 ks + lq)
  const uint32_t lane16 = 16u * (uint32_t)lane;
  const int myrow = li;                            // + 16 rt: the lane's row of the tile; its user is u0 + li
  const bool upper = SHARE && (wc & 1);            // SHARE: the wave whose window STARTS with the pair's shared tile
  const int wt0 = SHARE ? (wc >> 1) * (2 * Q4 + 1) + (upper ? Q4 : 0) : t0 + CW * wc;   // the wave's first column tile
  const int mycol = 16 * wt0 + 4 * lq;             // + 16 ct: the first of its four columns
  float* __restrict__ otile = Act + myrow * LDA + mycol;
  f32x4 acc[RT][CW];
  f32x4 b0[CW], b1[CW];
  f32x4 a0[RT], a1[RT];
  f32x4 xq[CW];   // x0 at this lane's accumulator positions (loss sums), requested before the out layer's loop
  RcStream sw{};  // (no tile stream here)
  const brsrc nores = make_brsrc(a.U, 0u);

  const int nlayers = a.H + 2;
  for (int layer = 0; layer < nlayers; ++layer) {
    const bool last = layer == nlayers - 1;
    const brsrc Wf = make_brsrc((layer == 0 ? a.W0f : (last ? a.Wof : a.Whf)) + (size_t)wt0 * 256, (uint32_t)((KS * NCT - wt0) * 1024));
    // accumulators start at the bias (layer 0: the row's own row of b0 + C0[t]); tiles beyond the layer read the slack behind it
    {
      const float* bsrc = layer == 0 ? a.B0tab + (size_t)trow[myrow] * a.ldtab : (last ? a.bo : a.bh);
#pragma unroll
      for (int ct = 0; ct < CW; ++ct) {
        const bool tv = SHARE ? !(upper && ct == 0) : (ALLV || wt0 + ct < t1);   // (SHARE: the upper wave's partial sums start at zero)
        const float4 bv = tv ? *reinterpret_cast<const float4*>(bsrc + mycol + 16 * ct) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = f32x4{bv.x, bv.y, bv.z, bv.w};
      }
    }
    if (last) {
      const int usr = u0 + myrow;
#pragma unroll
      for (int ct = 0; ct < CW; ++ct) {
        const int col = mycol + 16 * ct;
        const float4 x = (usr < a.B && col < a.L) ? load4_unpadded(a.x0, usr, col, a.L) : make_float4(0.f, 0.f, 0.f, 0.f);
        xq[ct] = f32x4{x.x, x.y, x.z, x.w};
      }
    }
    // prologue: K-step 0's fragments
#pragma unroll
    for (int ct = 0; ct < CW; ++ct) b0[ct] = bload4(Wf, lane16 + ct * 1024u, 0u);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a0[rt] = *reinterpret_cast<const f32x4*>(abase + rt * R48_USERS * LDA);
    constexpr uint32_t WS = NCT * 1024;
    if constexpr (!SHARE) rc_acc_begin<CW>(acc);
    if constexpr (SHARE) {
      // even K-steps: the lower wave multiplies its whole window [0, CW), the upper wave [1, CW); odd K-steps the other way round
      // ([0, CW - 1) / [0, CW)); each K-step fetches the B fragments of the tiles the NEXT one multiplies
      // the previous layer's outputs leave for HBM behind this layer's MFMAs, one chunk per K-step (layer 0: nothing to move, the
      // sweep starts with no chunks left and every store of it is dropped)
      R48Sweep swp;
      swp.template init<QP, LDA>(tid, a.ldp);
      if (layer == 0 || !a.sweep) swp.left = 0;
      const size_t lrow = (size_t)(layer > 0 ? layer - 1 : 0) * a.pre_stride + grow0 * a.ldp;
      const brsrc ares = make_brsrc(a.act + lrow, layer > 0 ? (uint32_t)(R48_ROWS * a.ldp * 4) : 0u);
      const bool prev_keep_pre = !(a.skip_pre && (layer <= 1 ? *a.slope0 : *a.slopeh) >= SLOPE_FROM_ACT_MIN);   // (of the layer whose outputs are swept now)
      const brsrc pres = make_brsrc(a.pre + lrow, layer > 0 && prev_keep_pre ? (uint32_t)(R48_ROWS * a.ldp * 4) : 0u);
      auto kloop = [&](auto up_tag) {
        constexpr bool UP = decltype(up_tag)::value;
        constexpr int E0 = UP ? 1 : 0, E1 = CW, O0 = 0, O1 = UP ? CW : CW - 1;   // tile ranges of the even / the odd K-steps
        rc_acc_begin<CW>(acc);   // (the guards inside each branch: dgrad_rows.h, dr_layer)
#pragma unroll 1
        for (uint32_t ks = 0; ks < (uint32_t)KS - 2; ks += 2) {
          r48_kstep<CW, E0, E1, O0, O1, LDA, RC_PLAIN, true, QP>(acc, a0, b0, a1, b1, Wf, (ks + 1) * WS, lane16, aoff + 64u * (ks + 1), Act, &swp, ares, pres, STG);
          r48_kstep<CW, O0, O1, E0, E1, LDA, RC_PLAIN, true, QP>(acc, a1, b1, a0, b0, Wf, (ks + 2) * WS, lane16, aoff + 64u * (ks + 2), Act, &swp, ares, pres, STG);
        }
        if constexpr (LIGHT) {
          r48_kstep<CW, E0, E1, O0, O1, LDA, RC_NEXT_LIGHT, true, QP>(acc, a0, b0, a1, b1, Wf, (KS - 1) * WS, lane16, aoffl + 64u * (KS - 1), Act, &swp, ares, pres, STG);
          r48_kstep<CW, O0, O1, O0, O0, LDA, RC_LIGHT>(acc, a1, b1, a0, b0, Wf, 0u, lane16, aoff, Act, &swp, ares, pres, STG);
        } else {
          r48_kstep<CW, E0, E1, O0, O1, LDA, RC_PLAIN, true, QP>(acc, a0, b0, a1, b1, Wf, (KS - 1) * WS, lane16, aoff + 64u * (KS - 1), Act, &swp, ares, pres, STG);
          r48_kstep<CW, O0, O1, O0, O0, LDA, RC_PLAIN, true, QP>(acc, a1, b1, a0, b0, Wf, (KS - 2) * WS, lane16, aoff + 64u * (KS - 2), Act, &swp, ares, pres, STG);   // (A fragments past the end: a harmless re-read)
        }
        rc_acc_settle<CW>(acc);
      };
      if (upper) kloop(std::true_type{});
      else kloop(std::false_type{});
    } else {
#pragma unroll 1   // (a narrow net's few trips would be unrolled into the layer loop: code size for nothing)
    for (uint32_t ks = 0; ks < (uint32_t)KS - 2; ks += 2) {
      rc_kstep<CW, LDA, 0, false, false, RC_PLAIN, R48_USERS>(acc, a0, b0, a1, b1, Wf, (ks + 1) * WS, lane16, aoff + 64u * (ks + 1), Act, nores, sw);
      rc_kstep<CW, LDA, 0, false, false, RC_PLAIN, R48_USERS>(acc, a1, b1, a0, b0, Wf, (ks + 2) * WS, lane16, aoff + 64u * (ks + 2), Act, nores, sw);
    }
    // the last pair: K-step KS - 1 may be the compact one (a.light: four real k in its sixteen)
    if constexpr (LIGHT) {
      rc_kstep<CW, LDA, 0, false, false, RC_NEXT_LIGHT, R48_USERS>(acc, a0, b0, a1, b1, Wf, (KS - 1) * WS, lane16, aoffl + 64u * (KS - 1), Act, nores, sw);
      rc_kstep<CW, LDA, 0, false, false, RC_LIGHT, R48_USERS>(acc, a1, b1, a0, b0, Wf, 0u, lane16, aoff, Act, nores, sw);
    } else {
      rc_kstep<CW, LDA, 0, false, false, RC_PLAIN, R48_USERS>(acc, a0, b0, a1, b1, Wf, (KS - 1) * WS, lane16, aoff + 64u * (KS - 1), Act, nores, sw);
      rc_kstep<CW, LDA, 0, false, false, RC_PLAIN, R48_USERS>(acc, a1, b1, a0, b0, Wf, (KS - 2) * WS, lane16, aoff + 64u * (KS - 2), Act, nores, sw);   // past the end: a harmless re-read
    }
    }
    if constexpr (!SHARE) rc_acc_settle<CW>(acc);
    if constexpr (SHARE) {
      // the pair's shared tile: the upper wave's half of the sum goes to the lower wave (same tile, same lane layout in both)
      if (upper) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xsh[((wc >> 1) * 3 + rt) * 64 + lane] = acc[rt][0];
      }
      __syncthreads();   // (for a hidden layer also: every wave is done reading the tile)
      if (!upper) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][CW - 1] += xsh[((wc >> 1) * 3 + rt) * 64 + lane];
      }
    }
    if (last) break;

    // in-place epilogue: every wave is done reading the tile; then the pre-activations go to HBM as they are (read next by the
    // dgrads), their PReLU into the tile - the next layer's input - and to HBM (act[layer]: what the weight gradients read)
    if constexpr (!SHARE) __syncthreads();
    {
      const float slope = layer == 0 ? *a.slope0 : *a.slopeh;
      const bool keep_pre = !(a.skip_pre && slope >= SLOPE_FROM_ACT_MIN);   // (uniform; rowchain.h: skip_pre)
      gchar* pw = uniform_gptr(a.pre + (size_t)layer * a.pre_stride + grow0 * a.ldp);
      gchar* aw = uniform_gptr(a.act + (size_t)layer * a.pre_stride + grow0 * a.ldp);
      const uint32_t pbase = (uint32_t)((myrow * a.ldp + mycol) * 4);
      const uint32_t prt = (uint32_t)(R48_USERS * a.ldp * 4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CW; ++ct) {
          if (!ALLV && wt0 + ct >= t1) continue;   // (wave-uniform) a tile beyond the layer's / the work-group's columns
          if (SHARE && upper && ct == 0) continue;  // (the pair's shared tile belongs to the lower wave)
          const f32x4 v = acc[rt][ct];
          const float4 h = make_float4(prelu_any(v[0], slope), prelu_any(v[1], slope), prelu_any(v[2], slope), prelu_any(v[3], slope));
          if (SHARE && a.sweep) {
            // both stay in LDS: the next layer's K-steps move them to HBM (R48Sweep)
            if (keep_pre) *reinterpret_cast<float4*>(otile + (R48_ROWS + rt * R48_USERS) * LDA + 16 * ct) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            if (keep_pre && !(R48_DIAG & 1)) gstore4(pw + ct * 64, pbase + rt * prt, make_float4(v[0], v[1], v[2], v[3]));
            if (!(R48_DIAG & 2)) gstore4(aw + ct * 64, pbase + rt * prt, h);
          }
          *reinterpret_cast<float4*>(otile + rt * R48_USERS * LDA + 16 * ct) = h;
        }
    }
    if constexpr (G > 1) {
      // the other work-groups of the group hold the rest of the next layer's input: their stores of act[layer] have reached the
      // XCD's L2 when the group has met; pull their column quads into the tile (every quad of a thread in flight before the wait)
      ++xphase;
      if (!r48_group_barrier(xmy, a.xbase + (unsigned)G * xphase, a.xabort, &go)) return;
      const float* asrc = a.act + (size_t)layer * a.pre_stride + grow0 * a.ldp;
      const int q0 = 4 * t0, q1 = 4 * t1;   // this work-group's own column quads
      f32x4 pv[C::NPULL];
#pragma unroll
      for (int i = 0; i < C::NPULL; ++i) {
        const int q = tid + NTHREADS * i, row = q / QP, cq = q - row * QP;
        if (q < R48_ROWS * QP && (cq < q0 || cq >= q1)) pv[i] = r48_load_l2(asrc + (size_t)row * a.ldp + 4 * cq);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < C::NPULL; ++i) {
        const int q = tid + NTHREADS * i, row = q / QP, cq = q - row * QP;
        if (q < R48_ROWS * QP && (cq < q0 || cq >= q1)) *reinterpret_cast<f32x4*>(Act + row * LDA + 4 * cq) = pv[i];
      }
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- out layer: tanh, Y, loss partial sums (:196-198), from registers
  gchar* yw = uniform_gptr(a.Y + grow0 * a.ldy);
  const uint32_t ybase = (uint32_t)((myrow * a.ldy + mycol) * 4), yrt = (uint32_t)(R48_USERS * a.ldy * 4);
  const bool uok = u0 + myrow < a.B;
  double sD = 0, sC = 0, sR = 0, sR2 = 0;
#pragma unroll
  for (int ct = 0; ct < CW; ++ct) {
    if (!ALLV && wt0 + ct >= t1) continue;
    if (SHARE && upper && ct == 0) continue;
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
        const float D = (Q[i] - S[i]) * (1.f / MU2) - R;
        const float RS = R - S[i];
        fD += D * D; fC += RS * RS; fR += R; fR2 += R * R;
      }
    }
    sD += fD; sC += fC; sR += fR; sR2 += fR2;
  }
  double tot[4] = {sD, sC, sR, sR2};
  block_sum4(tot, red);
  if (tid == 0) {
    double* o = a.loss_part + 4 * ((size_t)g * G + part);   // one partial per work-group: groups x G of them
    o[0] = tot[0]; o[1] = tot[1]; o[2] = tot[2]; o[3] = tot[3];
  }
}

// The whole input-gradient chain of such a step in ONE launch (dgrad_rows.h: k_dgrad_chain, on 48-row work-groups): the loss value
// and the gradient seeds of the work-group's 16 users, then every layer's dgrad from the output layer down.
// `pad_rows`: stacked rows behind the last group up to the padded row count the tile kernels may sum over (a weight gradient on the
// 64-row tiles): the last work-group zero-fills them in dY and every layer's output, whatever an earlier step left there.
struct DgradChain48Args {
  DgradChainArgs c;
  int pad_rows;
  // column-split row groups only: hand-shake counters (their own set: the forward's count on), number of row groups
  unsigned* xcnt; unsigned xbase; unsigned* xabort; int ngroups;
};

// G > 1 (column-split row groups, see the head of this file): every work-group of a group computes the group's seeds itself (its
// own copy of dY: the same values at the same addresses), owns 1 / G of every layer's output columns, and the group meets between
// two layers - layer l + 1 reduces over ALL columns of layer l's output, which it then reads from the XCD's L2 (sc1 loads).
// SHARE: the shared-tile form (see k_rows48_fwd): G == 1, 4 q + 2 column tiles.
template <int CT, bool LIGHT = false, int G = 1, bool SHARE = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_rows48_dgrad_chain(const DgradChain48Args ca) {
  typedef Rows48Cfg<CT, G> C;
  constexpr int NCT = C::NCT, NCG = C::NCG, NQ = C::NQ;
  constexpr int CW = SHARE ? NCT / 4 + 1 : C::CW;
  static_assert(!SHARE || (G == 1 && NCT % 4 == 2), "the shared-tile form: one work-group per row group, 4 q + 2 column tiles");
  __shared__ f32x4 xsh[SHARE ? 2 * 3 * 64 : 1];
  __shared__ float red[4];
  __shared__ double shs[4], tot[4];
  __shared__ int go;
  const DgradChainArgs& c = ca.c;
  const SeedArgs& a = c.seed;
  const int tid = threadIdx.x;
  int g = blockIdx.x, part = 0, ngroups = gridDim.x;
  if constexpr (G > 1) {
    const int slot = (int)blockIdx.x >> 3;
    part = slot % G;
    g = ((int)blockIdx.x & 7) + 8 * (slot / G);
    ngroups = ca.ngroups;
    if (g >= ngroups) return;
  }
  const int t0 = part * NCG, t1 = min(t0 + NCG, NCT);
  {
    // the five sums: given (sharded step, after the all-reduce) or folded here from the forward's per-work-group partials,
    // exactly as k_loss_seed does
    double s0, s1, s2, s3, N;
    if (a.sums) {
      s0 = a.sums[0]; s1 = a.sums[1]; s2 = a.sums[2]; s3 = a.sums[3]; N = a.sums[4];
    } else {
      double v[4] = {0, 0, 0, 0};
      for (int i = tid; i < a.nblk; i += NTHREADS)
        for (int j = 0; j < 4; ++j) v[j] += a.part[4 * (size_t)i + j];
      for (int j = 0; j < 4; ++j) {
        const double t = block_sum(v[j], shs);
        if (tid == 0) tot[j] = t;
      }
      __syncthreads();
      s0 = tot[0]; s1 = tot[1]; s2 = tot[2]; s3 = tot[3]; N = a.count;
    }
    const double A = s0 / N, Cc = s1 / N, Rbar = s2 / N;
    const double V = (N > 1.0) ? (s3 - N * Rbar * Rbar) / (N - 1.0) : __builtin_nan("");
    const double den = 1e-8 + V;
    const double k = 0.5 / den;
    const float cD = (float)(2.0 * k / N);
    const float cV = (float)(-(0.5 * (A + Cc) / (den * den)) * 2.0 / (N - 1.0));
    const float rbar = (float)Rbar;
    if (g == 0 && part == 0 && tid == 0 && a.loss) *a.loss = (float)(0.5 * (A + Cc) / den);
    // thread -> (user tid / 16 of the group, column quads tid % 16 + 16 j)
    const int su = tid >> 4, sq = tid & 15, r = R48_USERS * g + su;
    const size_t rowP = (size_t)R48_ROWS * g + su;
    float4 P4[NQ], S4[NQ], Q4[NQ], X4[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int col = 4 * (sq + 16 * j);
      const size_t yP = rowP * a.LP + col, yS = yP + (size_t)R48_USERS * a.LP, yQ = yS + (size_t)R48_USERS * a.LP;
      const bool ok = r < a.B && col < a.L;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      P4[j] = ok ? *reinterpret_cast<const float4*>(a.Y + yP) : z;
      S4[j] = ok ? *reinterpret_cast<const float4*>(a.Y + yS) : z;
      Q4[j] = ok ? *reinterpret_cast<const float4*>(a.Y + yQ) : z;
      X4[j] = ok ? load4_unpadded(a.x0, r, col, a.L) : z;
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int col = 4 * (sq + 16 * j);
      if (col >= a.LP) continue;
      const size_t yP = rowP * a.LP + col, yS = yP + (size_t)R48_USERS * a.LP, yQ = yS + (size_t)R48_USERS * a.LP;
      f32x4 gP = {0.f, 0.f, 0.f, 0.f}, gS = gP, gQ = gP;
      if (r < a.B && col < a.L) {
        const f32x4 P = {P4[j].x, P4[j].y, P4[j].z, P4[j].w}, S = {S4[j].x, S4[j].y, S4[j].z, S4[j].w},
                    Q = {Q4[j].x, Q4[j].y, Q4[j].z, Q4[j].w}, X = {X4[j].x, X4[j].y, X4[j].z, X4[j].w};
        const f32x4 R = P - X;
        const f32x4 D = (Q - S) * (1.f / MU2) - R;
        const f32x4 gD = cD * D;
        const f32x4 gC = cD * (R - S);
        const f32x4 gV = cV * (R - rbar);
        const f32x4 gDm = gD * (1.f / MU2);
        gP = (-gD + gC + gV) * (1.f - P * P);
        gQ = gDm * (1.f - Q * Q);
        gS = (-gDm - gC) * (1.f - S * S);
#pragma unroll
        for (int i = 1; i < 4; ++i)
          if (col + i >= a.L) { gP[i] = 0.f; gQ[i] = 0.f; gS[i] = 0.f; }
      }
      *reinterpret_cast<float4*>(a.dY + yP) = make_float4(gP[0], gP[1], gP[2], gP[3]);
      *reinterpret_cast<float4*>(a.dY + yS) = make_float4(gS[0], gS[1], gS[2], gS[3]);
      *reinterpret_cast<float4*>(a.dY + yQ) = make_float4(gQ[0], gQ[1], gQ[2], gQ[3]);
    }
    if (g == ngroups - 1 && part == 0 && ca.pad_rows > 0) {
      // the padding rows behind the last group: zero gradients in every buffer a weight gradient may read them from
      const size_t r0 = (size_t)R48_ROWS * ngroups;
      const int q4 = a.LP >> 2;
      for (int i = tid; i < ca.pad_rows * q4; i += NTHREADS)
        *reinterpret_cast<float4*>(a.dY + (r0 + i / q4) * a.LP + 4 * (i % q4)) = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int l = 0; l < c.nlayers; ++l) {
        const int ldo = c.layer[l].ldo, qo = ldo >> 2;
        for (int i = tid; i < ca.pad_rows * qo; i += NTHREADS)
          *reinterpret_cast<float4*>(c.layer[l].out + (r0 + i / qo) * ldo + 4 * (i % qo)) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  unsigned* const xmy = G > 1 ? ca.xcnt + 32 * (size_t)g : nullptr;
  for (int l = 0; l < c.nlayers; ++l) {
    if constexpr (G > 1) {
      if (l == 0) {
        __syncthreads();   // this work-group's own copy of the seeds has landed
        dr_layer<CW, LIGHT, R48_ROWS, NCT, true, 0>(c.layer[l], g, red, t0, t1, g * G + part);
      } else {
        // layer l - 1's output columns of the other work-groups of the group: in the XCD's L2 once the group has met
        if (!r48_group_barrier(xmy, ca.xbase + (unsigned)G * (unsigned)l, ca.xabort, &go)) return;
        dr_layer<CW, LIGHT, R48_ROWS, NCT, true, BUF_SC1>(c.layer[l], g, red, t0, t1, g * G + part);
      }
    } else {
      __syncthreads();   // the work-group's own stores of the previous stage have landed (and `red` is free again)
      dr_layer<CW, LIGHT, R48_ROWS, NCT, false, 0, SHARE>(c.layer[l], g, red, 0, NCT, -1, xsh);
    }
  }
}

}  // namespace sdrm
