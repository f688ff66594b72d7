// Row-owned input gradient (dgrad) of one eps-net layer behind the row-owned forward (rowchain.h), gfx950:
//   out[r][c] = (sum_k G[r][k] * W[k][c]) * prelu'(pre[r][c]),   slope partial = sum acc * min(pre, 0)
// (train_SDRM.py:336, loss.backward(): the input gradients of the hidden and output Linears with PReLU' folded in).
//
// Ownership as in the forward - ONE work-group per CU owns 96 stacked rows and ALL NP output columns, 4 waves as 2 x 2, a wave
// 3 row tiles x CT column tiles of v_mfma_f32_16x16x4_f32 (33 accumulator quads at NP = 352) - but nothing lives in LDS:
//   * G fragments come straight from global memory: lane (row li, k group lq) takes the 16 bytes G[row][16 ks + 4 lq ..], a
//     wave-load is 16 rows x 64 B; the work-group's 96 rows are one contiguous block of HBM, read once (the partner wave's
//     copy and the second half of every 128-byte line hit in L2), fetched TWO K-steps ahead (four register sets);
//   * W fragments come from L2 out of the fragment-packed copy of the layer's weight in [k = out][n = in] order (k_adam's
//     dstFT), one K-step ahead, exactly as in the forward;
//   * no LDS, no barrier, no VALU in the K loop: every memory instruction sits behind an MFMA, where it is free (measured
//     at the ISA level, profiles/r03_mfma_shadow_asm_probe.txt: 132 MFMAs + 11 consumed wave-loads = 1.02 x the MFMAs alone);
//   * the MFMA operands are SWAPPED (weights as srcA, gradients as srcB): the tile comes out transposed, so a lane holds four
//     consecutive COLUMNS of one row - pre-activations in and input gradients out are 16-byte accesses of row-major memory,
//     and the pre-activation quads are requested in the shadows of the last two K-steps.
// 24576 stacked rows = 256 work-groups = one round of the chip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowchain.h"

#ifndef DR_DIAG   // diagnostic builds only: bit0 drops the W fragment loads of the K loop, bit1 the G fragment loads, bit2 ADDS an s_nop 1 in front of every MFMA
#define DR_DIAG 0
#endif
#ifndef DR_PIECE_STRIDE
#define DR_PIECE_STRIDE 3
#endif

namespace sdrm {

struct DgradRowsArgs {
  const float* G; int ldg;      // incoming gradient [MP][ldg], columns [0, NP) are the reduction axis
  const float* WfT;             // fragment-packed [NP/16][NP/16][64][4]: element (n = in, k = out), wfrag_index(n, k, NP / 16)
  const float* pre; int ldp;    // pre-activations of the layer below [MP][ldp]
  const float* act;             // ... its activations prelu(pre) in the same layout, and whether to read THEM (`from_act`, with a positive
  int from_act;                 // slope): PReLU'(v) = (prelu(v) > 0 ? 1 : slope) and sum dh min(v, 0) = (sum dh min(prelu(v), 0)) / slope -
                                // the forward then need not store the pre-activations at all (rowchain.h: skip_pre)
  const float* slope;           // its PReLU slope
  float* out; int ldo;          // [MP][ldo]
  float* slope_part;            // [gridDim.x]
  unsigned long long* stamps;   // diagnostic builds only (-DDR_STAMPS): 8 slots per work-group
};

#ifdef DR_STAMPS
#define DR_STAMP(i) do { if (tid == 0) st_[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DR_STAMP(i) do { } while (0)
#endif

// one K-step of phase PH (= K-step index mod 4): 12 * CT MFMAs out of (A[PH], B[PH & 1]); in their shadows the W fragments of the
// next K-step (NB = CT of them -> B[(PH + 1) & 1]), the G fragments of the K-step after that (NA = 3 -> A[(PH + 2) & 3]), and NPRE
// pre-activation quads (-> pq[PRE0 ..], flat index rt * CT + ct).  One basic block, every load pinned at its slot through an
// opaque scalar base (see rc_kstep).
// MODE (rowchain.h): RC_NEXT_LIGHT - the G pieces fetch the compact fragments of the layer's last K-step (one float per lane at
// k = 16 ks + lane group: `goff` then holds those offsets); RC_LIGHT - this is that K-step: one MFMA per tile.
// GAUX: cache policy of the G fragment loads (bufres.h; BUF_SC1 where another work-group of the launch wrote G)
// [C0, C1): the tiles of the wave's CT this K-step multiplies; its NB B-fragment loads are for the tiles [L0, L0 + NB) (the shared-tile
// form of rows48.h: a wave multiplies one tile of its window on alternate K-steps only)
template <int CT, int PH, int NA, int NB, int PRE0, int NPRE, int MODE = RC_PLAIN, int GAUX = 0, int C0 = 0, int C1 = CT, int L0 = 0>
__device__ __forceinline__ void dr_kstep(f32x4 (&acc)[3][CT], f32x4 (&A)[4][3], f32x4 (&B)[2][CT], brsrc gr, uint32_t gnext,
                                         const uint32_t (&goff)[3], brsrc wr_, uint32_t wnext, uint32_t lane16, f32x4 (&pq)[3 * CT],
                                         brsrc pr, const uint32_t (&poff)[3]) {
  constexpr int NE = MODE == RC_LIGHT ? 1 : 4;
  constexpr int NTL = C1 - C0;
  constexpr int NSLOT = 3 * NE * NTL, NPIECE = NB + NA + NPRE;
  static_assert(0 <= C0 && C0 < C1 && C1 <= CT && L0 >= 0 && L0 + NB <= CT, "tile ranges");
  // the pieces go to the FRONT of the K-step, one per DR_PIECE_STRIDE MFMAs (behind an MFMA a load's issue is free)
  constexpr int STRIDE = NPIECE > 0 ? (NSLOT / NPIECE >= DR_PIECE_STRIDE ? DR_PIECE_STRIDE : (NSLOT / NPIECE >= 1 ? NSLOT / NPIECE : 1)) : NSLOT;
  static_assert(NPIECE <= NSLOT, "not enough MFMA slots for the pipeline pieces");
  const f32x4 (&au)[3] = A[PH];
  const f32x4 (&bu)[CT] = B[PH & 1];
  f32x4 (&al)[3] = A[(PH + 2) & 3];
  f32x4 (&bl)[CT] = B[(PH + 1) & 1];
#pragma unroll
  for (int e = 0; e < NE; ++e)
#pragma unroll
  for (int ct = C0; ct < C1; ++ct)
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int s = (e * NTL + (ct - C0)) * 3 + rt;
    // swapped operands: the tile comes out transposed.  Written as asm with the accumulator tied in place: through the builtin
    // the register allocator renames accumulators in the last K-step of the loop body and copies them back at its top
    // (~250 v_accvgpr_mov per trip).  The price: the compiler's hazard recogniser does not see an MFMA here: rc_acc_begin /
    // rc_acc_settle (rowchain.h) guard the two places where compiler-made VALU code meets the accumulators (their zeros, their first read);
    // tests/test_isa_lint.py checks the generated code for any other
    if (DR_DIAG & 4) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[rt][ct]) : "v"(bu[ct][e]), "v"(au[rt][e]));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[rt][ct]) : "v"(bu[ct][e]), "v"(au[rt][e]));
    if (s % STRIDE == 0 && s / STRIDE < NPIECE) {
      const int p = s / STRIDE;
      if (p < NB) {
        if (!(DR_DIAG & 1)) {
          uint32_t so = wnext + ((L0 + p) / 4) * 4096;   // opaque at this slot: pins the load here
          asm volatile("" : "+s"(so));
          bl[L0 + p] = bload4(wr_, lane16 + ((L0 + p) % 4) * 1024, so);
        }
      } else if (p < NB + NA) {
        if (!(DR_DIAG & 2)) {
          uint32_t so = gnext;
          asm volatile("" : "+s"(so));
          if (MODE == RC_NEXT_LIGHT) al[p - NB][0] = bload1a<GAUX>(gr, goff[p - NB], so);
          else al[p - NB] = bload4a<GAUX>(gr, goff[p - NB], so);
        }
      } else {
        const int q = PRE0 + (p - NB - NA), rt2 = q / CT, ct2 = q % CT;
        uint32_t so = ct2 * 64;
        asm volatile("" : "+s"(so));
        pq[q] = bload4(pr, poff[rt2], so);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// SH (the shared-tile form): 0 every K-step multiplies all CT tiles; 1 (the lower wave of a pair) the odd K-steps leave out the
// window's LAST tile; 2 (the upper wave) the even K-steps leave out its FIRST tile
__host__ __device__ constexpr int dr_c0(int SH, int K) { return (SH == 2 && (K & 1) == 0) ? 1 : 0; }
__host__ __device__ constexpr int dr_c1(int SH, int K, int CT) { return (SH == 1 && (K & 1) == 1) ? CT - 1 : CT; }

// the last PEEL K-steps, straight-line: K-step KS - PEEL + J has phase J & 3 (KS - PEEL is a multiple of four), fetches what is
// still to come, and takes its share of the 3 * CT pre-activation quads - spread over all of them: in the last two K-steps alone
// they are a 35 MB burst chip-wide, more than HBM delivers in that time
// LIGHT: K-step KS - 1 is the compact one (its G fragments, fetched by K-step KS - 3, come from `goffl`)
template <int CT, int KS, int PEEL, bool LIGHT, int J, int NCT = 2 * CT, int GAUX = 0, int SH = 0>
__device__ __forceinline__ void dr_peel(f32x4 (&acc)[3][CT], f32x4 (&A)[4][3], f32x4 (&B)[2][CT], brsrc Gw, const uint32_t (&goff)[3],
                                        const uint32_t (&goffl)[3], brsrc Wf, uint32_t lane16, f32x4 (&pq)[3 * CT], brsrc Pw,
                                        const uint32_t (&poff)[3]) {
  static_assert(PEEL >= 3, "the compact K-step's fragments are fetched inside the peeled part");
  if constexpr (J < PEEL) {
    constexpr int K = KS - PEEL + J, NQ = 3 * CT;
    constexpr int MODE = !LIGHT ? RC_PLAIN : (K == KS - 3 ? RC_NEXT_LIGHT : (K == KS - 1 ? RC_LIGHT : RC_PLAIN));
    constexpr int NA = K + 2 < KS ? 3 : 0;
    constexpr int L0 = dr_c0(SH, K + 1), NB = K + 1 < KS ? dr_c1(SH, K + 1, CT) - L0 : 0;   // the next K-step's tiles
    constexpr int PRE0 = J * NQ / PEEL, NPRE = (J + 1) * NQ / PEEL - PRE0;
    constexpr uint32_t WSTEP = (uint32_t)NCT * 1024u;
    dr_kstep<CT, J & 3, NA, NB, PRE0, NPRE, MODE, GAUX, dr_c0(SH, K), dr_c1(SH, K, CT), (NB > 0 ? L0 : 0)>(
        acc, A, B, Gw, 64u * (K + 2 < KS ? K + 2 : 0), MODE == RC_NEXT_LIGHT ? goffl : goff, Wf, (K + 1 < KS ? K + 1 : 0) * WSTEP, lane16, pq, Pw, poff);
    dr_peel<CT, KS, PEEL, LIGHT, J + 1, NCT, GAUX, SH>(acc, A, B, Gw, goff, goffl, Wf, lane16, pq, Pw, poff);
  }
}

// one layer for the work-group's ROWS stacked rows (block index g); `red`: four floats of LDS.
// ROWS = 96 (this file's kernels: 32 users, waves as 2 x 2, CT column tiles each) or 48 (rows48.h: 16 users, the four waves side
// by side, CT = ceil(NCT / 4) column tiles each - the last wave's tiles beyond the NCT real ones are computed and dropped).
// SPLIT (rows48.h, several work-groups per row group): this work-group owns the column tiles [t0, t1) only and writes slope partial
// `pidx`; GAUX: cache policy of its G loads (the other work-groups of the group wrote most of G in this launch).
// SHARE (rows48.h: ROWS = 48, NCT = 4 q + 2, CT = q + 1): the waves of a pair (0, 1) / (2, 3) have windows of CT consecutive tiles
// that overlap in one tile, multiplied by the lower wave on the even K-steps and by the upper wave on the odd ones; the upper wave
// hands its partial sums of that tile to the lower wave through `xsh` (2 x 3 x 64 quads of LDS), which owns the tile's epilogue.
template <int CT, bool LIGHT, int ROWS = RC_ROWS, int NCT = 2 * CT, bool SPLIT = false, int GAUX = 0, bool SHARE = false>
__device__ __forceinline__ void dr_layer(const DgradRowsArgs& a, int g, float* red, int t0 = 0, int t1 = NCT, int pidx = -1, f32x4* xsh = nullptr) {
  constexpr int KS = NCT, NQ = 3 * CT;
  constexpr int WC = 4 / (ROWS / 48);             // waves side by side along the columns
  constexpr bool ALLV = SHARE || (!SPLIT && CT * WC == NCT); // every tile of every wave is a real one
  static_assert(ROWS == 48 || ROWS == 96, "a work-group owns the P, S, Q rows of 16 or 32 users");
  static_assert(SPLIT || SHARE || (CT * WC >= NCT && (CT - 1) * WC < NCT), "per-wave column tiles do not cover the layer");
  static_assert(!SHARE || (ROWS == 48 && !SPLIT && NCT % 4 == 2 && CT == NCT / 4 + 1), "the shared-tile form");
  static_assert(KS % 2 == 0 && KS >= 4, "K-steps are taken in fours with a tail of two or four");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 15, lq = lane >> 4;
  const size_t grow0 = (size_t)ROWS * g;
  const bool upper = SHARE && (wave & 1);         // the wave whose window starts with the pair's shared tile
  const int wt0 = SHARE ? (wave >> 1) * (NCT / 2) + (upper ? NCT / 4 : 0) : t0 + CT * wc;   // the wave's first column tile
  const uint32_t lane16 = 16u * (uint32_t)lane;
#ifdef DR_STAMPS
  unsigned long long st_[8] = {0};
#endif
  DR_STAMP(0);

  // byte offsets of this lane's 16-byte pieces relative to the work-group's first row: G fragment of row tile rt (+ 64 ks),
  // and the accumulator quad of (rt, ct) in pre / out (+ 64 ct): row 48 wr + 16 rt + li, columns 16 (CT wc + ct) + 4 lq ..
  uint32_t goff[3], goffl[3], poff[3], ooff[3];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int row = 48 * wr + 16 * rt + li;
    goff[rt] = (uint32_t)((row * a.ldg + 4 * lq) * 4);
    goffl[rt] = (uint32_t)((row * a.ldg + lq) * 4);   // the compact K-step: k = 16 ks + lq
    poff[rt] = (uint32_t)((row * a.ldp + 16 * wt0 + 4 * lq) * 4);
    ooff[rt] = (uint32_t)((row * a.ldo + 16 * wt0 + 4 * lq) * 4);
  }
  const brsrc Gw = make_brsrc(a.G + grow0 * a.ldg, (uint32_t)(ROWS * a.ldg * 4));
  const brsrc Wf = make_brsrc(a.WfT + (size_t)wt0 * 256, (uint32_t)((KS * NCT - wt0) * 1024));
  const float slope = *a.slope;
  const bool ua = a.from_act && slope >= SLOPE_FROM_ACT_MIN;   // (uniform) the activations stand in for the pre-activations
  const brsrc Pw = make_brsrc((ua ? a.act : a.pre) + grow0 * a.ldp, (uint32_t)(ROWS * a.ldp * 4));
  gchar* Ow = uniform_gptr(a.out + grow0 * a.ldo);

  f32x4 acc[3][CT];
  f32x4 A[4][3];    // G fragments of K-step k live in A[k & 3] (fetched two K-steps ahead)
  f32x4 B[2][CT];   // W fragments of K-step k in B[k & 1] (one K-step ahead)
  f32x4 pq[NQ];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  // prologue: G fragments of K-steps 0 and 1, W fragments of K-step 0
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    A[0][rt] = bload4a<GAUX>(Gw, goff[rt], 0u);
    A[1][rt] = bload4a<GAUX>(Gw, goff[rt], 64u);
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) B[0][ct] = bload4(Wf, lane16 + ct * 1024u, 0u);

  DR_STAMP(1);
  constexpr uint32_t WSTEP = (uint32_t)NCT * 1024u;   // bytes of one K-step of the fragment-packed weights
  // main loop: four K-steps per trip (the G sets rotate with period four, the W sets with period two); the last PEEL K-steps
  // (PEEL = KS mod 4, at most ten, at least one trip left) are straight-line code and carry the pre-activation loads
  constexpr int PEEL = KS % 4 == 2 ? (KS >= 14 ? 10 : (KS >= 10 ? 6 : 2)) : (KS >= 12 ? 8 : 4);
  static_assert((KS - PEEL) % 4 == 0 && KS - PEEL >= 4, "the main loop takes whole trips of four K-steps");
  // (rc_acc_begin / rc_acc_settle INSIDE each instantiation of the loop: where two of them meet - the shared-tile form - the
  // register allocator may reconcile the accumulators' registers with copies at the branch's entry and exit, and those copies must
  // sit outside the guards, away from the asm MFMAs: tests/test_isa_lint.py found `v_accvgpr_mov a48, a4` one wait state in front
  // of the first MFMA of a branch - wrong, run-to-run different gradients - when the guards stood around the branch)
  auto kloop = [&](auto sh_tag) {
    constexpr int SH = decltype(sh_tag)::value;
    constexpr int E0 = dr_c0(SH, 0), E1 = dr_c1(SH, 0, CT), O0 = dr_c0(SH, 1), O1 = dr_c1(SH, 1, CT);   // tiles of the even / odd K-steps
    rc_acc_begin<CT>(acc);
    for (uint32_t ks = 0; ks < (uint32_t)(KS - PEEL); ks += 4) {
      dr_kstep<CT, 0, 3, O1 - O0, 0, 0, RC_PLAIN, GAUX, E0, E1, O0>(acc, A, B, Gw, 64u * (ks + 2), goff, Wf, (ks + 1) * WSTEP, lane16, pq, Pw, poff);
      dr_kstep<CT, 1, 3, E1 - E0, 0, 0, RC_PLAIN, GAUX, O0, O1, E0>(acc, A, B, Gw, 64u * (ks + 3), goff, Wf, (ks + 2) * WSTEP, lane16, pq, Pw, poff);
      dr_kstep<CT, 2, 3, O1 - O0, 0, 0, RC_PLAIN, GAUX, E0, E1, O0>(acc, A, B, Gw, 64u * (ks + 4), goff, Wf, (ks + 3) * WSTEP, lane16, pq, Pw, poff);
      dr_kstep<CT, 3, 3, E1 - E0, 0, 0, RC_PLAIN, GAUX, O0, O1, E0>(acc, A, B, Gw, 64u * (ks + 5), goff, Wf, (ks + 4) * WSTEP, lane16, pq, Pw, poff);
    }
    DR_STAMP(2);
    dr_peel<CT, KS, PEEL, LIGHT, 0, NCT, GAUX, SH>(acc, A, B, Gw, goff, goffl, Wf, lane16, pq, Pw, poff);
    rc_acc_settle<CT>(acc);
  };
  if constexpr (SHARE) {
    if (upper) kloop(std::integral_constant<int, 2>{});
    else kloop(std::integral_constant<int, 1>{});
  } else {
    kloop(std::integral_constant<int, 0>{});
  }
  DR_STAMP(3);
  if constexpr (SHARE) {
    // the pair's shared tile: the upper wave's half of the sum goes to the lower wave (same tile, same lane layout in both)
    if (upper) {
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) xsh[((wave >> 1) * 3 + rt) * 64 + lane] = acc[rt][0];
    }
    __syncthreads();
    if (!upper) {
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) acc[rt][CT - 1] += xsh[((wave >> 1) * 3 + rt) * 64 + lane];
    }
  }
  // epilogue: PReLU' (slope at pre <= 0, as the reference's autograd), the slope-gradient partial sum, 16-byte stores
  float part4[4] = {0.f, 0.f, 0.f, 0.f};   // four chains: one wave per SIMD, nothing else hides a dependent FMA's latency
#pragma unroll
  for (int rt = 0; rt < 3; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (!ALLV && t0 + CT * wc + ct >= t1) continue;   // (wave-uniform) a tile beyond the layer's / the work-group's columns: what it computed is dropped
      if (SHARE && upper && ct == 0) continue;          // (the pair's shared tile belongs to the lower wave)
      const f32x4 p = pq[rt * CT + ct];
      f32x4 v = acc[rt][ct];
      asm("" : "+v"(v));   // one copy out of the accumulator registers, every use below reads it
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float m;
        asm("v_min_f32 %0, 0, %1" : "=v"(m) : "v"(p[i]));   // min(pre, 0) without the canonicalising v_max fminf() puts in front
        part4[i] = fmaf(v[i], m, part4[i]);
        v[i] *= p[i] > 0.f ? 1.f : slope;
      }
      gstore4(Ow + ct * 64, ooff[rt], make_float4(v[0], v[1], v[2], v[3]));
    }
  float part = (part4[0] + part4[1]) + (part4[2] + part4[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) {
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    a.slope_part[pidx >= 0 ? pidx : g] = ua ? tot / slope : tot;   // (min(prelu(v), 0) = slope * min(v, 0))
  }
#ifdef DR_STAMPS
  DR_STAMP(4);
  if (tid == 0 && a.stamps) {
    st_[5] = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 8; ++i) a.stamps[8 * (size_t)g + i] = st_[i];
  }
#endif
}


// LIGHT: the weight copy's last K-step is compact (elementwise.h: wfrag_index; the host picks the instantiation)
template <int CT, bool LIGHT = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_dgrad_rows(const DgradRowsArgs a) {
  __shared__ float red[4];
  dr_layer<CT, LIGHT>(a, (int)blockIdx.x, red);
}

// The whole input-gradient chain of a train step in ONE launch: the loss value and the closed-form gradient seeds of the
// work-group's 32 users (k_loss_seed's arithmetic, elementwise.h, with 1 / mu^2 as a multiply), then every layer's dgrad from the output layer
// down - rows never meet, so a work-group runs down the chain on its own: between two layers only its own stores have to land
// (a barrier).  Two launches and the seeds' own pass over Y fewer than k_loss_seed + one k_dgrad_rows per layer.
constexpr int DR_MAX_LAYERS = 8;
struct DgradChainArgs {
  SeedArgs seed;                      // dY = the chain's first input; grouped row order
  int nlayers;                        // H + 1
  DgradRowsArgs layer[DR_MAX_LAYERS];
  unsigned long long* seed_stamps;    // diagnostic builds only (-DDR_STAMPS): cycles of the seed stage per work-group
};

template <int CT, bool LIGHT = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_dgrad_chain(const DgradChainArgs c) {
  __shared__ float red[4];
  __shared__ double shs[4], tot[4];
  const SeedArgs& a = c.seed;
  const int tid = threadIdx.x, g = blockIdx.x;
#ifdef DR_STAMPS
  const unsigned long long t_seed0 = __builtin_amdgcn_s_memtime();
#endif
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
    const double A = s0 / N, C = s1 / N, Rbar = s2 / N;
    const double V = (N > 1.0) ? (s3 - N * Rbar * Rbar) / (N - 1.0) : __builtin_nan("");
    const double den = 1e-8 + V;
    const double k = 0.5 / den;
    const float cD = (float)(2.0 * k / N);
    const float cV = (float)(-(0.5 * (A + C) / (den * den)) * 2.0 / (N - 1.0));
    const float rbar = (float)Rbar;
    if (g == 0 && tid == 0 && a.loss) *a.loss = (float)(0.5 * (A + C) / den);
    // thread -> (user tid / 8 of the group, column quads tid % 8 + 8 j)
    const int su = tid >> 3, sq = tid & 7, r = RC_USERS * g + su;
    const size_t rowP = (size_t)RC_ROWS * g + su;
    // every quad of the thread is requested before the first is used: one HBM round trip, not CT of them
    float4 P4[CT], S4[CT], Q4[CT], X4[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int col = 4 * (sq + 8 * j);
      const size_t yP = rowP * a.LP + col, yS = yP + (size_t)RC_USERS * a.LP, yQ = yS + (size_t)RC_USERS * a.LP;
      const bool ok = r < a.B && col < a.L;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      P4[j] = ok ? *reinterpret_cast<const float4*>(a.Y + yP) : z;
      S4[j] = ok ? *reinterpret_cast<const float4*>(a.Y + yS) : z;
      Q4[j] = ok ? *reinterpret_cast<const float4*>(a.Y + yQ) : z;
      X4[j] = ok ? load4_unpadded(a.x0, r, col, a.L) : z;
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int col = 4 * (sq + 8 * j);
      const size_t yP = rowP * a.LP + col, yS = yP + (size_t)RC_USERS * a.LP, yQ = yS + (size_t)RC_USERS * a.LP;
      // four columns at a time as vector arithmetic: the compiler pairs it into v_pk_mul / v_pk_add / v_pk_fma (two elements per
      // instruction) - with one wave per SIMD this stage is as much VALU time as memory time.
      // 1 / mu^2 as a multiply (as the forward's loss sums do): an IEEE division is ten instructions, three of them per element
      // made this stage 41 k cycles of the launch
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
  }
#ifdef DR_STAMPS
  if (tid == 0 && c.seed_stamps) c.seed_stamps[g] = __builtin_amdgcn_s_memtime() - t_seed0;
#endif
  for (int l = 0; l < c.nlayers; ++l) {
    __syncthreads();   // the work-group's own stores of the previous stage have landed (and `red` is free again)
    dr_layer<CT, LIGHT>(c.layer[l], g, red);
  }
}

}  // namespace sdrm
