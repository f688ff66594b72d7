// Strip-owned split-K weight gradients for mid-width eps-nets (padded width 128..352) on gfx950.
//
// All weight gradients of a train step, dW[n][k] = sum over stacked rows m of dpre[m][n] * act[m][k] (train_SDRM.py:336), in ONE
// launch of one work-group per CU.  The outputs of all layers are cut into STRIPS: all NT row tiles (32 output rows n each)
// of one 32-wide column tile k - NT accumulators of v_mfma_f32_32x32x2_f32 (16 registers each) in ONE wave, 11 at WP = 352.
// A work-group is four strips (four waves, one per SIMD) over one K-slice of the stacked rows; every strip is the same work,
// so the launch is one balanced round with no tail (the 64x64-tile split-K launch it replaces loses a quarter of its time to
// the staggered finish of the five work-groups sharing a CU).  Per 16-row K-step a wave issues 8 x NT MFMAs against 7-13
// global loads and LDS stores and 8 x (NT + 1) ds_read_b32: the operand tile [16][WP] of dpre is staged once per work-group
// and read by all four waves (the strips of a unit share their row tiles), each wave adds its own [16][32] column block.
// Operands are stored as the kernel needs them (the row-owned forward writes activations, not pre-activations): no transform
// on load.  Bias gradients: the operand carries a column of ones (pad column `ones_col` of U / act), so column ones_col of
// every slab IS the bias gradient - no column sums in the loop.
//
// Slabs: [slice][n][k] per problem, reduced in fixed order by k_tail (csrc/tail.h; deterministic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bufres.h"
#include "gemm.h"

// diagnostic builds only (-DWG2_DIAG=mask): bit0 no MFMAs / fragment reads, bit1 no LDS stores in the loop, bit2 no global loads in the loop
#ifndef WG2_DIAG
#define WG2_DIAG 0
#endif

namespace sdrm {

constexpr int WG2_MAX_PROBLEMS = 8;    // layers with a weight gradient (H + 2 <= 8)
constexpr int WG2_MAX_UNITS = 32;      // work-group units (4 strips each) per K-slice
constexpr int WG2_BK = 16;             // rows per K-step

struct Wg2Problem {
  const float* A; int lda;     // gradient operand dpre / dY: [rows][n], n < WP
  const float* B; int ldb;     // forward operand U / act: [rows][k]
  float* slab; int ldc;        // [slice][WP][ldc]
  size_t slab_stride;
};

struct Wg2Strip { int8_t problem; int8_t ktile; };   // problem < 0: idle wave

struct Wg2Args {
  Wg2Problem p[WG2_MAX_PROBLEMS];
  Wg2Strip strip[WG2_MAX_UNITS][4];
  int units, slices, rows, kchunk;   // rows = stacked rows (multiple of 16), kchunk = rows per slice (multiple of 16)
};

// problem `idx` of the argument block through constant indices and selects: a dynamic index into the by-value kernel argument
// makes the compiler keep a private copy of the table (scratch, or 32 KB of LDS)
__device__ __forceinline__ Wg2Problem wg2_problem(const Wg2Args& a, int idx) {
  Wg2Problem r = a.p[0];
#pragma unroll
  for (int p = 1; p < WG2_MAX_PROBLEMS; ++p)
    if (p == idx) r = a.p[p];
  return r;
}

template <int NT>
struct Wg2Cfg {
  static constexpr int WP = 32 * NT, LDA = WP + 4, LDB = 32 + 4;
  static constexpr int A_TILE = WG2_BK * LDA, B_TILE = WG2_BK * LDB;
  static constexpr int STAGE = 2 * A_TILE + 4 * B_TILE;          // two A tiles (a unit spans at most two problems), four B blocks
  static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * 4;
  static constexpr int NA4 = WG2_BK * WP / 4, NB4 = 4 * WG2_BK * 32 / 4;   // float4 per A tile / per four B blocks
  static constexpr int NLA = (NA4 + NTHREADS - 1) / NTHREADS, NLB = NB4 / NTHREADS;
  static_assert(NLB == 2, "two B blocks per half work-group");
  static_assert(LDS_BYTES <= 160 * 1024, "stages do not fit LDS");
};

// grid: units x slices work-groups; the (slice, unit) pairs, slice-major, are dealt to the 8 XCDs in contiguous runs (work-groups
// go round-robin over the XCDs, xcd_remap undoes that): the units of a slice share an XCD's L2 copy of the slice's operand rows,
// and every XCD gets its even share - at most one work-group per CU (a first mapping by slice % 8 gave four XCDs 36 work-groups
// for their 32 CUs: two rounds, 360 us).
template <int NT>
__global__ __launch_bounds__(NTHREADS, 1) void k_wgrad_strips(const Wg2Args a) {
  typedef Wg2Cfg<NT> C;
  constexpr int WP = C::WP, LDA = C::LDA, LDB = C::LDB, BK = WG2_BK;
  __shared__ __attribute__((aligned(16))) float smem[2 * C::STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int logical = xcd_remap((int)blockIdx.x, a.units * a.slices);
  const int slice = logical / a.units, unit = logical - slice * a.units;
  const int m_begin = slice * a.kchunk, m_end = min(m_begin + a.kchunk, a.rows);
  const int nt = (m_end - m_begin) / BK;   // K-steps (rows is a multiple of 16)

  // this wave's strip; the unit's (at most two) distinct problems: pa = the first strip's, pb = the last strip's
  const Wg2Strip st = a.strip[unit][wave];
  const int pa = a.strip[unit][0].problem;
  int pb = pa;
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (a.strip[unit][w].problem >= 0) pb = a.strip[unit][w].problem;
  const bool two = pb != pa;
  const int ai = (st.problem == pa || st.problem < 0) ? 0 : 1;   // which staged A tile this wave multiplies
  const bool active = st.problem >= 0;

  // ---- staging assignment: A tiles row-major [16][WP] as float4, thread f -> (row f / (WP/4), quad f % (WP/4))
  constexpr int QA = WP / 4;
  const Wg2Problem Pa = wg2_problem(a, pa), Pb = wg2_problem(a, pb);
  // operands through raw buffer resources (bufres.h): per-lane offsets fixed for the whole launch, the row block as a SCALAR
  // offset - a global_load's 64-bit address would be two VALU instructions per load, and with one wave per SIMD a VALU
  // instruction and the load that waits for it stall the MFMA stream (csrc/dgrad_rows.h)
  // (the resources start at the work-group's own K slice - 64-bit arithmetic, once - so the 32-bit offsets behind them span one
  // slice whatever the batch size)
  const int lda_a = Pa.lda, lda_b = Pb.lda;
  auto span = [&](int ld, int skip) __attribute__((always_inline)) {   // bytes from the slice's first row to the operand's end
    const size_t b = ((size_t)(a.rows - m_begin) * (size_t)ld - (size_t)skip) * 4;
    return (uint32_t)(b < 0xffffffffull ? b : 0xffffffffull);
  };
  const brsrc Ra = make_brsrc(Pa.A + (size_t)m_begin * lda_a, span(lda_a, 0)), Rb = make_brsrc(Pb.A + (size_t)m_begin * lda_b, span(lda_b, 0));
  // the B block this thread stages for j = 0, 1: strip (wave >> 1) + 2 j of the unit (idle strips repeat the first problem)
  brsrc RB0, RB1; int Bld[2];
  {
    const Wg2Strip s0 = a.strip[unit][(wave >> 1)], s1 = a.strip[unit][(wave >> 1) + 2];
    const Wg2Problem P0 = wg2_problem(a, s0.problem < 0 ? pa : s0.problem), P1 = wg2_problem(a, s1.problem < 0 ? pa : s1.problem);
    Bld[0] = P0.ldb; Bld[1] = P1.ldb;
    const int c0 = 32 * (s0.problem < 0 ? 0 : s0.ktile), c1 = 32 * (s1.problem < 0 ? 0 : s1.ktile);
    RB0 = make_brsrc(P0.B + (size_t)m_begin * P0.ldb + c0, span(P0.ldb, c0));
    RB1 = make_brsrc(P1.B + (size_t)m_begin * P1.ldb + c1, span(P1.ldb, c1));
  }
  // Operands stream from HBM and one work-group per CU has few loads in flight: the loads of a K-step are requested THREE steps
  // before they are stored to LDS (three register sets in rotation).  With one wave per SIMD every memory instruction's issue time
  // is exposed unless it sits right behind an MFMA, so the staging is cut into single instructions, one behind every STRIDE-th MFMA
  // (a burst of 8 loads or 8 LDS stores per thread between two k-pairs cost 30 us of the launch); the K-step stays ONE basic
  // block: no predicates (threads beyond a tile's last quad repeat it, idle waves repeat strip 0 and only skip their stores), the
  // two-problem units run their own instantiation of the loop.
  struct RSet { float4 a[C::NLA], a2[C::NLA], b[C::NLB]; };
  RSet r0, r1, r2;
  auto zero = [&](RSet& r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < C::NLA; ++j) r.a[j] = r.a2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < C::NLB; ++j) r.b[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  zero(r0); zero(r1); zero(r2);
  // staging pieces of a K-step, TWO = the unit spans two problems: [0, NLA) A tile, [NLA, 2 NLA) second A tile (TWO), then NLB B blocks
  int rowA[C::NLA], quadA[C::NLA];
#pragma unroll
  for (int j = 0; j < C::NLA; ++j) {
    const int f = min(tid + j * NTHREADS, C::NA4 - 1);
    rowA[j] = f / QA; quadA[j] = f - rowA[j] * QA;
  }
  const int rowB = (tid & 127) >> 3, quadB = tid & 7;
  uint32_t voA[C::NLA], voA2[C::NLA], voB[2];   // byte offsets of this thread's pieces inside a 16-row block
#pragma unroll
  for (int j = 0; j < C::NLA; ++j) {
    voA[j] = (uint32_t)((rowA[j] * lda_a + 4 * quadA[j]) * 4);
    voA2[j] = (uint32_t)((rowA[j] * lda_b + 4 * quadA[j]) * 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) voB[j] = (uint32_t)((rowB * Bld[j] + 4 * quadB) * 4);
  auto as4 = [](f32x4_b v) __attribute__((always_inline)) { return make_float4(v[0], v[1], v[2], v[3]); };
  auto gload_piece = [&](auto two_tag, RSet& r, int m0, int k) __attribute__((always_inline)) {
    constexpr bool TWO = decltype(two_tag)::value;
    constexpr int NA = TWO ? 2 * C::NLA : C::NLA;
    if (k < C::NLA) r.a[k] = as4(bload4(Ra, voA[k], (uint32_t)((m0 - m_begin) * lda_a) * 4u));
    else if (TWO && k < NA) r.a2[k - C::NLA] = as4(bload4(Rb, voA2[k - C::NLA], (uint32_t)((m0 - m_begin) * lda_b) * 4u));
    else if (k == NA) r.b[0] = as4(bload4(RB0, voB[0], (uint32_t)((m0 - m_begin) * Bld[0]) * 4u));
    else if (k == NA + 1) r.b[1] = as4(bload4(RB1, voB[1], (uint32_t)((m0 - m_begin) * Bld[1]) * 4u));
  };
  auto lstore_piece = [&](auto two_tag, const RSet& r, float* S, int k) __attribute__((always_inline)) {
    constexpr bool TWO = decltype(two_tag)::value;
    constexpr int NA = TWO ? 2 * C::NLA : C::NLA;
    if (k < C::NLA) *reinterpret_cast<float4*>(S + rowA[k] * LDA + 4 * quadA[k]) = r.a[k];
    else if (TWO && k < NA) *reinterpret_cast<float4*>(S + C::A_TILE + rowA[k - C::NLA] * LDA + 4 * quadA[k - C::NLA]) = r.a2[k - C::NLA];
    else if (k < NA + C::NLB)
      *reinterpret_cast<float4*>(S + 2 * C::A_TILE + ((wave >> 1) + 2 * (k - NA)) * C::B_TILE + rowB * LDB + 4 * quadB) = r.b[k - NA];
  };
  auto gload = [&](auto two_tag, RSet& r, int step) __attribute__((always_inline)) {
    constexpr int NPC = (decltype(two_tag)::value ? 2 * C::NLA : C::NLA) + C::NLB;
    const int m0 = m_begin + min(step, nt - 1) * BK;   // past the end: a harmless re-read
#pragma unroll
    for (int k = 0; k < NPC; ++k) gload_piece(two_tag, r, m0, k);
  };
  auto lstore = [&](auto two_tag, const RSet& r, int stage) __attribute__((always_inline)) {
    constexpr int NPC = (decltype(two_tag)::value ? 2 * C::NLA : C::NLA) + C::NLB;
#pragma unroll
    for (int k = 0; k < NPC; ++k) lstore_piece(two_tag, r, smem + stage * C::STAGE, k);
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fragment addresses (floats) inside a stage: A[k][32 t + l31], k = 2 g + lhi; B[k][l31]
  const int aoff = ai * C::A_TILE + lhi * LDA + l31;
  const int boff = 2 * C::A_TILE + wave * C::B_TILE + lhi * LDB + l31;

  // K-step i: MFMAs out of LDS stage i & 1; `nx` holds step i + 1: stored to the other stage (last read in step i - 1: every wave
  // is past that barrier) in the first half, refilled with step i + 4 in the second half, one instruction per piece slot
  auto kstep = [&](auto two_tag, int i, RSet& nx) __attribute__((always_inline)) {
    constexpr int NPC = (decltype(two_tag)::value ? 2 * C::NLA : C::NLA) + C::NLB;
    constexpr int NSLOT = (BK / 2) * NT, HALF = NSLOT / 2;
    constexpr int STRIDE = HALF / NPC >= 1 ? HALF / NPC : 1;
    static_assert(NPC <= HALF, "more staging pieces than MFMA slots");
    const float* S = smem + (i & 1) * C::STAGE;
    float* Sn = smem + ((i + 1) & 1) * C::STAGE;
    const int m4 = m_begin + min(i + 4, nt - 1) * BK;
    float fa0[NT], fa1[NT], fb0 = 0.f, fb1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) fa0[t] = S[aoff + 32 * t];
    fb0 = S[boff];
#pragma unroll
    for (int g = 0; g < BK / 2; ++g) {
      float (&ca)[NT] = (g & 1) ? fa1 : fa0;
      float (&na)[NT] = (g & 1) ? fa0 : fa1;
      const float cb = (g & 1) ? fb1 : fb0;
      float& nb = (g & 1) ? fb0 : fb1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!(WG2_DIAG & 1)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[t], cb, acc[t], 0, 0, 0);
        if (g + 1 < BK / 2 && !(WG2_DIAG & 8)) {
          if (t == 0) nb = S[boff + 2 * (g + 1) * LDB];
          na[t] = S[aoff + 2 * (g + 1) * LDA + 32 * t];
        }
        const int slot = g * NT + t;
        if (slot < HALF) {
          if (slot % STRIDE == 0 && slot / STRIDE < NPC && !(WG2_DIAG & 2)) lstore_piece(two_tag, nx, Sn, slot / STRIDE);
        } else {
          const int s2 = slot - HALF;
          if (s2 % STRIDE == 0 && s2 / STRIDE < NPC && !(WG2_DIAG & 4)) gload_piece(two_tag, nx, m4, s2 / STRIDE);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };
  auto run = [&](auto two_tag) __attribute__((always_inline)) {
    gload(two_tag, r0, 0);
    lstore(two_tag, r0, 0);
    gload(two_tag, r1, 1); gload(two_tag, r2, 2); gload(two_tag, r0, 3);
    __syncthreads();
    int i = 0;
    for (; i + 2 < nt; i += 3) {
      kstep(two_tag, i, r1);
      kstep(two_tag, i + 1, r2);
      kstep(two_tag, i + 2, r0);
    }
    if (i < nt) kstep(two_tag, i, r1);
    if (i + 1 < nt) kstep(two_tag, i + 1, r2);
  };
  if (nt > 0) {
    if (two) run(std::true_type{});
    else run(std::false_type{});
  }

  // ---- epilogue: the strip's NT tiles into this slice's slab of its problem
  if (!active) return;
  const Wg2Problem P = wg2_problem(a, st.problem);
  float* __restrict__ dst = P.slab + (size_t)slice * P.slab_stride + (size_t)(4 * lhi) * P.ldc + 32 * st.ktile + l31;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)(32 * t + (r & 3) + 8 * (r >> 2)) * P.ldc] = acc[t][r];
}

// Host side: the strip list of a step's problems, problem by problem, cut into units of four.  ktiles[p] = 32-wide column tiles of
// problem p's forward operand.  Returns the number of units (0: does not fit the tables).
inline int wg2_plan(const int* ktiles, int nproblems, Wg2Args& a) {
  int n = 0;
  for (int p = 0; p < nproblems; ++p) n += ktiles[p];
  const int units = (n + 3) / 4;
  if (nproblems > WG2_MAX_PROBLEMS || units > WG2_MAX_UNITS) return 0;
  int u = 0, w = 0;
  for (int p = 0; p < nproblems; ++p)
    for (int k = 0; k < ktiles[p]; ++k) {
      a.strip[u][w].problem = (int8_t)p; a.strip[u][w].ktile = (int8_t)k;
      if (++w == 4) { w = 0; ++u; }
    }
  for (; w > 0 && w < 4; ++w) { a.strip[u][w].problem = -1; a.strip[u][w].ktile = 0; }
  a.units = units;
  return units;
}

}  // namespace sdrm
