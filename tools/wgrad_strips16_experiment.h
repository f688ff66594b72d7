// EXPERIMENT (round 3, measured, not adopted): the strip-owned weight gradients (sdrm_amd/csrc/wgrad2.h) with strips half as
// wide - 22 accumulators of v_mfma_f32_16x16x4_f32 per wave, 18 units x 14 slices instead of 9 x 28, so half the slab bytes for
// the epilogue to write and for k_grad_finalize to read (-6.5 us there).  tools/wgrad2_probe.hip runs it beside the product
// kernel: 163 us against 156 us (profiles/r03_wgrad_strips16_probe.txt) - twice the barriers and twice the LDS fragment reads per
// MFMA cycle cost what the smaller slabs save.  (Through the MFMA builtin it was 179 us: the allocator renamed the accumulators
// inside the loop, 124 v_accvgpr_mov per three K-steps; the asm MFMA with the accumulator tied in place removes them.)
#pragma once
#include "../sdrm_amd/csrc/wgrad2.h"

namespace sdrm {

// The same launch with strips HALF as wide: all NT2 = WP / 16 row tiles (16 output rows each) of one 16-wide column tile, NT2
// accumulators of v_mfma_f32_16x16x4_f32 (4 registers each: 88 at WP = 352 instead of 176).  Twice the strips, so twice the
// units per K-slice and HALF THE SLICES for the same one-work-group-per-CU round (18 units x 14 slices at ML-1M): half the
// slab bytes to write at the end and for k_grad_finalize to read.  The [16][WP] operand tile is staged exactly as before (a unit
// now multiplies it with four 16-wide column blocks), so the staging traffic through L2 is unchanged.
template <int NT2>
struct Wg2Cfg16 {
  static constexpr int WP = 16 * NT2;
  // row stride of the staged A tile: the four lane groups of a 16x16x4 fragment read four consecutive rows, 16 floats each -
  // conflict-free when the stride is 16 or 48 (mod 64 banks)
  static constexpr int pad() { int p = 0; while (((WP + p) % 64) != 16 && ((WP + p) % 64) != 48) p += 4; return p; }
  static constexpr int LDA = WP + pad(), LDB = 16;
  static constexpr int A_TILE = WG2_BK * LDA, B_TILE = WG2_BK * LDB;
  static constexpr int STAGE = 2 * A_TILE + 4 * B_TILE;
  static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * 4;
  static constexpr int NA4 = WG2_BK * WP / 4;
  static constexpr int NLA = (NA4 + NTHREADS - 1) / NTHREADS;
  static_assert(4 * WG2_BK * 16 / 4 == NTHREADS, "one float4 of its wave's B block per thread");
  static_assert(LDS_BYTES <= 160 * 1024, "stages do not fit LDS");
};

template <int NT2>
__global__ __launch_bounds__(NTHREADS, 1) void k_wgrad_strips16(const Wg2Args a) {
  typedef Wg2Cfg16<NT2> C;
  constexpr int WP = C::WP, LDA = C::LDA, LDB = C::LDB, BK = WG2_BK;
  __shared__ __attribute__((aligned(16))) float smem[2 * C::STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int logical = xcd_remap((int)blockIdx.x, a.units * a.slices);
  const int slice = logical / a.units, unit = logical - slice * a.units;
  const int m_begin = slice * a.kchunk, m_end = min(m_begin + a.kchunk, a.rows);
  const int nt = (m_end - m_begin) / BK;

  const Wg2Strip st = a.strip[unit][wave];
  const int pa = a.strip[unit][0].problem;
  int pb = pa;
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (a.strip[unit][w].problem >= 0) pb = a.strip[unit][w].problem;
  const bool two = pb != pa;
  const int ai = (st.problem == pa || st.problem < 0) ? 0 : 1;
  const bool active = st.problem >= 0;

  constexpr int QA = WP / 4;
  const Wg2Problem Pa = wg2_problem(a, pa), Pb = wg2_problem(a, pb);
  const int lda_a = Pa.lda, lda_b = Pb.lda;
  auto span = [&](int ld, int skip) __attribute__((always_inline)) {
    const size_t b = ((size_t)(a.rows - m_begin) * (size_t)ld - (size_t)skip) * 4;
    return (uint32_t)(b < 0xffffffffull ? b : 0xffffffffull);
  };
  const brsrc Ra = make_brsrc(Pa.A + (size_t)m_begin * lda_a, span(lda_a, 0)), Rb = make_brsrc(Pb.A + (size_t)m_begin * lda_b, span(lda_b, 0));
  // the B block of this thread's own wave (idle strips repeat the unit's first problem, column block 0)
  const Wg2Problem Pw = wg2_problem(a, active ? st.problem : pa);
  const int ldb_w = Pw.ldb, cw = 16 * (active ? st.ktile : 0);
  const brsrc RBw = make_brsrc(Pw.B + (size_t)m_begin * ldb_w + cw, span(ldb_w, cw));

  struct RSet { float4 a[C::NLA], a2[C::NLA], b; };
  RSet r0, r1, r2;
  auto zero = [&](RSet& r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < C::NLA; ++j) r.a[j] = r.a2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    r.b = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  zero(r0); zero(r1); zero(r2);
  int rowA[C::NLA], quadA[C::NLA];
#pragma unroll
  for (int j = 0; j < C::NLA; ++j) {
    const int f = min(tid + j * NTHREADS, C::NA4 - 1);
    rowA[j] = f / QA; quadA[j] = f - rowA[j] * QA;
  }
  const int rowB = lane >> 2, quadB = lane & 3;
  uint32_t voA[C::NLA], voA2[C::NLA];
#pragma unroll
  for (int j = 0; j < C::NLA; ++j) {
    voA[j] = (uint32_t)((rowA[j] * lda_a + 4 * quadA[j]) * 4);
    voA2[j] = (uint32_t)((rowA[j] * lda_b + 4 * quadA[j]) * 4);
  }
  const uint32_t voB = (uint32_t)((rowB * ldb_w + 4 * quadB) * 4);
  auto as4 = [](f32x4_b v) __attribute__((always_inline)) { return make_float4(v[0], v[1], v[2], v[3]); };
  auto gload_piece = [&](auto two_tag, RSet& r, int m0, int k) __attribute__((always_inline)) {
    constexpr bool TWO = decltype(two_tag)::value;
    constexpr int NA = TWO ? 2 * C::NLA : C::NLA;
    if (k < C::NLA) r.a[k] = as4(bload4(Ra, voA[k], (uint32_t)((m0 - m_begin) * lda_a) * 4u));
    else if (TWO && k < NA) r.a2[k - C::NLA] = as4(bload4(Rb, voA2[k - C::NLA], (uint32_t)((m0 - m_begin) * lda_b) * 4u));
    else if (k == NA) r.b = as4(bload4(RBw, voB, (uint32_t)((m0 - m_begin) * ldb_w) * 4u));
  };
  auto lstore_piece = [&](auto two_tag, const RSet& r, float* S, int k) __attribute__((always_inline)) {
    constexpr bool TWO = decltype(two_tag)::value;
    constexpr int NA = TWO ? 2 * C::NLA : C::NLA;
    if (k < C::NLA) *reinterpret_cast<float4*>(S + rowA[k] * LDA + 4 * quadA[k]) = r.a[k];
    else if (TWO && k < NA) *reinterpret_cast<float4*>(S + C::A_TILE + rowA[k - C::NLA] * LDA + 4 * quadA[k - C::NLA]) = r.a2[k - C::NLA];
    else if (k == NA) *reinterpret_cast<float4*>(S + 2 * C::A_TILE + wave * C::B_TILE + rowB * LDB + 4 * quadB) = r.b;
  };
  auto gload = [&](auto two_tag, RSet& r, int step) __attribute__((always_inline)) {
    constexpr int NPC = (decltype(two_tag)::value ? 2 * C::NLA : C::NLA) + 1;
    const int m0 = m_begin + min(step, nt - 1) * BK;
#pragma unroll
    for (int k = 0; k < NPC; ++k) gload_piece(two_tag, r, m0, k);
  };
  auto lstore = [&](auto two_tag, const RSet& r, int stage) __attribute__((always_inline)) {
    constexpr int NPC = (decltype(two_tag)::value ? 2 * C::NLA : C::NLA) + 1;
#pragma unroll
    for (int k = 0; k < NPC; ++k) lstore_piece(two_tag, r, smem + stage * C::STAGE, k);
  };

  f32x4_b acc[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t) acc[t] = f32x4_b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT2; ++t) asm volatile("" : "+a"(acc[t]));   // every zero in ITS register before the first asm MFMA ...
  asm volatile("s_nop 7");                                         // ... and the wait states behind a VALU write

  // fragment addresses (floats) inside a stage: A[k][16 t + li], k = 4 g + lq; B[k][li]
  const int aoff = ai * C::A_TILE + lq * LDA + li;
  const int boff = 2 * C::A_TILE + wave * C::B_TILE + lq * LDB + li;

  auto kstep = [&](auto two_tag, int i, RSet& nx) __attribute__((always_inline)) {
    constexpr int NPC = (decltype(two_tag)::value ? 2 * C::NLA : C::NLA) + 1;
    constexpr int NG = BK / 4, NSLOT = NG * NT2, HALF = NSLOT / 2;
    constexpr int STRIDE = HALF / NPC >= 1 ? HALF / NPC : 1;
    static_assert(NPC <= HALF, "more staging pieces than MFMA slots");
    const float* S = smem + (i & 1) * C::STAGE;
    float* Sn = smem + ((i + 1) & 1) * C::STAGE;
    const int m4 = m_begin + min(i + 4, nt - 1) * BK;
    float fa0[NT2], fa1[NT2], fb0 = 0.f, fb1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT2; ++t) fa0[t] = S[aoff + 16 * t];
    fb0 = S[boff];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float (&ca)[NT2] = (g & 1) ? fa1 : fa0;
      float (&na)[NT2] = (g & 1) ? fa0 : fa1;
      const float cb = (g & 1) ? fb1 : fb0;
      float& nb = (g & 1) ? fb0 : fb1;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        // (asm with the accumulator tied in place: see rowchain.h - through the builtin the allocator renames the accumulators
        // inside the loop body, 124 v_accvgpr_mov per three K-steps)
        if (!(WG2_DIAG & 1)) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(ca[t]), "v"(cb));
        if (g + 1 < NG && !(WG2_DIAG & 8)) {
          if (t == 0) nb = S[boff + 4 * (g + 1) * LDB];
          na[t] = S[aoff + 4 * (g + 1) * LDA + 16 * t];
        }
        const int slot = g * NT2 + t;
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

  // ---- epilogue: lane (li, lq) holds of tile t the output rows 16 t + 4 lq .. + 3 at column li of the strip
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0]));   // MFMA results before their first VALU read (the asm MFMAs are opaque)
#pragma unroll
  for (int t = 1; t < NT2; ++t) asm volatile("" : "+a"(acc[t]));
  if (!active) return;
  const Wg2Problem P = wg2_problem(a, st.problem);
  float* __restrict__ dst = P.slab + (size_t)slice * P.slab_stride + (size_t)(4 * lq) * P.ldc + 16 * st.ktile + li;
#pragma unroll
  for (int t = 0; t < NT2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)(16 * t + r) * P.ldc] = acc[t][r];
}


}  // namespace sdrm
