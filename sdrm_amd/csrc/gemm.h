// fp32 MFMA GEMM for the SDRM eps-net on gfx950 (CDNA4).
//
// One kernel template covers the three contractions of a Linear layer
// (forward x*W^T, dgrad dY*W, wgrad dY^T*X) by choosing how each operand tile is
// brought into LDS.  Inside LDS both operands are always k-major
// (As[k][i], Bs[k][j]), which is the layout v_mfma_f32_32x32x2_f32 consumes with
// conflict-free ds_read_b32: lane l reads As[k0 + (l>>5)][i0 + (l&31)].
//
//   block tile BM x BN (template), K-step 16, 256 threads = 4 waves as WM x WN,
//   each wave (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each),
//   LDS double-buffered, one barrier per K-step, global->register prefetch of the next K-step
//   issued before the MFMAs of the current one, all fragments of a K-step read into registers
//   ahead of its MFMAs (the compiler places counted lgkmcnt waits).
//
// MFMAs are issued unconditionally: operand rows/cols beyond the matrix are zero-filled in LDS, and
// a work-group is as slow as its busiest wave anyway, so predicating whole MFMA tiles buys nothing
// and costs exec-mask branches around every MFMA.
//
// fp32-input MFMA is bit-for-bit a k-ordered fmaf chain, so results sit well inside the 1e-4
// normwise parity bar.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.h"

namespace sdrm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int { LD_KCONTIG = 0, LD_MCONTIG = 1 };  // operand element (i,k) at src[i*ld+k]  /  src[k*ld+i]
enum : int { XF_NONE = 0, XF_PRELU = 1 };
enum : int {
  EPI_BIAS = 0,          // C = acc + bias[n]                         (hidden pre-activations)
  EPI_BIAS_TANH = 1,     // C = tanh(acc + bias[n])                   (eps-net output)
  EPI_DPRELU = 3,        // C = acc * prelu'(aux) ; partial sum of acc*min(aux,0) (slope gradient)
  EPI_SLAB = 4,          // C = acc into split-K slab blockIdx.z ; optional column sums (bias grad)
  EPI_PLAIN = 5          // C = acc (debug)
};

constexpr int NTHREADS = 256;

// MF = edge of the MFMA output tile: 32 -> v_mfma_f32_32x32x2_f32 (16 accumulator VGPRs per tile), 16 ->
// v_mfma_f32_16x16x4_f32 (4 VGPRs).  Same flop rate; the 16-wide form lets a 32x32 block tile be split over four
// waves, which is what a launch with few rows needs: it cannot fill the chip with 64x64 tiles, and a lone block's
// time is its per-wave chain of dependent MFMAs (K/2 x 64 cycles for a 32x32 tile, K/4 x 32 for a 16x16 one).
template <int BM_, int BN_, int WM_, int WN_, int MINW_ = 4, int BK_ = 16, int MF_ = 32>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_, MF = MF_;
  static constexpr int MINW = MINW_;   // waves per SIMD the register allocator must leave room for
  static constexpr int TM = BM_ / WM_ / MF_, TN = BN_ / WN_ / MF_;   // MFMA tiles per wave
  static constexpr int PADMAX = (MF_ == 32) ? 4 : 16;
  static constexpr int LDA = BM_ + PADMAX, LDB = BN_ + PADMAX;     // upper bound of the LDS row strides (floats)
  static constexpr int STAGE = BK * (LDA + LDB);                   // floats per pipeline stage
  static_assert(WM_ * WN_ == 4, "4 waves per work-group");
  static_assert(MF_ == 32 || MF_ == 16, "MFMA tile edge");
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one MFMA tile");
};

template <int MF> struct AccT;
template <> struct AccT<32> { typedef float type __attribute__((ext_vector_type(16))); };
template <> struct AccT<16> { typedef float type __attribute__((ext_vector_type(4))); };

struct GemmArgs {
  const float* A; int lda; int limA;   // limA: valid extent of A's output-side index (multiple of 32)
  const float* B; int ldb; int limB;
  float* C; int ldc;
  int K;            // full reduction length (multiple of 32)
  int kchunk;       // reduction range handled by one blockIdx.z (multiple of 32)
  int tiles_n;      // number of tiles along N (grid.x = tiles_m * tiles_n, XCD-swizzled)
  int nblocks;      // tiles_m * tiles_n
  int nsplits;      // K-slices (EPI_SLAB only; 1 otherwise)
  const float* bias;
  const float* slopeA;  // PReLU slope applied to A elements on load (XF_PRELU)
  const float* slopeB;
  // EPI_DPRELU
  const float* aux; int ldaux; const float* slopeE; float* slope_partial;
  // EPI_SLAB
  size_t slab_stride; float* dbias; int dbias_stride;
  unsigned long long* stamps;   // diagnostic builds only (-DSDRM_STAMPS): 4 s_memtime stamps per block
  // EPI_BIAS_TANH: output may be an unpadded caller buffer
  int rows_valid, cols_valid;
};

__device__ __forceinline__ float prelu_f(float v, float a) { return v > 0.f ? v : a * v; }

// tanh through the hardware exp2/rcp: 1 - 2/(e^{2x}+1).  Absolute error ~1e-7 (the eps-net output is
// compared at 1e-4 of max|y| ~ 0.3); libm's tanhf costs ~30 instructions per element, which on a
// 64-element-per-thread epilogue is microseconds per work-group.
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);   // e^{2x} = 2^{2x*log2(e)}
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// Global -> registers for one operand tile (ROWS x BK elements, ROWS = BM or BN).  No bounds checks:
// every operand buffer is allocated with its tiled extent (rows rounded up to the tile, plus a slack
// tail), rows/cols beyond the matrix feed only output rows/cols the epilogue discards, and weight pad
// rows are zero.  Branch-free loads are what lets the compiler keep them in flight behind counted waits.
template <int LOAD, int ROWS, int BK>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int ld, int i0, int k0,
                                          float4 (&r)[ROWS * BK / 4 / NTHREADS], int tid) {
  constexpr int NV = ROWS * BK / 4 / NTHREADS;
  constexpr int KQ = BK / 4;   // float4 per row of a k-contiguous operand
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const int f = tid + s * NTHREADS;
    if (LOAD == LD_KCONTIG) {
      const int i = f / KQ, kq = f % KQ;
      r[s] = *reinterpret_cast<const float4*>(src + (size_t)(i0 + i) * ld + k0 + 4 * kq);
    } else {
      constexpr int VPR = ROWS / 4;  // float4 per k-row
      const int k = f / VPR, iq = f - k * VPR;
      r[s] = *reinterpret_cast<const float4*>(src + (size_t)(k0 + k) * ld + i0 + 4 * iq);
    }
  }
}

// Registers -> LDS (k-major, row stride LD).  The operand transform (PReLU of stored pre-activations)
// is applied here, not at load time, so the global loads stay in flight across two K-steps.
template <int LOAD, int XF, int ROWS, int LD, int BK>
__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float4 (&r)[ROWS * BK / 4 / NTHREADS],
                                           float slope, int tid) {
  constexpr int NV = ROWS * BK / 4 / NTHREADS;
  constexpr int KQ = BK / 4;
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const int f = tid + s * NTHREADS;
    float4 v = r[s];
    if (XF == XF_PRELU) {
      v.x = prelu_f(v.x, slope); v.y = prelu_f(v.y, slope); v.z = prelu_f(v.z, slope); v.w = prelu_f(v.w, slope);
    }
    if (LOAD == LD_KCONTIG) {
      const int i = f / KQ, kq = f % KQ;
      float* d = dst + (4 * kq) * LD + i;
      d[0] = v.x; d[LD] = v.y; d[2 * LD] = v.z; d[3 * LD] = v.w;
    } else {
      constexpr int VPR = ROWS / 4;
      const int k = f / VPR, iq = f - k * VPR;
      *reinterpret_cast<float4*>(dst + k * LD + 4 * iq) = v;
    }
  }
}

// XCD-aware remap (cdna guide T1, bijective form): hardware deals consecutive block ids round-robin
// over the 8 XCDs; give every XCD a contiguous range of logical tiles so that the N-tiles of one
// M-tile (which re-read the same activation rows) share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

template <class Cfg, int LOADA, int LOADB, int XFA, int XFB, int EPI>
__global__ __launch_bounds__(NTHREADS, Cfg::MINW) void gemm_kernel(const GemmArgs p) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN, MF = Cfg::MF;
  constexpr int NR = (MF == 32) ? 16 : 4;      // accumulator registers per MFMA tile
  constexpr int KG = (MF == 32) ? 2 : 4;       // k consumed by one MFMA
  typedef typename AccT<MF>::type acc_t;
  // LDS row strides.  32-wide MFMA: +2 floats for transposing (k-contiguous) stores, which makes the 4*kq*LD + i
  // bank pattern of a half-wave at most 2-way (free) for BK = 16 and 32; +4 keeps ds_write_b128 rows 16-byte
  // aligned.  16-wide MFMA: a fragment read touches rows k+0..3 at 16 consecutive floats each, conflict-free
  // when LD = 16 (mod 32).
  constexpr int LDA = BM + (MF == 16 ? 16 : (LOADA == LD_KCONTIG ? 2 : 4));
  constexpr int LDB = BN + (MF == 16 ? 16 : (LOADB == LD_KCONTIG ? 2 : 4));
  constexpr int NVA = BM * BK / 4 / NTHREADS, NVB = BN * BK / 4 / NTHREADS;
  static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 loader threads");
  __shared__ __attribute__((aligned(16))) float smem[2 * Cfg::STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  // lane -> (index within the tile edge, k sub-index of the MFMA): 32x32x2 has 32 x 2, 16x16x4 has 16 x 4
  const int l31 = (MF == 32) ? (lane & 31) : (lane & 15), lhi = (MF == 32) ? (lane >> 5) : (lane >> 4);

  int logical, split;
  if (EPI == EPI_SLAB) {
    // split-K launch: all tiles of one K-slice go to ONE XCD (blocks are dealt round-robin over the 8
    // XCDs), so the slice's two operand strips (a few MB) are fetched from HBM once and re-read from that
    // XCD's L2 by the other tiles.  Spread over XCDs they were fetched 4.6x (measured: FETCH_SIZE).
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / p.nblocks) * 8 + x;
    logical = slot % p.nblocks;
    if (split >= p.nsplits) return;
  } else {
    logical = xcd_remap(blockIdx.x, p.nblocks);
    split = 0;
  }
  const int tile_m = logical / p.tiles_n, tile_n = logical - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kb = split * p.kchunk;
  const int ke = min(kb + p.kchunk, p.K);

  const float slopeA = (XFA == XF_PRELU) ? *p.slopeA : 0.f;
  const float slopeB = (XFB == XF_PRELU) ? *p.slopeB : 0.f;

  acc_t acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[a][b][r] = 0.f;

  // Two register sets hold the K-steps i+1 and i+2 while step i is multiplied out of LDS: every global
  // load has two full K-steps to land (an HBM round trip is longer than one 16-deep step of MFMAs).
  float4 ra0[NVA], rb0[NVB], ra1[NVA], rb1[NVB];
  float dbsum = 0.f;
  const bool do_dbias = (EPI == EPI_SLAB) && (p.dbias != nullptr) && (tile_n == 0) && (tid < BM);
  const int nt = (ke - kb + BK - 1) / BK;   // K-steps of this block (>= 0)

  const int aoff = lhi * LDA + wm * (BM / Cfg::WM) + l31;
  const int boff = lhi * LDB + wn * (BN / Cfg::WN) + l31;

  auto ld = [&](float4 (&xa)[NVA], float4 (&xb)[NVB], int i) {
    const int k0 = kb + min(i, nt - 1) * BK;   // past the end: re-read the last K-step (never consumed)
    load_tile<LOADA, BM, BK>(p.A, p.lda, m0, k0, xa, tid);
    load_tile<LOADB, BN, BK>(p.B, p.ldb, n0, k0, xb, tid);
  };
  auto st = [&](const float4 (&xa)[NVA], const float4 (&xb)[NVB], int stage) {
    float* An = smem + stage * Cfg::STAGE;
    store_tile<LOADA, XFA, BM, LDA, BK>(An, xa, slopeA, tid);
    store_tile<LOADB, XFB, BN, LDB, BK>(An + BK * LDA, xb, slopeB, tid);
  };
  // A wave whose whole 32-granular tile range lies outside the matrix (the half-empty last tile row /
  // column: 352 = 5.5 x 64) issues no LDS reads and no MFMAs; it still loads, stores and meets the
  // barriers.  Its SIMD's matrix pipe goes to the other work-groups resident on the CU.
  const bool wave_active = (m0 + wm * (BM / Cfg::WM) < p.limA) && (n0 + wn * (BN / Cfg::WN) < p.limB);
  // compute() is split in two so that a K-step can issue its first fragment reads right after the
  // barrier, ahead of the LDS stores of the next step's operands (they queue in order on the LDS
  // pipe): the first MFMA then waits ~one LDS latency instead of stores + latency.
  float af[2][TM], bf[2][TN];
  auto first_frags = [&](int stage) {
    const float* As = smem + stage * Cfg::STAGE;
    const float* Bs = As + BK * LDA;
    if (wave_active) {
#pragma unroll
      for (int a = 0; a < TM; ++a) af[0][a] = As[aoff + MF * a];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[0][b] = Bs[boff + MF * b];
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](int stage) {
    const float* As = smem + stage * Cfg::STAGE;
    const float* Bs = As + BK * LDA;
    if (wave_active) {
      // Fragment reads run one MFMA group ahead of their use (two register sets); sched_barrier pins the
      // issue order so the LDS latency of group s+1 hides under the MFMAs of group s.
#pragma unroll
      for (int s = 0; s < BK / KG; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < BK / KG) {
#pragma unroll
          for (int a = 0; a < TM; ++a) af[nxt][a] = As[aoff + KG * (s + 1) * LDA + MF * a];
#pragma unroll
          for (int b = 0; b < TN; ++b) bf[nxt][b] = Bs[boff + KG * (s + 1) * LDB + MF * b];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            if constexpr (MF == 32) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
            else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (do_dbias) {
#pragma unroll
      for (int k = 0; k < BK; ++k) dbsum += As[k * LDA + tid];
    }
  };

  if (nt > 0) {
    ld(ra0, rb0, 0);
    st(ra0, rb0, 0);
    ld(ra0, rb0, 1);
    ld(ra1, rb1, 2);
    __syncthreads();
#ifdef SDRM_STAMPS
    if (EPI == EPI_PLAIN) t_pro = __builtin_amdgcn_s_memtime();
#endif
    int i = 0;
    for (; i + 1 < nt; i += 2) {
      first_frags(0);
      st(ra0, rb0, 1);          // set 0 holds K-step i+1
      ld(ra0, rb0, i + 3);
      compute(0);
      __syncthreads();
      first_frags(1);
      st(ra1, rb1, 0);          // set 1 holds K-step i+2
      ld(ra1, rb1, i + 4);
      compute(1);
      __syncthreads();
    }
    if (i < nt) { first_frags(0); compute(0); }   // odd count: the last K-step already sits in stage 0
  }
#ifdef SDRM_STAMPS
  if (EPI == EPI_PLAIN) t_loop = __builtin_amdgcn_s_memtime();
#endif

  // ------------------------------------------------------------------ epilogue
  // accumulator register r of tile (a,b), 32-wide: row = m0 + wm*(BM/WM) + a*32 + (r&3) + 8*(r>>2) + 4*lhi,
  //                                                 col = n0 + wn*(BN/WN) + b*32 + l31
  //                                        16-wide: row = ... + a*16 + 4*lhi + r ; col = ... + b*16 + l31
  auto rowoff = [](int r) { return (MF == 32) ? (r & 3) + 8 * (r >> 2) : r; };
  float slope_sum = 0.f;
  const float slopeE = (EPI == EPI_DPRELU) ? *p.slopeE : 0.f;
  float* __restrict__ Cp = p.C;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int tm0 = m0 + wm * (BM / Cfg::WM) + a * MF, tn0 = n0 + wn * (BN / Cfg::WN) + b * MF;
      __builtin_amdgcn_sched_barrier(0);   // one tile's addresses live at a time (keeps the kernel at <=128 VGPRs)
      if (tm0 >= p.limA || tn0 >= p.limB) continue;  // wave-uniform: tile entirely outside the matrix
      const int col = tn0 + l31;
      const int rbase = tm0 + 4 * lhi;
      float bias = 0.f;
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) bias = p.bias[col];
      if (EPI == EPI_BIAS || EPI == EPI_PLAIN) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          Cp[(size_t)row * p.ldc + col] = acc[a][b][r] + bias;
        }
      } else if (EPI == EPI_BIAS_TANH) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          if (row < p.rows_valid && col < p.cols_valid) Cp[(size_t)row * p.ldc + col] = tanh_fast(acc[a][b][r] + bias);
        }
      } else if (EPI == EPI_DPRELU) {
        const float* __restrict__ auxp = p.aux;
        float pre[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) pre[r] = auxp[(size_t)(rbase + rowoff(r)) * p.ldaux + col];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          const float v = acc[a][b][r];
          const bool pos = pre[r] > 0.f;
          Cp[(size_t)row * p.ldc + col] = pos ? v : slopeE * v;
          slope_sum += pos ? 0.f : v * pre[r];
        }
      } else if (EPI == EPI_SLAB) {
        float* __restrict__ Sp = p.C + (size_t)split * p.slab_stride;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          Sp[(size_t)row * p.ldc + col] = acc[a][b][r];
        }
      }
    }
  }

#ifdef SDRM_STAMPS
  if (EPI == EPI_PLAIN && p.stamps && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* o = p.stamps + 4 * (size_t)blockIdx.x;
    o[0] = t_in; o[1] = t_pro; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
  }
#endif
  if (EPI == EPI_DPRELU) {
    // block-wide sum of the slope-gradient partial -> one float per block (deterministic order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) slope_sum += __shfl_down(slope_sum, off, 64);
    __syncthreads();
    if (lane == 0) smem[wave] = slope_sum;
    __syncthreads();
    if (tid == 0) p.slope_partial[blockIdx.x] = smem[0] + smem[1] + smem[2] + smem[3];
  }
  if (EPI == EPI_SLAB) {
    if (do_dbias && (m0 + tid) < p.limA) p.dbias[(size_t)split * p.dbias_stride + m0 + tid] = dbsum;
  }
}

}  // namespace sdrm
