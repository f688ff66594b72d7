// fp32 MFMA GEMM for the SDRM eps-net on gfx950 (CDNA4).
//
// One kernel template covers the contractions of a Linear layer (forward x*W^T, dgrad dY*(W^T)^T against the
// transposed weight copy, wgrad dY^T*X) by choosing how each operand tile is brought into LDS:
//
//   NT GEMMs (both operands k-contiguous in memory: every forward and every dgrad) keep the tiles k-MINOR in
//   LDS - As[i][k], row stride BK+4 floats: the float4 a thread loaded is stored as it is (ds_write_b128) and a
//   lane fetches its fragments of a whole K-step with BK/8 ds_read_b128 per MFMA tile edge.  The MFMA
//   contraction does not care which k a (group, lane-half) slot carries as long as A and B agree.
//   wgrad (both operands row-major over the reduction index) and the 16-wide-MFMA tiles keep them k-MAJOR,
//   As[k][i], read with conflict-free ds_read_b32: lane l reads As[k0 + (l>>5)][i0 + (l&31)].
//
//   block tile BM x BN (template), K-step BK, 256 threads = 4 waves as WM x WN, each wave TM x TN MFMA tiles of
//   32x32 (v_mfma_f32_32x32x2_f32, 16 accumulator VGPRs) or 16x16 (v_mfma_f32_16x16x4_f32, 4 VGPRs).
//
// Main loop (both layouts): two LDS stages, ONE barrier per K-step, fragments double-buffered in registers.
// A wave issues in order and its MFMAs are one dependent chain per accumulator tile, so everything else of
// the pipeline is cut into small pieces that ride in the 64-cycle shadow behind each MFMA issue (order pinned
// with sched_barrier): fragment reads of step i+1, LDS stores of step i+2 out of the prefetch registers, global
// loads of step i+4.  Measured on a lone work-group (tools/gemm_stamps.py): 715 -> 597 cycles per 8 MFMAs
// (512 is the matrix pipe itself); full launches 24576x352x352: 58.7 -> 55.2 us (110 TFLOP/s padded).
//
// MFMAs are issued unconditionally inside an active wave: operand rows/cols beyond the matrix are zero-filled
// in LDS.  A wave whose whole tile range lies outside the matrix skips reads and MFMAs (it only feeds the
// pipeline), which returns its SIMD's matrix pipe to the other work-groups on the CU.
//
// fp32-input MFMA is bit-for-bit a k-ordered fmaf chain, so results sit well inside the 1e-4
// normwise parity bar.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bufres.h"

#include <type_traits>

#include "philox.h"
#include "select.h"

namespace sdrm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int { LD_KCONTIG = 0, LD_MCONTIG = 1 };  // operand element (i,k) at src[i*ld+k]  /  src[k*ld+i]
enum : int { XF_NONE = 0, XF_PRELU = 1 };
enum : int {
  EPI_BIAS = 0,          // C = acc + bias[n]                         (hidden pre-activations)
  EPI_BIAS_TANH = 1,     // C = tanh(acc + bias[n])                   (eps-net output into the padded Y buffer)
  EPI_BIAS_TANH_G = 2,   // same, bounds-checked: the output is an unpadded caller buffer (sdrm_forward)
  EPI_DPRELU = 3,        // C = acc * prelu'(aux) ; partial sum of acc*min(aux,0) (slope gradient)
  EPI_SLAB = 4,          // C = acc into split-K slab blockIdx.z ; optional column sums (bias grad)
  EPI_PLAIN = 5,         // C = acc (debug)
  EPI_TANH_REV = 6,      // eps_hat = tanh(acc + bias) feeds the DDPM reverse update of the sampler state in place
                         // (k_reverse_update fused: on-device Philox, full-resolution sampling)
  EPI_BIAS_G = 7,        // C = acc + bias[n], bounds-checked into an unpadded caller buffer (VAE decode output layer)
  EPI_BIAS_ROWTAB = 9,   // C = acc + tab[t(row)][n]: layer 0 of the train step, whose bias + time-embedding term b0 + C0[t] is a
                         // row of the per-step table B0tab picked by the row's timestep (the stacked passes P, S, Q share t)
  EPI_BIAS_PRELU = 10    // C = prelu(acc + bias[n]; slopeE): the sampler's hidden layers store the ACTIVATION - nothing reads their
                         // pre-activations again (no backward), and the next layer then loads its operand without the PReLU-on-load
                         // transform, which costs the 5429-row launches ~1.5 us each (12.7 us plain against 16 us with it)
};

constexpr int NTHREADS = 256;

// diagnostic builds only (tools/gemm_stamps.py -DSDRM_DIAG=mask): drop a piece of the NT main loop to see what
// it costs a work-group - bit0 loop loads, bit1 LDS stores, bit2 barriers
#ifndef SDRM_DIAG
#define SDRM_DIAG 0
#endif
#define DIAG_LD(x) do { if (!(SDRM_DIAG & 1)) { x; } } while (0)
#define DIAG_ST(x) do { if (!(SDRM_DIAG & 2)) { x; } } while (0)
#define DIAG_BAR() do { if (!(SDRM_DIAG & 4)) __syncthreads(); } while (0)

// MF = edge of the MFMA output tile: 32 -> v_mfma_f32_32x32x2_f32 (16 accumulator VGPRs per tile), 16 ->
// v_mfma_f32_16x16x4_f32 (4 VGPRs).  Same flop rate; the 16-wide form lets a 32x32 block tile be split over four
// waves, which is what a launch with few rows needs: it cannot fill the chip with 64x64 tiles, and a lone block's
// time is its per-wave chain of dependent MFMAs (K/2 x 64 cycles for a 32x32 tile, K/4 x 32 for a 16x16 one).
// PF = global-prefetch register sets of the NT loop: 2 (loads requested four K-steps ahead of their MFMAs) or 1 (three
// ahead; what a 32-deep K-step can afford inside 128 VGPRs, and its step is twice as long anyway).
template <int BM_, int BN_, int WM_, int WN_, int MINW_ = 4, int BK_ = 16, int MF_ = 32, int PF_ = 2>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_, MF = MF_, PF = PF_;
  static constexpr int MINW = MINW_;   // waves per SIMD the register allocator must leave room for
  static constexpr int TM = BM_ / WM_ / MF_, TN = BN_ / WN_ / MF_;   // MFMA tiles per wave
  static constexpr int PADMAX = (MF_ == 32) ? 4 : 16;
  static constexpr int LDA = BM_ + PADMAX, LDB = BN_ + PADMAX;     // upper bound of the LDS row strides (floats)
  static constexpr int LDK = BK_ + 4;                              // row stride of the k-minor layout (NT GEMMs)
  static constexpr int STAGE_KMAJOR = BK * (LDA + LDB), STAGE_KMINOR = (BM_ + BN_) * LDK;
  static constexpr int STAGE = STAGE_KMAJOR > STAGE_KMINOR ? STAGE_KMAJOR : STAGE_KMINOR;   // floats per pipeline stage
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
  uint32_t magic_tiles_n, magic_nblocks;   // floor(2^32 / d) + 1 for d = tiles_n / nblocks (0 when d == 1): the work-group's tile
                    // coordinates come from two scalar multiplies instead of two emulated integer divisions (gemm_magic below)
  const float* bias;
  const float* slopeA;  // PReLU slope applied to A elements on load (XF_PRELU)
  const float* slopeB;
  // EPI_DPRELU
  const float* aux; int ldaux; const float* slopeE; float* slope_partial;
  // EPI_SLAB
  size_t slab_stride; float* dbias; int dbias_stride;
  unsigned long long* stamps;   // diagnostic builds only (-DSDRM_STAMPS): 8 slots per block: 4 s_memtime stamps, 2 s_memrealtime
  // EPI_BIAS_TANH_G: extent of the unpadded caller buffer
  int rows_valid, cols_valid;
  // EPI_TANH_REV: sampler state X [rows][ldx] (updated in place), next step's dropped-out input U [rows][ldx];
  // rows of this launch are slots rev_s0 .. of the sampler, global row (Philox key) = rev_row0 + slot
  float* revX; float* revU; int rev_ldx, rev_s0, rev_n, rev_L, rev_step;
  float rev_c1, rev_sqrt_alpha, rev_sqrt_beta, rev_nd;
  uint32_t rev_seed_lo, rev_seed_hi, rev_call_id; int64_t rev_row0;
  const int* rev_rowid;   // multi-resolution sampling: slot -> original row (the Philox key is rev_row0 + that row); null: the slot itself
  // EPI_BIAS_ROWTAB: `bias` is the table [T+1][ldtab]; stacked row r (< 3 * trow_B) belongs to user r mod trow_B, whose timestep
  // is trow[user]; pad rows use t = 0
  const int* trow; int trow_B, ldtab;
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
// The loads are raw buffer loads (bufres.h): resource = the operand from the tile's first row / column on, per-thread byte
// offsets fixed for the whole launch (tile_offsets), the K position as a SCALAR offset - a global_load's moving 64-bit address is
// a VALU add per load, and the in-order wave stalls for the add and for the load that waits for it (csrc/dgrad_rows.h).
template <int LOAD, int ROWS, int BK>
__device__ __forceinline__ void tile_offsets(int ld, uint32_t (&vo)[ROWS * BK / 4 / NTHREADS], int tid) {
  constexpr int NV = ROWS * BK / 4 / NTHREADS;
  constexpr int KQ = BK / 4;   // float4 per row of a k-contiguous operand
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const int f = tid + s * NTHREADS;
    if (LOAD == LD_KCONTIG) {
      const int i = f / KQ, kq = f % KQ;
      vo[s] = (uint32_t)(i * ld + 4 * kq) * 4u;
    } else {
      constexpr int VPR = ROWS / 4;  // float4 per k-row
      const int k = f / VPR, iq = f - k * VPR;
      vo[s] = (uint32_t)(k * ld + 4 * iq) * 4u;
    }
  }
}
// The resource starts at the work-group's own corner of the operand - first row / column of its tile AND first k of its K range
// (64-bit arithmetic, once) - so the 32-bit offsets behind it only span one tile x one K chunk, whatever the operand's size.
template <int LOAD, int ROWS, int BK>
__device__ __forceinline__ brsrc tile_resource(const float* __restrict__ src, int ld, int i0, int kb) {
  return make_brsrc(LOAD == LD_KCONTIG ? src + (size_t)i0 * ld + kb : src + (size_t)kb * ld + i0, 0xffffffffu);
}
// AUX: cache-policy bits of the loads (bufres.h; BUF_SC1 where another work-group of the SAME launch wrote the operand: csrc/sample_persist.h)
template <int LOAD, int ROWS, int BK, int AUX = 0>
__device__ __forceinline__ void load_tile(brsrc res, const uint32_t (&vo)[ROWS * BK / 4 / NTHREADS], int ld, int k0 /* from the resource's first k */,
                                          float4 (&r)[ROWS * BK / 4 / NTHREADS]) {
  constexpr int NV = ROWS * BK / 4 / NTHREADS;
  const uint32_t so = (LOAD == LD_KCONTIG ? (uint32_t)k0 : (uint32_t)k0 * (uint32_t)ld) * 4u;
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const f32x4_b v = bload4a<AUX>(res, vo[s], so);
    r[s] = make_float4(v[0], v[1], v[2], v[3]);
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

template <int NV>
__device__ __forceinline__ void prelu_regs(float4 (&r)[NV], float slope) {
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    r[s].x = prelu_f(r[s].x, slope); r[s].y = prelu_f(r[s].y, slope);
    r[s].z = prelu_f(r[s].z, slope); r[s].w = prelu_f(r[s].w, slope);
  }
}

// k-minor variant (both operands k-contiguous in global memory, i.e. the NT GEMMs): the float4 a thread
// loaded goes to LDS as one ds_write_b128 at [row][4*kq], row stride BK+4 floats (conflict-free for the
// 16-lane groups of a b128 access: 16 B x odd multiple).
template <int XF, int ROWS, int LDK, int BK>
__device__ __forceinline__ void store_tile_km(float* __restrict__ dst, const float4 (&r)[ROWS * BK / 4 / NTHREADS],
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
    const int i = f / KQ, kq = f % KQ;
    *reinterpret_cast<float4*>(dst + i * LDK + 4 * kq) = v;
  }
}

// XCD-aware remap (cdna guide T1, bijective form): hardware deals consecutive block ids round-robin
// over the 8 XCDs; give every XCD a contiguous range of logical tiles so that the N-tiles of one
// M-tile (which re-read the same activation rows) share an L2.
// n / d for 0 <= n, n * d < 2^32, with m = floor(2^32 / d) + 1 (d >= 2) or 0 (d == 1): exact (the error term n * (m * d - 2^32)
// stays below 2^32).  The launch helpers fill the magics; an emulated 32-bit division is ~25 dependent scalar / vector
// instructions at the head of every work-group, where nothing else can be issued yet.
__host__ __device__ __forceinline__ uint32_t gemm_magic(int d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint32_t)d) + 1u; }
__device__ __forceinline__ int div_magic(int n, uint32_t m) { return m ? (int)__umulhi((uint32_t)n, m) : n; }

// The grid bookkeeping of a launch in ONE place, so that no caller can set a tile count without its magic: tiles along N, tiles
// in all, K-slices and the two magics.  The magic division n / d is exact only while n * d < 2^32 - n ranges over the
// tiles (d = tiles_n) and over slices x tiles (d = nblocks) - so a grid beyond that is refused (false) instead of
// letting work-groups silently compute wrong tiles.
__host__ inline bool gemm_set_grid(GemmArgs& a, int tiles_m, int tiles_n, int splits) {
  if (tiles_m < 1 || tiles_n < 1 || splits < 1) return false;
  const uint64_t nb = (uint64_t)tiles_m * (uint64_t)tiles_n;
  if (nb >= (1ull << 31) || nb * (uint64_t)tiles_n >= (1ull << 32) || nb * nb * (uint64_t)splits >= (1ull << 32)) return false;
  a.tiles_n = tiles_n; a.nblocks = (int)nb; a.nsplits = splits;
  a.magic_tiles_n = gemm_magic(a.tiles_n); a.magic_nblocks = gemm_magic(a.nblocks);
  return true;
}

__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// The work-group body; `bid` is the work-group's index inside ITS problem (== blockIdx.x for a plain launch).
// value of lane (l ^ 1) / (l ^ 2) of the same quad (DPP quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E), and of quad lane K
template <int CTRL>
__device__ __forceinline__ float quad_xor(float v) {
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, true));
}
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, K * 0x55, 0xF, 0xF, true);
}

// COHA: cache policy of the A operand's loads; logical_in >= 0: the caller's tile index (instead of the XCD remap of `bid`).
template <class Cfg, int LOADA, int LOADB, int XFA, int XFB, int EPI, int COHA = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bid, const int logical_in = -1) {
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
  // NT GEMMs (both operands k-contiguous: every forward, and dgrad against the transposed weight copy) keep
  // the operands k-MINOR in LDS: stores are the loaded float4 as they are, and a lane fetches its whole K-step
  // of fragments with BK/8 ds_read_b128 per MFMA tile edge instead of BK/2 ds_read_b32.  The MFMA contraction
  // does not care which k a (group, lane-half) slot carries as long as A and B agree: lane-half h takes
  // k = h*BK/2 .. h*BK/2 + BK/2-1 of the K-step.
  // (16-wide MFMA: lane group q = lane>>4 takes k = 16u + 4q + e of the K-step for MFMA (u, e), u < BK/16 - the same
  //  permutation trick, one ds_read_b128 per u)
  constexpr bool KM = (LOADA == LD_KCONTIG && LOADB == LD_KCONTIG);
  constexpr int LDK = Cfg::LDK;
  constexpr int BOFF = KM ? BM * LDK : BK * LDA;   // B operand's offset inside a stage
  constexpr int NQ = (MF == 32) ? BK / 8 : BK / 16;   // float4 fragments per lane per MFMA tile edge per K-step
  constexpr int QSTEP = (MF == 32) ? 4 : 16;          // distance (floats) between a lane's consecutive fragments
  constexpr int NVA = BM * BK / 4 / NTHREADS, NVB = BN * BK / 4 / NTHREADS;
  static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 loader threads");
  __shared__ __attribute__((aligned(16))) float smem[2 * Cfg::STAGE];
#ifdef SDRM_STAMPS
  unsigned long long t_in = __builtin_amdgcn_s_memtime(), t_pro = 0, t_loop = 0;
  unsigned long long t_ld = 0, t_land = 0;   // -DSDRM_STAMPS=2: inside the NT prologue
  const unsigned long long r_in = __builtin_amdgcn_s_memrealtime();   // 100 MHz: with t_in / t_out gives the clock the chip held
#endif
  // Everything the prologue needs from the kernel arguments is requested in ONE batch of scalar loads here (the compiler
  // otherwise fetches the operand pointers in a third dependent round trip, after the tile coordinates are known): the
  // empty asm makes the values live at this point.
  asm volatile("" ::"s"(p.A), "s"(p.B), "s"(p.lda), "s"(p.ldb), "s"(p.K), "s"(p.kchunk), "s"(p.tiles_n), "s"(p.nblocks),
               "s"(p.magic_tiles_n), "s"(p.limA), "s"(p.limB));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  // lane -> (index within the tile edge, k sub-index of the MFMA): 32x32x2 has 32 x 2, 16x16x4 has 16 x 4
  const int l31 = (MF == 32) ? (lane & 31) : (lane & 15), lhi = (MF == 32) ? (lane >> 5) : (lane >> 4);

  int logical, split;
  if (EPI == EPI_SLAB) {
    // split-K launch: the (slice, tile) units, slice-major, are dealt to the 8 XCDs in contiguous runs (work-groups go
    // round-robin over the XCDs, xcd_remap undoes that), so the tiles of one K-slice sit on one XCD - two when a run ends
    // inside the slice - and the slice's two operand strips (a few MB) are fetched from HBM once and re-read from that
    // XCD's L2 by the other tiles.  Spread over XCDs they were fetched 4.6x (measured: FETCH_SIZE).  Any slice count fills
    // all XCDs evenly (round 1 dealt whole slices, s -> XCD s mod 8, and needed a multiple of 8).
    const int total = p.nblocks * p.nsplits;
    if (bid >= total) return;   // a batched launch pads every problem's range to a multiple of 8
    const int u = xcd_remap(bid, total);
    split = div_magic(u, p.magic_nblocks);
    logical = u - split * p.nblocks;
  } else {
    logical = logical_in >= 0 ? logical_in : xcd_remap(bid, p.nblocks);
    split = 0;
  }
  const int tile_m = div_magic(logical, p.magic_tiles_n), tile_n = logical - tile_m * p.tiles_n;
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

  const brsrc resA = tile_resource<LOADA, BM, BK>(p.A, p.lda, m0, kb), resB = tile_resource<LOADB, BN, BK>(p.B, p.ldb, n0, kb);
  uint32_t voA[NVA], voB[NVB];
  tile_offsets<LOADA, BM, BK>(p.lda, voA, tid);
  tile_offsets<LOADB, BN, BK>(p.ldb, voB, tid);
  auto ld = [&](float4 (&xa)[NVA], float4 (&xb)[NVB], int i) {
    const int k0 = kb + min(i, nt - 1) * BK;   // past the end: re-read the last K-step (never consumed)
    load_tile<LOADA, BM, BK, COHA>(resA, voA, p.lda, k0 - kb, xa);
    load_tile<LOADB, BN, BK>(resB, voB, p.ldb, k0 - kb, xb);
  };
  auto st = [&](const float4 (&xa)[NVA], const float4 (&xb)[NVB], int stage) {
    float* An = smem + stage * Cfg::STAGE;
    if constexpr (KM) {
      store_tile_km<XFA, BM, LDK, BK>(An, xa, slopeA, tid);
      store_tile_km<XFB, BN, LDK, BK>(An + BOFF, xb, slopeB, tid);
    } else {
      store_tile<LOADA, XFA, BM, LDA, BK>(An, xa, slopeA, tid);
      store_tile<LOADB, XFB, BN, LDB, BK>(An + BOFF, xb, slopeB, tid);
    }
  };
  // A wave whose whole 32-granular tile range lies outside the matrix (the half-empty last tile row /
  // column: 352 = 5.5 x 64) issues no LDS reads and no MFMAs; it still loads, stores and meets the
  // barriers.  Its SIMD's matrix pipe goes to the other work-groups resident on the CU.
  const bool wave_active = (m0 + wm * (BM / Cfg::WM) < p.limA) && (n0 + wn * (BN / Cfg::WN) < p.limB);
  // EPI_TANH_REV: the reverse update's randoms (z_i and the keep bits of step i-1) depend on (row, column, step) only, so
  // they are drawn BEFORE the main loop, while the prologue's global loads are in flight, and wait in registers; drawn
  // in the epilogue they sit on the critical path of a launch whose work-groups all finish together (5429 rows: one
  // round).  One Philox call serves a column QUAD (four normals, four keep bits), as in k_reverse_update: the four
  // lanes of a quad each draw for a quarter of the tile's rows, then a 4x4 transpose across the quad (two DPP butterfly
  // stages) hands every lane the normals of its own column.
  constexpr int RQ = (EPI == EPI_TANH_REV) ? NR / 4 : 1;
  float rev_z[TM][TN][RQ][4];     // [..][h][kk]: normal (x nd) of row kk*RQ + h of the MFMA tile, this lane's column
  uint32_t rev_kb[TM][TN][RQ];    // bit kk: keep bit of that row
  auto rev_draw = [&]() {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * (BN / Cfg::WN) + b * MF + l31;
        const int rbase = m0 + wm * (BM / Cfg::WM) + a * MF + 4 * lhi;
        const int kq = col & 3;
        const bool k0 = kq & 1, k1 = kq & 2;
#pragma unroll
        for (int h = 0; h < RQ; ++h) {
          float n[4] = {0.f, 0.f, 0.f, 0.f};
          uint32_t bits = 0u;
#ifndef SDRM_DIAG_REV_NODRAW   // diagnostic builds only (timing the fused epilogue without its Philox work: results are wrong)
          if (p.rev_step > 1) {
#else
          if (p.rev_step > 1000000) {
#endif
            const int slot = p.rev_s0 + rbase + (MF == 32 ? 8 * kq + h : kq);   // rowoff(kq * RQ + h)
            const int rkey = (p.rev_rowid != nullptr && slot < p.rev_n) ? p.rev_rowid[slot] : slot;   // (multi-resolution: slots are sorted by start step)
            const U4 w = philox4x32_10((uint32_t)(p.rev_row0 + rkey), (uint32_t)(col >> 2),
                                       PURPOSE_SAMPLE_STEP | ((uint32_t)p.rev_step << 8), p.rev_call_id, p.rev_seed_lo, p.rev_seed_hi);
            box_muller(w.x, w.y, n[0], n[1]);
            box_muller(w.z, w.w, n[2], n[3]);
            bits = (w.x & 1u) | ((w.y & 1u) << 1) | ((w.z & 1u) << 2) | ((w.w & 1u) << 3);
          }
          float g;
          g = quad_xor<0xB1>(k0 ? n[0] : n[1]); if (k0) n[0] = g; else n[1] = g;
          g = quad_xor<0xB1>(k0 ? n[2] : n[3]); if (k0) n[2] = g; else n[3] = g;
          g = quad_xor<0x4E>(k1 ? n[0] : n[2]); if (k1) n[0] = g; else n[2] = g;
          g = quad_xor<0x4E>(k1 ? n[1] : n[3]); if (k1) n[1] = g; else n[3] = g;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) rev_z[a][b][h][kk] = n[kk] * p.rev_nd;   // now: lane kk's draw for MY column
          rev_kb[a][b][h] = ((quad_bcast<0>(bits) >> kq) & 1u) | (((quad_bcast<1>(bits) >> kq) & 1u) << 1) |
                            (((quad_bcast<2>(bits) >> kq) & 1u) << 2) | (((quad_bcast<3>(bits) >> kq) & 1u) << 3);
        }
      }
  };
  const int aoffk = (wm * (BM / Cfg::WM) + l31) * LDK + ((MF == 32) ? (BK / 2) : 4) * lhi;
  const int boffk = (wn * (BN / Cfg::WN) + l31) * LDK + ((MF == 32) ? (BK / 2) : 4) * lhi;

  if constexpr (KM) {
    // NT main loop.  A wave issues in order and its MFMAs form one dependent chain per accumulator tile, so the
    // only slots in which its other instructions cost nothing are the 64-cycle shadows right after each MFMA
    // issue (measured on a lone work-group: every LDS read / LDS store / global load / barrier placed outside
    // those shadows adds its full issue time to the K-step: 715 cycles per 8 MFMAs instead of 512+).  Hence one
    // small piece of the pipeline goes behind each MFMA group, order pinned with sched_barrier:
    //   fragments of K-step i+1 are read (b128) into the other register set,
    //   the operands of step i+2 go from their prefetch registers into the LDS stage step i vacated,
    //   the global loads of step i+4 are issued.
    // Two LDS stages, one barrier per K-step.
    constexpr int NS = 4 * NQ;   // MFMA groups (slots) per K-step
    static_assert(NS >= 2 * NQ + 4, "not enough MFMA shadows for the pipeline pieces");
    float4 fa0[TM][NQ], fb0[TN][NQ], fa1[TM][NQ], fb1[TN][NQ];
    auto rd_all = [&](float4 (&fa)[TM][NQ], float4 (&fb)[TN][NQ], int stage) {
      const float* As = smem + stage * Cfg::STAGE;
      const float* Bs = As + BOFF;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[a][q] = *reinterpret_cast<const float4*>(As + aoffk + MF * a * LDK + QSTEP * q);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[b][q] = *reinterpret_cast<const float4*>(Bs + boffk + MF * b * LDK + QSTEP * q);
      }
    };
    // one K-step: MFMAs out of (ca, cb); pieces: read (na, nb) from stage rs, store (xa, xb) into stage ss, load step li
    auto kstep = [&](auto active_tag, const float4 (&ca)[TM][NQ], const float4 (&cb)[TN][NQ], float4 (&na)[TM][NQ],
                     float4 (&nb)[TN][NQ], int rs, float4 (&xa)[NVA], float4 (&xb)[NVB], int ss, int li) {
      constexpr bool ACTIVE = decltype(active_tag)::value;
      const float* Ar = smem + rs * Cfg::STAGE;
      const float* Br = Ar + BOFF;
      float* Aw = smem + ss * Cfg::STAGE;
      const int k0 = kb + min(li, nt - 1) * BK;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const int q = sl >> 2, j = sl & 3;
        if constexpr (ACTIVE) {
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
              const float av = j == 0 ? ca[a][q].x : j == 1 ? ca[a][q].y : j == 2 ? ca[a][q].z : ca[a][q].w;
              const float bv = j == 0 ? cb[b][q].x : j == 1 ? cb[b][q].y : j == 2 ? cb[b][q].z : cb[b][q].w;
              if constexpr (MF == 32) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
              else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a][b], 0, 0, 0);
            }
        }
        // the piece that rides in this group's shadow
        if (sl < 2 * NQ) {
          if constexpr (ACTIVE) {
            const int rq = sl >> 1;
            if ((sl & 1) == 0) {
#pragma unroll
              for (int a = 0; a < TM; ++a) na[a][rq] = *reinterpret_cast<const float4*>(Ar + aoffk + MF * a * LDK + QSTEP * rq);
            } else {
#pragma unroll
              for (int b = 0; b < TN; ++b) nb[b][rq] = *reinterpret_cast<const float4*>(Br + boffk + MF * b * LDK + QSTEP * rq);
            }
          }
        } else if (sl == 2 * NQ) {
          // the operand transform (PReLU of stored pre-activations) gets a shadow of its own: with the stores it
          // overran one (12 VALU ops + two 13-cycle ds_write_b128 > 64 cycles)
          if constexpr (XFA == XF_PRELU) prelu_regs<NVA>(xa, slopeA);
          else DIAG_ST((store_tile_km<XF_NONE, BM, LDK, BK>(Aw, xa, 0.f, tid)));
        } else if (sl == 2 * NQ + 1) {
          if constexpr (XFA == XF_PRELU) DIAG_ST((store_tile_km<XF_NONE, BM, LDK, BK>(Aw, xa, 0.f, tid)));
          DIAG_ST((store_tile_km<XFB, BN, LDK, BK>(Aw + BOFF, xb, slopeB, tid)));
        } else if (sl == 2 * NQ + 2) {
          DIAG_LD((load_tile<LOADA, BM, BK, COHA>(resA, voA, p.lda, k0 - kb, xa)));
        } else if (sl == 2 * NQ + 3) {
          DIAG_LD((load_tile<LOADB, BN, BK>(resB, voB, p.ldb, k0 - kb, xb)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      DIAG_BAR();
    };
    if (nt > 0) {
      if constexpr (Cfg::PF == 2) {
        ld(ra0, rb0, 0);
        ld(ra1, rb1, 1);
#if defined(SDRM_STAMPS) && SDRM_STAMPS == 2
        t_ld = __builtin_amdgcn_s_memtime();     // setup done, first loads issued
        __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (EPI == EPI_TANH_REV) rev_draw();   // VALU work under the first loads' round trip
        st(ra0, rb0, 0);
#if defined(SDRM_STAMPS) && SDRM_STAMPS == 2
        __builtin_amdgcn_sched_barrier(0);
        t_land = __builtin_amdgcn_s_memtime();   // the first K-step's operands have landed and sit in LDS
#endif
        ld(ra0, rb0, 2);
        st(ra1, rb1, 1);
        ld(ra1, rb1, 3);
      } else {
        ld(ra0, rb0, 0);
        if constexpr (EPI == EPI_TANH_REV) rev_draw();
        st(ra0, rb0, 0);
        ld(ra0, rb0, 1);
        st(ra0, rb0, 1);
        ld(ra0, rb0, 2);
      }
      __syncthreads();
#ifdef SDRM_STAMPS
      if (true) t_pro = __builtin_amdgcn_s_memtime();
#endif
      if (wave_active) rd_all(fa0, fb0, 0);
      __syncthreads();   // stage 0 is overwritten from the first K-step on: every wave must hold its step-0 fragments
      // step i multiplies out of fragment set i%2, reads step i+1's fragments from stage (i+1)%2, stores step i+2
      // (prefetch set i%2) into stage i%2 and requests step i+4 into the same set.  (Four prefetch sets, i.e. loads
      // requested four steps ahead, were measured: 3 % off a lone work-group's loop, 2 % slower on full launches.)
      // PF == 1: the single set holds step i+2 when step i stores it and is refilled with step i+3.
      auto run = [&](auto tag) {
        int i = 0;
        for (; i + 1 < nt; i += 2) {
          if constexpr (Cfg::PF == 2) {
            kstep(tag, fa0, fb0, fa1, fb1, 1, ra0, rb0, 0, i + 4);
            kstep(tag, fa1, fb1, fa0, fb0, 0, ra1, rb1, 1, i + 5);
          } else {
            kstep(tag, fa0, fb0, fa1, fb1, 1, ra0, rb0, 0, i + 3);
            kstep(tag, fa1, fb1, fa0, fb0, 0, ra0, rb0, 1, i + 4);
          }
        }
        if (i < nt) kstep(tag, fa0, fb0, fa1, fb1, 1, ra0, rb0, 0, i + 4);   // odd count (its pieces are harmless)
      };
      if (wave_active) run(std::true_type{});
      else run(std::false_type{});   // a wave whose tile range lies outside the matrix only feeds the pipeline
    }
  } else {
    // k-major operands (wgrad: both operands are row-major over the reduction index; also the 16-wide MFMA
    // tiles of small launches): the same pipeline, fragments fetched with ds_read_b32 - the pair of reads that
    // feeds MFMA group g of the NEXT K-step rides in the shadow of group g of this one.
    constexpr int G = BK / KG;   // MFMA groups per K-step
    static_assert(G >= 8, "pipeline pieces need 8 MFMA shadows");
    float ka0[TM][G], kb0[TN][G], ka1[TM][G], kb1[TN][G];
    auto col_sum = [&](int stage) {   // bias gradient: column sums of the A operand (dY^T) of the K-step in `stage`
      const float* As = smem + stage * Cfg::STAGE;
#pragma unroll
      for (int k = 0; k < BK; ++k) dbsum += As[k * LDA + tid];
    };
    auto kstep = [&](auto active_tag, const float (&ca)[TM][G], const float (&cb)[TN][G], float (&na)[TM][G],
                     float (&nb)[TN][G], int rs, float4 (&xa)[NVA], float4 (&xb)[NVB], int ss, int li, bool next_valid) {
      constexpr bool ACTIVE = decltype(active_tag)::value;
      const float* Ar = smem + rs * Cfg::STAGE;
      const float* Br = Ar + BOFF;
      float* Aw = smem + ss * Cfg::STAGE;
      const int k0 = kb + min(li, nt - 1) * BK;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if constexpr (ACTIVE) {
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
              if constexpr (MF == 32) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[a][g], cb[b][g], acc[a][b], 0, 0, 0);
              else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[a][g], cb[b][g], acc[a][b], 0, 0, 0);
            }
#pragma unroll
          for (int a = 0; a < TM; ++a) na[a][g] = Ar[aoff + KG * g * LDA + MF * a];
#pragma unroll
          for (int b = 0; b < TN; ++b) nb[b][g] = Br[boff + KG * g * LDB + MF * b];
        }
        if (EPI == EPI_SLAB && do_dbias && next_valid) {   // bias gradient: this thread's column of the NEXT K-step's
#pragma unroll                                            // dY^T tile, BK/G rows per shadow (wave-uniform branch)
          for (int k = g * (BK / G); k < (g + 1) * (BK / G); ++k) dbsum += Ar[k * LDA + tid];
        }
        if (g == G / 2) store_tile<LOADA, XFA, BM, LDA, BK>(Aw, xa, slopeA, tid);
        else if (g == G / 2 + 1) store_tile<LOADB, XFB, BN, LDB, BK>(Aw + BOFF, xb, slopeB, tid);
        else if (g == G / 2 + 2) load_tile<LOADA, BM, BK, COHA>(resA, voA, p.lda, k0 - kb, xa);
        else if (g == G / 2 + 3) load_tile<LOADB, BN, BK>(resB, voB, p.ldb, k0 - kb, xb);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    if (nt > 0) {
      ld(ra0, rb0, 0);
      ld(ra1, rb1, 1);
      st(ra0, rb0, 0);
      ld(ra0, rb0, 2);
      st(ra1, rb1, 1);
      ld(ra1, rb1, 3);
      __syncthreads();
#ifdef SDRM_STAMPS
      if (true) t_pro = __builtin_amdgcn_s_memtime();
#endif
      if (wave_active) {
        const float* As = smem;
        const float* Bs = As + BOFF;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int a = 0; a < TM; ++a) ka0[a][g] = As[aoff + KG * g * LDA + MF * a];
#pragma unroll
          for (int b = 0; b < TN; ++b) kb0[b][g] = Bs[boff + KG * g * LDB + MF * b];
        }
      }
      if (do_dbias) col_sum(0);
      __syncthreads();   // stage 0 is overwritten from the first K-step on
      auto run = [&](auto tag) {
        int i = 0;
        for (; i + 1 < nt; i += 2) {
          kstep(tag, ka0, kb0, ka1, kb1, 1, ra0, rb0, 0, i + 4, true);
          kstep(tag, ka1, kb1, ka0, kb0, 0, ra1, rb1, 1, i + 5, i + 2 < nt);
        }
        if (i < nt) kstep(tag, ka0, kb0, ka1, kb1, 1, ra0, rb0, 0, i + 4, false);
      };
      if (wave_active) run(std::true_type{});
      else run(std::false_type{});
    }
  }
#ifdef SDRM_STAMPS
  t_loop = __builtin_amdgcn_s_memtime();
#endif

  // ------------------------------------------------------------------ epilogue
  // accumulator register r of tile (a,b), 32-wide: row = m0 + wm*(BM/WM) + a*32 + (r&3) + 8*(r>>2) + 4*lhi,
  //                                                 col = n0 + wn*(BN/WN) + b*32 + l31
  //                                        16-wide: row = ... + a*16 + 4*lhi + r ; col = ... + b*16 + l31
  auto rowoff = [](int r) { return (MF == 32) ? (r & 3) + 8 * (r >> 2) : r; };
  float slope_sum = 0.f;
  const float slopeE = (EPI == EPI_DPRELU) ? *p.slopeE : 0.f;
  const float slopeP = (EPI == EPI_BIAS_PRELU) ? *p.slopeE : 0.f;
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
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH || EPI == EPI_BIAS_TANH_G || EPI == EPI_TANH_REV || EPI == EPI_BIAS_G ||
          EPI == EPI_BIAS_PRELU)
        bias = p.bias[col];
      if (EPI == EPI_BIAS || EPI == EPI_PLAIN) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          Cp[(size_t)row * p.ldc + col] = acc[a][b][r] + bias;
        }
      } else if (EPI == EPI_BIAS_PRELU) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          Cp[(size_t)row * p.ldc + col] = prelu_f(acc[a][b][r] + bias, slopeP);
        }
      } else if (EPI == EPI_BIAS_TANH) {
        float y[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) y[r] = tanh_fast(acc[a][b][r] + bias);
#pragma unroll
        for (int r = 0; r < NR; ++r) Cp[(size_t)(rbase + rowoff(r)) * p.ldc + col] = y[r];
      } else if (EPI == EPI_TANH_REV) {
        // x <- (x - eps_hat*c1)/sqrt(alpha_i) + sqrt(beta_i)*z ; U_next = keep ? 2x : 0   (train_SDRM.py:20-25 + :100),
        // the arithmetic of k_reverse_update, with the randoms rev_draw() left in registers before the main loop.
        constexpr int Q = NR / 4;
        const bool noise = p.rev_step > 1;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          const float z = rev_z[a][b][r % Q][r / Q];
          const bool kp = (rev_kb[a][b][r % Q] >> (r / Q)) & 1u;
          if (row + p.rev_s0 < p.rev_n) {
            const size_t xi = (size_t)(p.rev_s0 + row) * p.rev_ldx + col;
            const float e = tanh_fast(acc[a][b][r] + bias);
            const float xn = (col < p.rev_L) ? (p.revX[xi] - e * p.rev_c1) / p.rev_sqrt_alpha + p.rev_sqrt_beta * z : 0.f;
            p.revX[xi] = xn;
            if (noise) p.revU[xi] = kp ? 2.f * xn : 0.f;
          }
        }
      } else if (EPI == EPI_BIAS_TANH_G) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          if (row < p.rows_valid && col < p.cols_valid) Cp[(size_t)row * p.ldc + col] = tanh_fast(acc[a][b][r] + bias);
        }
      } else if (EPI == EPI_BIAS_ROWTAB) {
        int tt[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          const int pass = (row >= p.trow_B) + (row >= 2 * p.trow_B);
          tt[r] = (row < 3 * p.trow_B) ? p.trow[row - pass * p.trow_B] : 0;
        }
        float bt[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) bt[r] = p.bias[(size_t)tt[r] * p.ldtab + col];
#pragma unroll
        for (int r = 0; r < NR; ++r) Cp[(size_t)(rbase + rowoff(r)) * p.ldc + col] = acc[a][b][r] + bt[r];
      } else if (EPI == EPI_BIAS_G) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rbase + rowoff(r);
          if (row < p.rows_valid && col < p.cols_valid) Cp[(size_t)row * p.ldc + col] = acc[a][b][r] + bias;
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
  if (p.stamps && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* o = p.stamps + 8 * (size_t)bid;
    o[0] = t_in; o[1] = t_pro; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
    o[4] = r_in; o[5] = __builtin_amdgcn_s_memrealtime();
    // where the work-group ran: HW_REG_HW_ID (id 4: cu_id bits 11:8, sh_id 12, se_id 15:13) and HW_REG_XCC_ID (id 20)
#if SDRM_STAMPS == 2
    o[6] = t_ld; o[7] = t_land;
#else
    o[6] = (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    o[7] = (unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11));
#endif
  }
#endif
  if (EPI == EPI_DPRELU) {
    // block-wide sum of the slope-gradient partial -> one float per block (deterministic order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) slope_sum += __shfl_down(slope_sum, off, 64);
    __syncthreads();
    if (lane == 0) smem[wave] = slope_sum;
    __syncthreads();
    if (tid == 0) p.slope_partial[bid] = smem[0] + smem[1] + smem[2] + smem[3];
  }
  if (EPI == EPI_SLAB) {
    if (do_dbias && (m0 + tid) < p.limA) p.dbias[(size_t)split * p.dbias_stride + m0 + tid] = dbsum;
  }
}

template <class Cfg, int LOADA, int LOADB, int XFA, int XFB, int EPI>
__global__ __launch_bounds__(NTHREADS, Cfg::MINW) void gemm_kernel(const GemmArgs p) {
  gemm_body<Cfg, LOADA, LOADB, XFA, XFB, EPI>(p, (int)blockIdx.x);
}

// Several independent GEMMs of one kind in ONE launch (the weight gradients of all layers: every input is ready once
// the dgrad chain is done): one ramp and one drain instead of one per layer, and the layers' work-groups fill each
// other's tails.  Each problem owns the grid range [start[k], start[k+1]); the ranges are multiples of 8 work-groups,
// so a work-group's XCD (blockIdx.x & 7) is also its XCD inside the problem.
constexpr int GEMM_BATCH_MAX = 8;
struct GemmBatch {
  GemmArgs p[GEMM_BATCH_MAX];
  int start[GEMM_BATCH_MAX + 1];
  int n;
};

template <class Cfg, int LOADA, int LOADB, int XFA, int XFB, int EPI>
__global__ __launch_bounds__(NTHREADS, Cfg::MINW) void gemm_batch_kernel(const GemmBatch b) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < GEMM_BATCH_MAX; ++j)
    if (j < b.n && (int)blockIdx.x >= b.start[j]) k = j;
  gemm_body<Cfg, LOADA, LOADB, XFA, XFB, EPI>(b.p[k], (int)blockIdx.x - b.start[k]);
}

}  // namespace sdrm
