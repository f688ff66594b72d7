// fp32 MFMA GEMM for the SDRM eps-net on gfx950 (CDNA4).
//
// One kernel template covers the three contractions of a Linear layer
// (forward x*W^T, dgrad dY*W, wgrad dY^T*X) by choosing how each operand tile is
// brought into LDS.  Inside LDS both operands are always k-major
// (As[k][i], Bs[k][j]), which is the layout v_mfma_f32_32x32x2_f32 consumes with
// conflict-free ds_read_b32: lane l reads As[k0 + (l>>5)][i0 + (l&31)].
//
//   block tile 128x128, K-step 16, 256 threads = 4 waves as 2(M) x 2(N),
//   each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs),
//   LDS double-buffered (2 x 2 x 16 x 132 floats = 33 KB), one barrier per K-step,
//   global->register prefetch of the next K-step issued before the MFMAs of the current one.
//
// fp32-input MFMA is bit-for-bit a k-ordered fmaf chain, so the results sit well inside the 1e-4
// normwise parity bar.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int { LD_KCONTIG = 0, LD_MCONTIG = 1 };  // operand element (i,k) at src[i*ld+k]  /  src[k*ld+i]
enum : int { XF_NONE = 0, XF_PRELU = 1 };
enum : int {
  EPI_BIAS = 0,          // C = acc + bias[n]                         (hidden pre-activations)
  EPI_BIAS_TANH = 1,     // C = tanh(acc + bias[n])                   (eps-net output)
  EPI_TANH_REVERSE = 2,  // fused DDPM reverse update + next step's dropout (sampling)
  EPI_DPRELU = 3,        // C = acc * prelu'(aux) ; partial sum of acc*min(aux,0) (slope gradient)
  EPI_SLAB = 4,          // C = acc into split-K slab blockIdx.z ; optional column sums (bias grad)
  EPI_PLAIN = 5          // C = acc (debug)
};

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4, NTHREADS = 256;

struct GemmArgs {
  const float* A; int lda; int limA;   // limA: valid extent of A's output-side index (multiple of 32)
  const float* B; int ldb; int limB;
  float* C; int ldc;
  int K;            // full reduction length (multiple of BK)
  int kchunk;       // reduction range handled by one blockIdx.z (multiple of BK)
  int tiles_n;      // number of 128-wide tiles along N (grid.x = tiles_m * tiles_n, XCD-swizzled)
  int nblocks;      // tiles_m * tiles_n
  const float* bias;
  const float* slopeA;  // PReLU slope applied to A elements on load (XF_PRELU)
  const float* slopeB;
  // EPI_DPRELU
  const float* aux; int ldaux; const float* slopeE; float* slope_partial;
  // EPI_SLAB
  size_t slab_stride; float* dbias; int dbias_stride;
  // EPI_BIAS_TANH: output may be an unpadded caller buffer
  int rows_valid, cols_valid;
  // EPI_TANH_REVERSE
  float* X; const float* Z; const uint8_t* keep_next; float* Unext; int ldu;
  const int64_t* Tj; int step_i; float c1, sqrt_alpha, sqrt_beta, nd;
  int rng_mode; uint32_t seed_lo, seed_hi, call_id; int64_t row0; int Lreal;
};

__device__ __forceinline__ float prelu_f(float v, float a) { return v > 0.f ? v : a * v; }

template <int LOAD, int XF>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int ld, int i0, int lim, int k0,
                                          float slope, float4 (&r)[2], int tid) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int f = tid + s * NTHREADS;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LOAD == LD_KCONTIG) {
      const int i = f >> 2, kq = f & 3;
      if (i0 + i < lim) v = *reinterpret_cast<const float4*>(src + (size_t)(i0 + i) * ld + k0 + 4 * kq);
    } else {
      const int k = f >> 5, iq = f & 31;
      if (i0 + 4 * iq < lim) v = *reinterpret_cast<const float4*>(src + (size_t)(k0 + k) * ld + i0 + 4 * iq);
    }
    if (XF == XF_PRELU) {
      v.x = prelu_f(v.x, slope); v.y = prelu_f(v.y, slope); v.z = prelu_f(v.z, slope); v.w = prelu_f(v.w, slope);
    }
    r[s] = v;
  }
}

template <int LOAD>
__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float4 (&r)[2], int tid) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int f = tid + s * NTHREADS;
    if (LOAD == LD_KCONTIG) {
      const int i = f >> 2, kq = f & 3;
      float* d = dst + (4 * kq) * LDT + i;
      d[0] = r[s].x; d[LDT] = r[s].y; d[2 * LDT] = r[s].z; d[3 * LDT] = r[s].w;
    } else {
      const int k = f >> 5, iq = f & 31;
      *reinterpret_cast<float4*>(dst + k * LDT + 4 * iq) = r[s];
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

}  // namespace sdrm

// ---------------------------------------------------------------------------------------------
#include "philox.h"

namespace sdrm {

template <int LOADA, int LOADB, int XFA, int XFB, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LDT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int tile_m = logical / p.tiles_n, tile_n = logical - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kb = blockIdx.z * p.kchunk;
  const int ke = min(kb + p.kchunk, p.K);

  const float slopeA = (XFA == XF_PRELU) ? *p.slopeA : 0.f;
  const float slopeB = (XFB == XF_PRELU) ? *p.slopeB : 0.f;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // wave-uniform validity of the four 32x32 MFMA tiles (feature dims are multiples of 32)
  bool vm[2], vn[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    vm[a] = (m0 + wm * 64 + a * 32) < p.limA;
    vn[a] = (n0 + wn * 64 + a * 32) < p.limB;
  }

  float4 ra[2], rb[2];
  float dbsum = 0.f;
  const bool do_dbias = (EPI == EPI_SLAB) && (p.dbias != nullptr) && (tile_n == 0) && (tid < BM);

  int stage = 0;
  if (kb < ke) {
    load_tile<LOADA, XFA>(p.A, p.lda, m0, p.limA, kb, slopeA, ra, tid);
    load_tile<LOADB, XFB>(p.B, p.ldb, n0, p.limB, kb, slopeB, rb, tid);
    store_tile<LOADA>(smem, ra, tid);
    store_tile<LOADB>(smem + BK * LDT, rb, tid);
  }
  __syncthreads();

  for (int kt = kb; kt < ke; kt += BK) {
    const float* As = smem + stage * (2 * BK * LDT);
    const float* Bs = As + BK * LDT;
    const bool more = (kt + BK) < ke;
    if (more) {
      load_tile<LOADA, XFA>(p.A, p.lda, m0, p.limA, kt + BK, slopeA, ra, tid);
      load_tile<LOADB, XFB>(p.B, p.ldb, n0, p.limB, kt + BK, slopeB, rb, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float* ap = As + (kk + lhi) * LDT + wm * 64 + l31;
      const float* bp = Bs + (kk + lhi) * LDT + wn * 64 + l31;
      const float a0 = ap[0], a1 = ap[32];
      const float b0 = bp[0], b1 = bp[32];
      if (vm[0] && vn[0]) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      if (vm[0] && vn[1]) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      if (vm[1] && vn[0]) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      if (vm[1] && vn[1]) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (do_dbias) {
#pragma unroll
      for (int k = 0; k < BK; ++k) dbsum += As[k * LDT + tid];
    }
    if (more) {
      float* An = smem + (stage ^ 1) * (2 * BK * LDT);
      store_tile<LOADA>(An, ra, tid);
      store_tile<LOADB>(An + BK * LDT, rb, tid);
    }
    __syncthreads();
    stage ^= 1;
  }

  // ------------------------------------------------------------------ epilogue
  // accumulator register r of tile (a,b): row = m0 + wm*64 + a*32 + (r&3) + 8*(r>>2) + 4*lhi,
  //                                        col = n0 + wn*64 + b*32 + l31
  float slope_sum = 0.f;
  const float slopeE = (EPI == EPI_DPRELU) ? *p.slopeE : 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (!(vm[a] && vn[b])) continue;
      const int col = n0 + wn * 64 + b * 32 + l31;
      const int rbase = m0 + wm * 64 + a * 32 + 4 * lhi;
      float bias = 0.f;
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH || EPI == EPI_TANH_REVERSE) bias = p.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const float v = acc[a][b][r];
        if (EPI == EPI_BIAS) {
          p.C[(size_t)row * p.ldc + col] = v + bias;
        } else if (EPI == EPI_PLAIN) {
          p.C[(size_t)row * p.ldc + col] = v;
        } else if (EPI == EPI_BIAS_TANH) {
          if (row < p.rows_valid && col < p.cols_valid) p.C[(size_t)row * p.ldc + col] = tanhf(v + bias);
        } else if (EPI == EPI_DPRELU) {
          const float pre = p.aux[(size_t)row * p.ldaux + col];
          const bool pos = pre > 0.f;
          p.C[(size_t)row * p.ldc + col] = pos ? v : slopeE * v;
          slope_sum += pos ? 0.f : v * pre;
        } else if (EPI == EPI_SLAB) {
          p.C[(size_t)blockIdx.z * p.slab_stride + (size_t)row * p.ldc + col] = v;
        } else if (EPI == EPI_TANH_REVERSE) {
          if (row < p.rows_valid && col < p.Lreal) {
            const float eps_hat = tanhf(v + bias);
            const size_t xi = (size_t)row * p.ldc + col;
            const float x_old = p.X[xi];
            const bool active = (p.Tj == nullptr) || (p.Tj[row] >= (int64_t)p.step_i);
            float z = 0.f;
            bool keep = false;
            if (p.rng_mode == 0) {
              if (p.Z != nullptr) z = p.Z[(size_t)row * p.Lreal + col];
              if (p.keep_next != nullptr) keep = p.keep_next[(size_t)row * p.Lreal + col] != 0;
            } else {
              const uint32_t grow = (uint32_t)(p.row0 + row);
              if (p.step_i > 1) {
                const U4 w = philox4x32_10(grow, (uint32_t)(col >> 1), PURPOSE_SAMPLE_STEP | ((uint32_t)p.step_i << 8),
                                           p.call_id, p.seed_lo, p.seed_hi);
                float n0f, n1f;
                box_muller(w.x, w.y, n0f, n1f);
                z = ((col & 1) ? n1f : n0f) * p.nd;
                const U4 w2 = philox4x32_10(grow, (uint32_t)(col >> 1),
                                            PURPOSE_SAMPLE_STEP | ((uint32_t)(p.step_i - 1) << 8), p.call_id,
                                            p.seed_lo, p.seed_hi);
                keep = ((w2.z >> ((col & 1) * 8)) & 1u) != 0;
              }
            }
            const float x_new = active ? (x_old - eps_hat * p.c1) / p.sqrt_alpha + p.sqrt_beta * z : x_old;
            p.X[xi] = x_new;
            if (p.step_i > 1) p.Unext[(size_t)row * p.ldu + col] = keep ? 2.f * x_new : 0.f;
          }
        }
      }
    }
  }

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
    if (do_dbias && (m0 + tid) < p.limA) p.dbias[(size_t)blockIdx.z * p.dbias_stride + m0 + tid] = dbsum;
  }
}

}  // namespace sdrm
