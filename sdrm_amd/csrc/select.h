// Equal-sparsity threshold of the sampled matrix on the device (main.py:177-180):
//     threshold = np.quantile(M.flatten(), SPARSITY);  out = (M >= threshold)
// np.quantile(method="linear") needs two exact order statistics of the n = users x items floats (up to 82 M
// for ADM) and a linear interpolation between them.  The order statistics come from a three-pass radix select
// on the order-preserving 32-bit key of a float (11 + 11 + 10 bits): each pass is one HBM-bound sweep that
// histograms the next digit of the elements still matching the selected prefix (LDS histogram per block, one
// integer atomic per non-empty bin per block - integer atomics make the counts order-independent), and a
// one-block kernel walks the 2048 bins to the one holding the wanted rank.  Both ranks (floor and floor+1 of
// the virtual index) are tracked at once; they share their histograms while their prefixes agree.
// Bytes per element: 3 x 4 (select) + 4 + 1 (binarise) = 17.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

constexpr int SEL_BINS = 2048;
constexpr int SEL_PASSES = 3;
__device__ __host__ inline int sel_shift(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__device__ __host__ inline int sel_bits(int pass) { return pass == 2 ? 10 : 11; }

// monotone map float -> uint32 (ascending): negative floats reverse, positive floats get the sign bit set
__device__ __forceinline__ uint32_t float_key(float x) {
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct SelectState {
  uint32_t prefix[2];     // selected high bits so far (low bits zero)
  int64_t rank[2];        // rank still to find among the elements matching the prefix (0-based)
  uint32_t hist[SEL_PASSES][2][SEL_BINS];
  float value[2];         // the two order statistics (after the last pass)
  float threshold;        // lerp(value[0], value[1], gamma)
};

__global__ __launch_bounds__(256) void k_select_init(SelectState* s, int64_t r0, int64_t r1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t* h = &s->hist[0][0][0];
  for (int j = i; j < SEL_PASSES * 2 * SEL_BINS; j += gridDim.x * 256) h[j] = 0u;
  if (i == 0) {
    s->prefix[0] = s->prefix[1] = 0u;
    s->rank[0] = r0; s->rank[1] = r1;
  }
}

// One sweep: digit histogram of the elements whose higher bits equal the selected prefix (both ranks).
// Pass 0 sees every element and its digit is sign + exponent + 2 mantissa bits - a few dozen hot bins - so its LDS
// histogram is replicated REP times (lane % REP picks the copy) to cut same-address atomic serialisation; later
// passes touch only the elements inside the selected bin (a few %, then a few ppm) and are pure streaming reads.
template <int REP>
__global__ __launch_bounds__(256) void k_select_hist(const float* __restrict__ x, int64_t n, SelectState* s, int pass) {
  __shared__ uint32_t h0[REP][SEL_BINS];
  __shared__ uint32_t h1[SEL_BINS];
  for (int j = threadIdx.x; j < SEL_BINS; j += 256) {
#pragma unroll
    for (int r = 0; r < REP; ++r) h0[r][j] = 0u;
    h1[j] = 0u;
  }
  __syncthreads();
  const int shift = sel_shift(pass), bits = sel_bits(pass);
  const uint32_t himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + bits));
  const uint32_t p0 = s->prefix[0], p1 = s->prefix[1];
  const bool same = (p0 == p1);
  const uint32_t dmask = (1u << bits) - 1u;
  uint32_t* mine = h0[threadIdx.x % REP];
  const int64_t nv = n >> 2;   // float4 part (x is 16-byte aligned: a torch allocation)
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const float4 v = x4[i];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t k = float_key(e[j]);
      const uint32_t hi = k & himask, d = (k >> shift) & dmask;
      if (hi == p0) atomicAdd(&mine[d], 1u);
      if (!same && hi == p1) atomicAdd(&h1[d], 1u);
    }
  }
  if (blockIdx.x == 0) {   // tail
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += 256) {
      const uint32_t k = float_key(x[i]);
      const uint32_t hi = k & himask, d = (k >> shift) & dmask;
      if (hi == p0) atomicAdd(&mine[d], 1u);
      if (!same && hi == p1) atomicAdd(&h1[d], 1u);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < SEL_BINS; j += 256) {
    uint32_t c = 0u;
#pragma unroll
    for (int r = 0; r < REP; ++r) c += h0[r][j];
    if (c) atomicAdd(&s->hist[pass][0][j], c);
    if (!same && h1[j]) atomicAdd(&s->hist[pass][1][j], h1[j]);
  }
}

// One block of 256 threads: exclusive prefix sums of the 2048 bins (8 per thread + a scan of the 256 partials in
// LDS), then the thread whose bin range holds a rank extends that rank's prefix.  After the last pass the prefixes
// are the keys of the two order statistics; the threshold is numpy's _lerp in float32 (no fma contraction).
__global__ __launch_bounds__(256) void k_select_pick(SelectState* s, int pass, float gamma) {
  __shared__ uint32_t part[2][256];
  __shared__ uint32_t newp[2];
  __shared__ int64_t newr[2];
  const int tid = threadIdx.x;
  const bool same = (s->prefix[0] == s->prefix[1]);
  const int nb = 1 << sel_bits(pass);
  constexpr int PER = SEL_BINS / 256;
  uint32_t c[2][PER];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const uint32_t* h = s->hist[pass][(w == 1 && !same) ? 1 : 0];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int b = tid * PER + j;
      c[w][j] = (b < nb) ? h[b] : 0u;
      sum += c[w][j];
    }
    part[w][tid] = sum;
  }
  const int64_t r0 = s->rank[0], r1 = s->rank[1];
  const uint32_t p0 = s->prefix[0], p1 = s->prefix[1];
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    // exclusive prefix of this thread's chunk (counts fit 32 bits per bin; the running sum needs 64)
    int64_t before = 0;
    for (int i = 0; i < tid; ++i) before += part[w][i];
    const int64_t r = (w == 0) ? r0 : r1;
    int64_t acc = before;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int b = tid * PER + j;
      const int64_t cnt = c[w][j];
      // the last real bin also catches a rank beyond the total (cannot happen for a valid rank)
      if (b < nb && ((r >= acc && r < acc + cnt) || (b == nb - 1 && r >= acc + cnt))) {
        newp[w] = ((w == 0) ? p0 : p1) | ((uint32_t)b << sel_shift(pass));
        newr[w] = (r >= acc + cnt) ? 0 : r - acc;
      }
      acc += cnt;
    }
  }
  __syncthreads();
  if (tid == 0) {
    s->prefix[0] = newp[0]; s->prefix[1] = newp[1];
    s->rank[0] = newr[0]; s->rank[1] = newr[1];
    if (pass == SEL_PASSES - 1) {
      const float a = key_float(newp[0]), bb = key_float(newp[1]);
      s->value[0] = a; s->value[1] = bb;
      const float diff = __fsub_rn(bb, a);
      float t = __fadd_rn(a, __fmul_rn(diff, gamma));
      if (gamma >= 0.5f) t = __fsub_rn(bb, __fmul_rn(diff, __fsub_rn(1.0f, gamma)));
      s->threshold = t;
    }
  }
}

// out = (x >= threshold), one byte per element
__global__ __launch_bounds__(256) void k_binarize_ge(const float* __restrict__ x, int64_t n, const float* __restrict__ thr,
                                                     uint8_t* __restrict__ out) {
  const float t = *thr;
  const int64_t nv = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  uint32_t* o4 = reinterpret_cast<uint32_t*>(out);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const float4 v = x4[i];
    o4[i] = (v.x >= t ? 1u : 0u) | (v.y >= t ? 0x100u : 0u) | (v.z >= t ? 0x10000u : 0u) | (v.w >= t ? 0x1000000u : 0u);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += 256) out[i] = x[i] >= t ? 1 : 0;
}

}  // namespace sdrm
