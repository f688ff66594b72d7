// Counter-based RNG for the PHILOX mode: Philox4x32-10 keyed by the 64-bit
// seed, counter = (global row, column quad, purpose | sub-step << 8, step).  Keying by the GLOBAL
// row makes a G-GPU run draw exactly the randoms of the 1-GPU run.  The same function is restated
// in numpy in oracle/philox_ref.py so PHILOX-mode runs are parity-testable too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

enum : uint32_t {
  PURPOSE_TRAIN_ELEM = 1,  // four normals + 3x4 dropout bits of one train step (column quad)
  PURPOSE_TRAIN_T = 2,     // timestep of a row
  PURPOSE_SAMPLE_XT = 3,   // start noise of a column quad
  PURPOSE_SAMPLE_STEP = 4, // (four z, four dropout bits) of reverse step i (i in bits 8..), column quad
  PURPOSE_SAMPLE_TJ = 5,   // multi-resolution start step of a row
  PURPOSE_FORWARD = 6      // dropout bits of a plain forward call (column pair)
};

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// Two standard normals from two 32-bit words: Box-Muller on 24-bit uniforms with the hardware
// transcendentals (v_log_f32 = log2, v_sqrt_f32, v_sin/v_cos_f32 take their argument in turns).
// Their ~1e-6 absolute error is irrelevant for noise and keeps the generator off the critical path
// of the fused sampling epilogue; oracle/philox_ref.py mirrors the formula in numpy.
__device__ inline void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u = (float)((a >> 8) + 1u) * 5.9604644775390625e-8f;  // (0,1]
  const float v = (float)(b >> 8) * 5.9604644775390625e-8f;         // [0,1) turns
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));  // sqrt(-2 ln u)
  n0 = r * __builtin_amdgcn_cosf(v);
  n1 = r * __builtin_amdgcn_sinf(v);
}

// uniform integer in [0, n) from one word (multiply-high)
__host__ __device__ inline uint32_t bounded(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }

}  // namespace sdrm
