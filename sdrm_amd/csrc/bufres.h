// Raw buffer accesses for the one-wave-per-SIMD kernels (rowchain.h, dgrad_rows.h, wgrad2.h), gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

typedef float f32x4_b __attribute__((ext_vector_type(4)));

// Raw buffer accesses: base in a 4-SGPR resource, a per-lane 32-bit offset, a scalar offset advanced on the scalar unit -
// ONE instruction and no VALU (the global_load form costs a 64-bit VALU add per access once the base moves).  Loads beyond the
// resource's size return zero, stores beyond it are dropped.
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc make_brsrc(const void* p, uint32_t bytes) {   // p wave-uniform
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_b bload4(brsrc r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4_b, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float bload1(brsrc r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// ... with cache-policy bits: AUX = 16 is sc1 - the load is served by the XCD's L2, never by this CU's L1 (what a work-group reads
// of another work-group's stores of the SAME launch, csrc/rows48.h)
constexpr int BUF_SC1 = 16;
template <int AUX>
__device__ __forceinline__ f32x4_b bload4a(brsrc r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4_b, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
template <int AUX>
__device__ __forceinline__ float bload1a(brsrc r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX));
}
// NT: the non-temporal hint - bytes nobody reads before the weight gradients, a whole backward chain later (U, the activations):
// they should not push the pre-activations and Y, which the loss seeds and the dgrads read next, out of the caches
template <bool NT>
__device__ __forceinline__ void bstore4(brsrc r, uint32_t voff, uint32_t soff, f32x4_b v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, (int)voff, (int)soff, NT ? 2 : 0);
}


}  // namespace sdrm
