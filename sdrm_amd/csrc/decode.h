// VAE decode on the engine: staging kernels.  The decode itself is two launches of the MFMA GEMM
// of csrc/gemm.h (Linear -> tanh epilogue, Linear -> bias epilogue written bounds-checked into the caller's [n, items]
// matrix); these kernels bring the caller's unpadded latents and nn.Linear tensors into the zero-padded tiled layout the
// GEMM reads without bounds checks.  Reference: /root/reference/train_SDRM.py:212-214 (decoder), :252-254 (decode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

// dst[rowsP][colsP] (row-major, colsP a multiple of 4) = src[rows][cols] zero-padded; the five operands of a decode (latents, two
// weights, two biases) in ONE launch: segment = blockIdx.y (five launches of a few microseconds each were most of the decode's
// time at the small shapes, where the engine's decode lost to the PyTorch module)
struct PadSegs { const float* src[5]; float* dst[5]; int rows[5], cols[5], rowsP[5], colsP[5]; };
__global__ __launch_bounds__(256) void k_pad2d(const PadSegs sg) {
  const int k = blockIdx.y;
  const float* __restrict__ src = sg.src[k];
  float* __restrict__ dst = sg.dst[k];
  const int rows = sg.rows[k], cols = sg.cols[k], rowsP = sg.rowsP[k], colsP = sg.colsP[k];
  const int qpr = colsP >> 2;
  const int64_t total = (int64_t)rowsP * qpr;
  const bool vec = (cols & 3) == 0 && ((uintptr_t)src & 15u) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / qpr), c = 4 * (int)(i - (int64_t)r * qpr);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && c < cols) {
      const float* p = src + (size_t)r * cols + c;
      if (vec) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        v.x = p[0];
        if (c + 1 < cols) v.y = p[1];
        if (c + 2 < cols) v.z = p[2];
        if (c + 3 < cols) v.w = p[3];
      }
    }
    *reinterpret_cast<float4*>(dst + (size_t)r * colsP + c) = v;
  }
}

}  // namespace sdrm
