// Sparse batch feed on the device (dataloaders.py:46-79 + train_SDRM.py:323): the reference turns
// the CSR rows of a batch into a COO tensor on the host, ships it, and calls .to_dense() before vae.encode.  Here the
// whole CSR matrix stays resident in HBM and a batch is densified where it is consumed: one work-group per batch row
// zero-fills its [n_items] float row with 16-byte stores and scatters the row's stored values.  HBM-bound: 4 B written
// per dense element, the CSR entries read once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

struct FeedArgs {
  const int64_t* indptr; const int32_t* indices; const float* data;   // CSR of the whole feed [n_rows, n_items]
  const int64_t* rows;     // [b] row ids of this batch (null: rows row0 .. row0+b-1)
  int64_t row0, n_rows; int b, n_items;
  float* out;              // [b, n_items]
  unsigned* flag;          // the handle's feed status word: bit 0 a row id outside [0, n_rows), bit 1 a column index outside
                           // [0, n_items), bit 2 a row whose indptr pair is not ordered (read by sdrm_feed_status)
};

enum { FEED_BAD_ROW = 1u, FEED_BAD_COL = 2u, FEED_BAD_PTR = 4u };

__global__ __launch_bounds__(256) void k_csr_rows_to_dense(const FeedArgs a) {
  const int r = blockIdx.x;
  const int64_t src = a.rows ? a.rows[r] : a.row0 + r;
  float* dst = a.out + (size_t)r * a.n_items;
  // zero fill: the row start is 4-byte aligned only, so peel to a 16-byte boundary
  const int head = (int)(((16 - ((uintptr_t)dst & 15)) & 15) >> 2);
  const int h = head < a.n_items ? head : a.n_items;
  if ((int)threadIdx.x < h) dst[threadIdx.x] = 0.f;
  const int nv = (a.n_items - h) >> 2;
  float4* d4 = reinterpret_cast<float4*>(dst + h);
  for (int i = threadIdx.x; i < nv; i += 256) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = h + 4 * nv + threadIdx.x; i < a.n_items; i += 256) dst[i] = 0.f;
  __syncthreads();   // the scatter below must land after this block's own zero stores (same work-group, same row)
  // A caller's CSR is not trusted with the address of a store: a row id or column index outside the matrix leaves the output
  // row / that entry zero and raises the handle's status word instead of writing out of bounds (uniform branches; the compare
  // per entry is free beside its scattered 4-byte store).
  if (src < 0 || src >= a.n_rows) {
    if (threadIdx.x == 0) atomicOr(a.flag, (unsigned)FEED_BAD_ROW);
    return;
  }
  const int64_t p0 = a.indptr[src], p1 = a.indptr[src + 1];
  if (p0 < 0 || p1 < p0) {
    if (threadIdx.x == 0) atomicOr(a.flag, (unsigned)FEED_BAD_PTR);
    return;
  }
  bool bad = false;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
    const int32_t c = a.indices[p];
    if (c < 0 || c >= a.n_items) bad = true;
    else dst[c] = a.data ? a.data[p] : 1.f;
  }
  if (bad) atomicOr(a.flag, (unsigned)FEED_BAD_COL);
}

}  // namespace sdrm
