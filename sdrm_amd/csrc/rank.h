// Recall@k / NDCG@k of a score matrix against held-out interactions, on the device (SURVEY.md §8f-3;
// utilities.py:116-171: mask_training_examples, recall_at_k_batch, NDCG_binary_at_k_batch).
//
// The reference partitions every row for its top-k (bn.argpartition) and then looks the held-out items up in it.
// Only the held-out items matter, and there are few of them per user, so no top-k is built here: one work-group
// per user keeps the user's score row in LDS (already-seen items set to -inf, utilities.py:118-119) and computes
// for every held-out item j its rank = #{i : s_i > s_j} + #{i < j : s_i == s_j} (ties towards the lower index;
// argpartition leaves them unspecified).  rank < k  <=>  the item is in the top-k, and its DCG discount is
// tp[rank].  One pass serves every k of the list.  HBM-bound: the score matrix is read once.
//
// Bit-exactness with numpy: hits are small integers (exact in float32 as the reference holds them); recall is
// hits / min(k, n_true) in float64; DCG is the row sum of a length-k float64 vector, summed exactly the way
// numpy's pairwise add-reduce does (8 strided accumulators), with the discount table tp and the ideal-DCG table
// computed by the caller with numpy and passed in.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrm {

constexpr int RANK_MAX_K = 128;   // numpy's pairwise block: one level, no recursion
constexpr int RANK_MAX_NK = 8;

struct RankArgs {
  const float* scores; int U, I;
  const int64_t* held_indptr; const int32_t* held_indices;
  const int64_t* train_indptr; const int32_t* train_indices;   // may be null (no masking)
  int ks[RANK_MAX_NK]; int nk, kmax;
  const double* tp;     // [kmax]   1 / log2(r + 2)
  const double* idcg;   // [kmax+1] sum(tp[:m]) as numpy sums it
  double* recall;       // [nk][U]
  double* ndcg;         // [nk][U]
};

// numpy's pairwise_sum for n <= 128 (numpy/_core/src/umath/loops_utils.h.src)
__device__ inline double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

__global__ __launch_bounds__(256) void k_rank_metrics(const RankArgs a) {
  extern __shared__ float row[];            // [I] scores of this user, then the reduction scratch
  __shared__ int red[4];
  __shared__ double vec[RANK_MAX_K];        // hit * discount by rank
  __shared__ unsigned char hit[RANK_MAX_K];
  const int u = blockIdx.x, tid = threadIdx.x;
  const float* src = a.scores + (size_t)u * a.I;
  for (int i = tid; i < a.I; i += 256) row[i] = src[i];
  for (int r = tid; r < RANK_MAX_K; r += 256) hit[r] = 0;
  __syncthreads();
  if (a.train_indptr) {
    const int64_t t0 = a.train_indptr[u], t1 = a.train_indptr[u + 1];
    for (int64_t t = t0 + tid; t < t1; t += 256) row[a.train_indices[t]] = -INFINITY;
    __syncthreads();
  }
  const int64_t h0 = a.held_indptr[u], h1 = a.held_indptr[u + 1];
  const int n_true = (int)(h1 - h0);
  for (int64_t h = h0; h < h1; ++h) {
    const int j = a.held_indices[h];
    const float v = row[j];
    int c = 0;
    for (int i = tid; i < a.I; i += 256) {
      const float s = row[i];
      c += (s > v) || (s == v && i < j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
      const int rank = red[0] + red[1] + red[2] + red[3];
      if (rank < a.kmax) hit[rank] = 1;
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int q = 0; q < a.nk; ++q) {
      const int k = a.ks[q];
      int hits = 0;
      for (int r = 0; r < k; ++r) {
        vec[r] = hit[r] ? a.tp[r] : 0.0;
        hits += hit[r];
      }
      const double dcg = np_pairwise_sum(vec, k);
      const int m = n_true < k ? n_true : k;
      a.recall[(size_t)q * a.U + u] = (double)hits / (double)m;          // 0/0 -> nan, like the reference
      a.ndcg[(size_t)q * a.U + u] = dcg / a.idcg[m];                      // idcg[0] = 0 -> nan
    }
  }
}

}  // namespace sdrm
