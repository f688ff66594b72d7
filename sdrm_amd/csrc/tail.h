// The tail of a train step in two launches (gfx950): slab reduction, the embedding path's backward and Adam with the re-pack
// of the compute copies (train_SDRM.py:336-337: what loss.backward() leaves in .grad, then diff_optim.step()).
//
// Rounds 1-3 ran four dependent launches here (k_grad_finalize -> k_emb_bwd1 -> k_emb_bwd2 -> k_adam), each a chain of a few
// dependent memory round trips: 36 us of the 469 us headline step, 23 of the 70 us ADM step.  What tied them together was the
// embedding path (train_SDRM.py:98-101: e = emb_layer(temb[t]); layer 0 multiplies [x | e]): its gradients were taken from
// dC0[t][w] = sum over the rows with timestep t of dpre0[row][w] (one-hot(t) columns in the layer-0 operand), which had to be
// reduced over the slabs before dE = dC0 * W0e, and dE before dWe = dE^T * temb.
//
// Now the layer-0 operand U carries temb[t_row] itself in its trailing columns (elementwise.h / rowchain.h / skinny_step.h),
// so the layer-0 weight-gradient slabs deliver, beside dW0[:, :L] and db0,
//     M[w][i] = sum_rows dpre0[row][w] * temb[t_row][i]                                   ([W][T]),
// and every gradient of the embedding path is a small product of M with the CURRENT parameters:
//     d dnn.0.weight[w][L + j] = sum_i M[w][i] * We[j][i] + db0[w] * be[j]
//     d emb_layer.weight[j][i] = sum_w W0e[w][j] * M[w][i]            (W0e = dnn.0.weight[:, L:])
//     d emb_layer.bias[j]      = sum_w W0e[w][j] * db0[w]
// (the chain rule through e = temb * We^T + be, re-associated; the same sums in another order).
//   k_tail     : everything that needs only the slabs: every tensor except the embedding path - slabs -> gradient -> flat g ->
//                Adam -> p, m, v -> the padded / transposed / fragment-packed compute copies - and, for the second launch,
//                M (reduced over the slabs) and a snapshot of emb_layer.* and W0e as they are BEFORE this step's update;
//   k_tail_emb : the three products above out of M, db0 (= the b0 entries of g) and the snapshot, then Adam on W0e and
//                emb_layer.*.  No work-group of either launch reads what another work-group of the same launch writes.
// `update` = 0: the gradient only (the sharded step all-reduces it, then k_adam applies it).
//
// Every work-group here is a latency chain, and one CU pulls fresh data at about 10 B per cycle: what a launch costs is the
// DEPENDENT memory round trips of its slowest work-group (about 3 us each behind another kernel's stores) and the bytes that
// work-group asks for.  So: many small work-groups (at most ~30 KB of loads each), and each requests everything it will need -
// its slab pieces, the Adam state of its elements, the operands of its product - in ONE batch before the first value is used.
// (A first version - a 32 x 32 tile or 8 rows of W0e with their 170 KB of slab pieces per work-group, p / m / v read element
// by element after the sums - took 28-38 us per launch; the four launches it replaced 23-36.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"

namespace sdrm {

enum : int { TJ_MAT = 0, TJ_VEC = 1, TJ_SCALAR = 2, TJ_SNAP = 3 };
constexpr int TAIL_MAX_JOBS = 12;
constexpr int TAIL_THREADS = 256;

struct TailJob {
  int kind;
  int rows, cols, flat_ld;      // logical region: element (r, c) at flat_off + r * flat_ld + c (VEC: rows entries, SCALAR: one)
  int64_t flat_off;             // TJ_SNAP: source = p + flat_off + r * flat_ld + c
  const float* src; size_t slab_stride; int src_ld, nslabs;   // MAT: src[s * slab_stride + r * src_ld + c]; VEC: src[s * slab_stride + i * src_ld]
  int inner;                    // SCALAR: sum of src[k * slab_stride + q], k < nslabs, q < inner
  int lanes;                    // MAT: lanes per group of four columns: 4 (8 x 32 sub-tile per work-group) or 1 (32 x 32)
  int nblocks;                  // work-groups of this job
  float* red; int red_ld;       // MAT: non-null: the sums go to red[r * red_ld + c] and nothing else happens (M); SNAP: destination, row stride
  float* dst; float* dstT; float* dstF; float* dstFT;   // compute copies (any may be null): padded [r][c], transposed [c][r], fragment-packed
  int dst_ld, dstT_ld, fnct, fklast, fklastT;
};

struct TailArgs {
  TailJob j[TAIL_MAX_JOBS];
  int start[TAIL_MAX_JOBS + 1];
  int n;
  float* p; float* m; float* v; float* g;
  int L, W, T, LP, TP;
  int64_t off_we, off_be, off_w0, off_b0;
  // the second launch's inputs, made by the first: M [W][TP] and the pre-update snapshot We [T][T] | be [T] | W0e [W][T]
  float* Mred; float* snap;
  float step_size, bc2_sqrt, b1, b2, eps, wd;
  int update;
};

// Adam on an element whose state was loaded up front: g -> flat gradient, (w, m, v) -> p, m, v; returns the (new) value
__device__ __forceinline__ float tail_apply_pre(const TailArgs& a, int64_t fi, float g, float w, float m, float v) {
  a.g[fi] = g;
  if (a.update) {
    w = adam_math(w, g, m, v, a.step_size, a.bc2_sqrt, a.b1, a.b2, a.eps, a.wd);
    a.m[fi] = m; a.v[fi] = v;
    a.p[fi] = w;
  }
  return w;
}

// slabs [kb, ke) of the float4 at p + s * stride: up to sixteen loads in flight, summed in slab order (two chains: even, odd)
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ p, size_t stride, int kb, int ke) {
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  for (int k = kb; k < ke; k += 16) {
    float4 vv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      vv[u] = (k + u < ke) ? *reinterpret_cast<const float4*>(p + (size_t)(k + u) * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      a0.x += vv[u].x; a0.y += vv[u].y; a0.z += vv[u].z; a0.w += vv[u].w;
      a1.x += vv[u + 1].x; a1.y += vv[u + 1].y; a1.z += vv[u + 1].z; a1.w += vv[u + 1].w;
    }
  }
  return make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
}
__device__ __forceinline__ float slab_sum1(const float* __restrict__ p, size_t stride, int kb, int ke) {
  float a0 = 0.f, a1 = 0.f;
  for (int k = kb; k < ke; k += 16) {
    float vv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) vv[u] = (k + u < ke) ? p[(size_t)(k + u) * stride] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; u += 2) { a0 += vv[u]; a1 += vv[u + 1]; }
  }
  return a0 + a1;
}
// the four parts of a quad of lanes meet (lanes 4 q .. 4 q + 3 of a wave): every lane gets the total, summed in a fixed tree
__device__ __forceinline__ float quad_total(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// One sub-tile of a weight matrix.  LANES = 4 (more than eight slabs): 8 rows x 32 columns, thread -> (row, column quad, part):
// the four lanes of a column quad sum a quarter of the slabs each (at most sixteen 16-byte loads in flight per lane), the parts
// meet through the wave in a fixed order, every lane then owns one element.  LANES = 1: 32 x 32, thread -> (row, column quad),
// four elements each.
template <int LANES>
__device__ __forceinline__ void tail_mat(const TailArgs& a, const TailJob& jb, int bid, float* tsh) {
  constexpr int TR = LANES == 4 ? 8 : 32, NE = LANES == 4 ? 1 : 4;
  const int tid = threadIdx.x;
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(tsh);   // [TR][33]
  const int tc = (jb.cols + 31) >> 5;
  const int r0 = (bid / tc) * TR, c0 = (bid % tc) * 32;
  const int pos = tid / LANES, part = tid % LANES;
  const int pr = pos >> 3, pq = pos & 7;
  const int R = r0 + pr, Cq = c0 + 4 * pq;
  const bool rok = R < jb.rows;
  const bool reduce_only = jb.red != nullptr;
  // everything this thread will need, requested at once
  float w[NE], m[NE], v[NE];
  bool ok[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int C = Cq + (LANES == 4 ? part : e);
    ok[e] = rok && C < jb.cols;
    const int64_t fi = jb.flat_off + (int64_t)R * jb.flat_ld + C;
    w[e] = (ok[e] && !reduce_only) ? a.p[fi] : 0.f;
    m[e] = (ok[e] && !reduce_only && a.update) ? a.m[fi] : 0.f;
    v[e] = (ok[e] && !reduce_only && a.update) ? a.v[fi] : 0.f;
  }
  const int kb = (jb.nslabs * part) / LANES, ke = (jb.nslabs * (part + 1)) / LANES;
  // (rows beyond the matrix: the loads go to row 0 and are not used - a tile may reach past the slab's padded rows)
  float4 g4 = slab_sum4(jb.src + (size_t)(rok ? R : 0) * jb.src_ld + Cq, jb.slab_stride, kb, ke);
  if (LANES == 4) { g4.x = quad_total(g4.x); g4.y = quad_total(g4.y); g4.z = quad_total(g4.z); g4.w = quad_total(g4.w); }
  const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
  if (reduce_only) {
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (ok[e]) jb.red[(size_t)R * jb.red_ld + Cq + (LANES == 4 ? part : e)] = gv[LANES == 4 ? part : e];
    return;
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int ce = LANES == 4 ? part : e;
    if (ok[e]) w[e] = tail_apply_pre(a, jb.flat_off + (int64_t)R * jb.flat_ld + Cq + ce, gv[ce], w[e], m[e], v[e]);
  }
  if (!a.update) return;
  // compute copies: only the entries of real elements are written (their padding stays as sdrm_create left it)
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int C = Cq + (LANES == 4 ? part : e);
    if (ok[e]) {
      if (jb.dst) jb.dst[(size_t)R * jb.dst_ld + C] = w[e];
      if (jb.dstF) jb.dstF[wfrag_index(R, C, jb.fnct, jb.fklast)] = w[e];
    }
  }
  if (jb.dstT == nullptr && jb.dstFT == nullptr) return;
#pragma unroll
  for (int e = 0; e < NE; ++e) tile[pr][4 * pq + (LANES == 4 ? part : e)] = w[e];
  __syncthreads();
  // transposed: thread -> (column of the tile, row): the rows of a column are consecutive addresses of the transposed copies
  for (int f = tid; f < TR * 32; f += TAIL_THREADS) {
    const int cc = f / TR, rr = f - cc * TR;
    const int c = c0 + cc, r = r0 + rr;
    if (c < jb.cols && r < jb.rows) {
      const float t1 = tile[rr][cc];
      if (jb.dstT) jb.dstT[(size_t)c * jb.dstT_ld + r] = t1;
      if (jb.dstFT) jb.dstFT[wfrag_index(c, r, jb.fnct, jb.fklastT)] = t1;
    }
  }
}

__global__ __launch_bounds__(TAIL_THREADS) void k_tail(const TailArgs a) {
  __shared__ __attribute__((aligned(16))) float tsh[32 * 33];
  const int tid = threadIdx.x;
  int k = 0;
#pragma unroll
  for (int q = 1; q < TAIL_MAX_JOBS; ++q)
    if (q < a.n && (int)blockIdx.x >= a.start[q]) k = q;
  const TailJob& jb = a.j[k];
  const int bid = (int)blockIdx.x - a.start[k];

  if (jb.kind == TJ_MAT) {
    if (jb.lanes == 4) tail_mat<4>(a, jb, bid, tsh);
    else tail_mat<1>(a, jb, bid, tsh);
    return;
  }

  if (jb.kind == TJ_VEC) {   // four lanes per entry, 64 entries per work-group
    const int i = bid * (TAIL_THREADS / 4) + (tid >> 2), part = tid & 3;
    const bool ok = i < jb.rows;
    const int ii = ok ? i : 0;
    const int64_t fi = jb.flat_off + ii;
    const float w0 = a.p[fi], m0 = a.update ? a.m[fi] : 0.f, v0 = a.update ? a.v[fi] : 0.f;
    const float g = quad_total(slab_sum1(jb.src + (size_t)ii * jb.src_ld, jb.slab_stride, (jb.nslabs * part) / 4, (jb.nslabs * (part + 1)) / 4));
    if (ok && part == 0) {
      const float w = tail_apply_pre(a, fi, g, w0, m0, v0);
      if (a.update && jb.dst) jb.dst[i] = w;
    }
    return;
  }

  if (jb.kind == TJ_SNAP) {   // the pre-update snapshot of a parameter block [rows][cols] (read by the second launch)
    const int64_t total = (int64_t)jb.rows * jb.cols;
    const int64_t i0 = (int64_t)bid * (TAIL_THREADS * 8) + tid;
    float vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * TAIL_THREADS;
      vv[u] = i < total ? a.p[jb.flat_off + (i / jb.cols) * jb.flat_ld + (i % jb.cols)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * TAIL_THREADS;
      if (i < total) jb.red[(i / jb.cols) * jb.red_ld + (i % jb.cols)] = vv[u];
    }
    return;
  }

  // ---- TJ_SCALAR: thousands of partials (one per dgrad work-group and application): every thread takes up to 16, all in flight
  const float w0 = a.p[jb.flat_off], m0 = a.update ? a.m[jb.flat_off] : 0.f, v0 = a.update ? a.v[jb.flat_off] : 0.f;
  float s = 0.f;
  const int total = jb.nslabs * jb.inner;
  for (int i0 = 0; i0 < total; i0 += TAIL_THREADS * 16) {
    float vv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u * TAIL_THREADS + tid;
      vv[u] = i < total ? jb.src[(size_t)(i / jb.inner) * jb.slab_stride + (i % jb.inner)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += vv[u];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((tid & 63) == 0) tsh[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) tail_apply_pre(a, jb.flat_off, (tsh[0] + tsh[1]) + (tsh[2] + tsh[3]), w0, m0, v0);
}

// ---- second launch: the embedding path.  Work-groups [0, nA): 8 rows x 32 columns of W0e; [nA, nA + nB): 16 x 16 tiles of
// emb_layer.weight; the last one: emb_layer.bias.
constexpr int TE_RB = 8, TE_JT = 32;     // W0e work-groups: rows, columns
constexpr int TE_WT = 16, TE_WC = 128;   // emb_layer.weight work-groups: tile edge, rows of the contraction per chunk
__host__ __device__ inline int tail_emb_blocks_a(int W, int T) { return ((W + TE_RB - 1) / TE_RB) * ((T + TE_JT - 1) / TE_JT); }
__host__ __device__ inline int tail_emb_blocks_b(int T) { return ((T + TE_WT - 1) / TE_WT) * ((T + TE_WT - 1) / TE_WT); }
__host__ __device__ inline size_t tail_emb_lds_floats(int T, int TP) {
  const size_t a_ = (size_t)TE_JT * (T + 1) + (size_t)TE_RB * TP + TE_RB + TE_JT;
  const size_t b_ = (size_t)2 * 2 * TE_WC * (TE_WT + 1);
  return (a_ > b_ ? a_ : b_) + 8;
}

__global__ __launch_bounds__(256) void k_tail_emb(const TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float esh[];
  const int tid = threadIdx.x, T = a.T, TP = a.TP, ldw = a.L + a.T;
  const float* snapWe = a.snap;                       // [T][T]
  const float* snapbe = a.snap + (size_t)T * T;       // [T]
  const float* snapW0e = snapbe + T;                  // [W][T]
  const int nA = tail_emb_blocks_a(a.W, T), nB = tail_emb_blocks_b(T);
  int bid = blockIdx.x;
  if (bid < nA) {
    // d dnn.0.weight[w][L + j] = sum_i M[w][i] * We[j][i] + db0[w] * be[j] for rows w0 .. w0 + 7, columns j0 .. j0 + 31:
    // thread -> (row tid / 32, column tid % 32)
    const int ntj = (T + TE_JT - 1) / TE_JT;
    const int w0 = (bid / ntj) * TE_RB, j0 = (bid % ntj) * TE_JT;
    float* Ms = esh;                                  // [8][TP] (16-byte rows)
    float* WeS = Ms + TE_RB * TP;                     // [32][T + 1]
    float* db0S = WeS + TE_JT * (T + 1);              // [8]
    float* beS = db0S + TE_RB;                        // [32]
    const int r = tid >> 5, jj = tid & 31;
    const int w = w0 + r, j = j0 + jj;
    const bool own = w < a.W && j < T;
    const int64_t fi = a.off_w0 + (int64_t)(own ? w : 0) * ldw + a.L + (own ? j : 0);
    // one batch of requests: own Adam state, the 8 rows of M, db0, be, the 32 rows of We (coalesced along i)
    const float ow = a.p[fi], om = a.update ? a.m[fi] : 0.f, ov = a.update ? a.v[fi] : 0.f;
    const int nmq = TE_RB * TP / 4;
    float4 mq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < nmq && w0 + tid / (TP / 4) < a.W) mq = *reinterpret_cast<const float4*>(a.Mred + (size_t)(w0 + tid / (TP / 4)) * TP + 4 * (tid % (TP / 4)));
    const float dbv = (tid < TE_RB && w0 + tid < a.W) ? a.g[a.off_b0 + w0 + tid] : 0.f;
    const float bev = (tid >= 32 && tid < 64 && j0 + tid - 32 < T) ? snapbe[j0 + tid - 32] : 0.f;
    const int nwe = TE_JT * T;
    for (int f0 = 0; f0 < nwe; f0 += 16 * 256) {
      float vv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = f0 + u * 256 + tid;
        const int jr = f / T, i = f - jr * T;
        vv[u] = (f < nwe && j0 + jr < T) ? snapWe[(size_t)(j0 + jr) * T + i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = f0 + u * 256 + tid;
        if (f < nwe) WeS[(f / T) * (T + 1) + (f % T)] = vv[u];
      }
    }
    if (tid < nmq) *reinterpret_cast<float4*>(Ms + 4 * tid) = mq;
    for (int f = tid + 256; f < nmq; f += 256) {   // TP > 128
      const int rr = f / (TP / 4), q = f - rr * (TP / 4);
      *reinterpret_cast<float4*>(Ms + 4 * f) =
          (w0 + rr < a.W) ? *reinterpret_cast<const float4*>(a.Mred + (size_t)(w0 + rr) * TP + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < TE_RB) db0S[tid] = dbv;
    if (tid >= 32 && tid < 64) beS[tid - 32] = bev;
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    const float* mr = Ms + r * TP;
    const float* wr = WeS + jj * (T + 1);
    int i = 0;
    for (; i + 1 < T; i += 2) { s0 = fmaf(mr[i], wr[i], s0); s1 = fmaf(mr[i + 1], wr[i + 1], s1); }
    if (i < T) s0 = fmaf(mr[i], wr[i], s0);
    if (own) tail_apply_pre(a, fi, (s0 + s1) + db0S[r] * beS[jj], ow, om, ov);
    return;
  }
  bid -= nA;
  if (bid < nB) {
    // d emb_layer.weight[j][i] = sum_w W0e[w][j] * M[w][i]: a 16 x 16 tile, thread -> (j, i); the contraction in chunks of 128
    // rows through LDS, the next chunk's loads in flight while this one is multiplied
    const int nt = (T + TE_WT - 1) / TE_WT;
    const int j0 = (bid / nt) * TE_WT, i0 = (bid % nt) * TE_WT;
    const int jj = tid >> 4, ii = tid & 15;
    const bool own = j0 + jj < T && i0 + ii < T;
    const int64_t fi = a.off_we + (int64_t)(own ? j0 + jj : 0) * T + (own ? i0 + ii : 0);
    const float ow = a.p[fi], om = a.update ? a.m[fi] : 0.f, ov = a.update ? a.v[fi] : 0.f;
    constexpr int PER = TE_WC * TE_WT / 256;   // 8 floats of each operand per thread and chunk: (row f / 16, column f % 16)
    float xa[PER], xb[PER];
    auto load = [&](int wc0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = u * 256 + tid, wr_ = wc0 + (f >> 4), cc = f & 15;
        xa[u] = (wr_ < a.W && j0 + cc < T) ? snapW0e[(size_t)wr_ * T + j0 + cc] : 0.f;
        xb[u] = (wr_ < a.W && i0 + cc < T) ? a.Mred[(size_t)wr_ * TP + i0 + cc] : 0.f;
      }
    };
    float acc0 = 0.f, acc1 = 0.f;
    load(0);
    int buf = 0;
    for (int wc0 = 0; wc0 < a.W; wc0 += TE_WC, buf ^= 1) {
      float* As = esh + buf * 2 * TE_WC * (TE_WT + 1);
      float* Bs = As + TE_WC * (TE_WT + 1);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = u * 256 + tid;
        As[(f >> 4) * (TE_WT + 1) + (f & 15)] = xa[u];
        Bs[(f >> 4) * (TE_WT + 1) + (f & 15)] = xb[u];
      }
      if (wc0 + TE_WC < a.W) load(wc0 + TE_WC);
      __syncthreads();   // (two LDS buffers: the chunk written now was last read two iterations ago, behind the barrier in between)
#pragma unroll 8
      for (int q = 0; q < TE_WC; q += 2) {
        acc0 = fmaf(As[q * (TE_WT + 1) + jj], Bs[q * (TE_WT + 1) + ii], acc0);
        acc1 = fmaf(As[(q + 1) * (TE_WT + 1) + jj], Bs[(q + 1) * (TE_WT + 1) + ii], acc1);
      }
    }
    if (own) tail_apply_pre(a, fi, acc0 + acc1, ow, om, ov);
    return;
  }
  // d emb_layer.bias[j] = sum_w W0e[w][j] * db0[w]: thread -> j (and j + 256 ..), sixteen rows of the contraction in flight
  for (int j = tid; j < T; j += 256) {
    const int64_t fi = a.off_be + j;
    const float ow = a.p[fi], om = a.update ? a.m[fi] : 0.f, ov = a.update ? a.v[fi] : 0.f;
    float s0 = 0.f, s1 = 0.f;
    for (int w0 = 0; w0 < a.W; w0 += 16) {
      float xv[16], dv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        xv[u] = w0 + u < a.W ? snapW0e[(size_t)(w0 + u) * T + j] : 0.f;
        dv[u] = w0 + u < a.W ? a.g[a.off_b0 + w0 + u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; u += 2) { s0 = fmaf(xv[u], dv[u], s0); s1 = fmaf(xv[u + 1], dv[u + 1], s1); }
    }
    tail_apply_pre(a, fi, s0 + s1, ow, om, ov);
  }
}

}  // namespace sdrm
