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
  int lanes;                    // MAT: lanes per group of four columns: 4 (16 x 16 sub-tile per work-group), 2 (16 x 32) or 1 (32 x 32)
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
  // narrow nets: padded copies of emb_layer.weight [TPe][TPe] and of W0e [WP][TPe] (TPe = T rounded up to 16) that the forward
  // multiplies the users' time-embedding rows with (skinny_step.h); null otherwise
  float* WeP; float* W0eP; int TPe;
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

// slabs [kb, ke) of the float4 at p + s * stride: up to BATCH loads in flight, summed in slab order (two chains: even, odd)
template <int BATCH>
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ p, size_t stride, int kb, int ke) {
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  for (int k = kb; k < ke; k += BATCH) {
    float4 vv[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      vv[u] = (k + u < ke) ? *reinterpret_cast<const float4*>(p + (size_t)(k + u) * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < BATCH; u += 2) {
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

// One sub-tile of a weight matrix, thread -> (row, column quad, part): the LANES lanes of a column quad sum 1 / LANES of the slabs
// each (at most BATCH 16-byte loads in flight per lane), the parts meet through the wave in a fixed order, every lane then owns
// 4 / LANES of the quad's elements.  LANES = 4: 16 rows x 16 columns per work-group (more than 32 slabs); 2: 16 x 32 (a row of
// the sub-tile is a whole 128-byte line of every slab: 64-byte pieces ran the slab read at 2.7 TB/s); 1: 32 x 32 (at most eight
// slabs).  The compute copies are written out of an LDS image of the updated tile, each in the order that makes ITS addresses
// consecutive: a 16 x 16 block is one contiguous 1 KiB piece of a fragment-packed copy (wfrag_index: [k / 16][n / 16]
// [(k / 4) % 4][n % 16][k % 4]), 64-byte runs or more of the padded and the transposed copies.
template <int LANES, int BATCH>
__device__ __forceinline__ void tail_mat(const TailArgs& a, const TailJob& jb, int bid, float* tsh) {
  constexpr int TSR = LANES == 1 ? 32 : 16, TSC = LANES == 4 ? 16 : 32, QPR = TSC / 4, NE = 4 / LANES, NBC = TSC / 16;
  const int tid = threadIdx.x;
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(tsh);   // [TSR][33]
  const int tc = (jb.cols + TSC - 1) / TSC;
  const int r0 = (bid / tc) * TSR, c0 = (bid % tc) * TSC;
  const int pos = tid / LANES, part = tid % LANES;
  const int pr = pos / QPR, pq = pos % QPR;
  const int R = r0 + pr, Cq = c0 + 4 * pq;
  const bool rok = R < jb.rows;
  const bool reduce_only = jb.red != nullptr;
  // Two thread -> element mappings.  The slab pieces are 16-byte loads: thread -> (row, column quad, part).  The flat vectors
  // p / m / v / g have rows of flat_ld floats (not 16-byte aligned in general), so there the NE elements of a thread are element
  // tid + 256 u of the sub-tile, row-major: a wave instruction touches whole 64- / 128-byte runs (with the quad mapping every
  // instruction touched a quarter of each line it asked for, four instructions per line: the 830-wide net's tail ran at 2.7 TB/s).
  // The sums change mapping through the LDS tile.  Everything this thread will need is requested at once:
  float w[NE], m[NE], v[NE];
  bool ok[NE];
  int64_t fi[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int f = tid + TAIL_THREADS * u, rr = f / TSC, cc = f % TSC;
    ok[u] = !reduce_only && r0 + rr < jb.rows && c0 + cc < jb.cols;
    fi[u] = jb.flat_off + (int64_t)(r0 + rr) * jb.flat_ld + c0 + cc;
    w[u] = ok[u] ? a.p[fi[u]] : 0.f;
    m[u] = (ok[u] && a.update) ? a.m[fi[u]] : 0.f;
    v[u] = (ok[u] && a.update) ? a.v[fi[u]] : 0.f;
  }
  const int kb = (jb.nslabs * part) / LANES, ke = (jb.nslabs * (part + 1)) / LANES;
  // (rows beyond the matrix: the loads go to row 0 and are not used - a tile may reach past the slab's padded rows)
  float4 g4 = slab_sum4<BATCH>(jb.src + (size_t)(rok ? R : 0) * jb.src_ld + Cq, jb.slab_stride, kb, ke);
  if (LANES >= 2) { g4.x += __shfl_xor(g4.x, 1, 64); g4.y += __shfl_xor(g4.y, 1, 64); g4.z += __shfl_xor(g4.z, 1, 64); g4.w += __shfl_xor(g4.w, 1, 64); }
  if (LANES == 4) { g4.x += __shfl_xor(g4.x, 2, 64); g4.y += __shfl_xor(g4.y, 2, 64); g4.z += __shfl_xor(g4.z, 2, 64); g4.w += __shfl_xor(g4.w, 2, 64); }
  const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
  if (reduce_only) {
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (rok && Cq + part * NE + e < jb.cols) jb.red[(size_t)R * jb.red_ld + Cq + part * NE + e] = gv[part * NE + e];
    return;
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) tile[pr][4 * pq + part * NE + e] = gv[part * NE + e];
  lds_barrier();
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int f = tid + TAIL_THREADS * u, rr = f / TSC, cc = f % TSC;
    if (ok[u]) {
      w[u] = tail_apply_pre(a, fi[u], tile[rr][cc], w[u], m[u], v[u]);
      if (a.update && jb.dst) jb.dst[(size_t)(r0 + rr) * jb.dst_ld + c0 + cc] = w[u];   // the padded copy, whole rows of the sub-tile
    }
    tile[rr][cc] = w[u];   // (the entry this thread has just read: no other thread touches it before the barrier below)
  }
  if (!a.update) return;
  if (jb.dstT == nullptr && jb.dstF == nullptr && jb.dstFT == nullptr) return;
  lds_barrier();   // (not __syncthreads: that would wait for the p / m / v / g stores above)
  // the other copies out of the LDS image (only the entries of real elements: their padding stays as sdrm_create left it)
  for (int f = tid; f < TSR * TSC; f += TAIL_THREADS) {
    if (jb.dstT) {   // [c][r]: consecutive threads -> consecutive rows of one column
      const int cc = f / TSR, rr = f - cc * TSR;
      if (c0 + cc < jb.cols && r0 + rr < jb.rows) jb.dstT[(size_t)(c0 + cc) * jb.dstT_ld + r0 + rr] = tile[rr][cc];
    }
    // fragment-packed: 16 x 16 block sb of the tile; inside it thread order (k group, n, k % 4) = the copy's address order
    const int sb = f >> 8, e = f & 3, nn = (f >> 2) & 15, kg = (f >> 6) & 3;
    const int sr = sb / NBC, sc = sb % NBC;
    if (jb.dstF) {   // element (n = row, k = column)
      const int rr = 16 * sr + nn, cc = 16 * sc + 4 * kg + e;
      if (r0 + rr < jb.rows && c0 + cc < jb.cols) jb.dstF[wfrag_index(r0 + rr, c0 + cc, jb.fnct, jb.fklast)] = tile[rr][cc];
    }
    if (jb.dstFT) {  // element (n = column, k = row)
      const int cc = 16 * sc + nn, rr = 16 * sr + 4 * kg + e;
      if (r0 + rr < jb.rows && c0 + cc < jb.cols) jb.dstFT[wfrag_index(c0 + cc, r0 + rr, jb.fnct, jb.fklastT)] = tile[rr][cc];
    }
  }
}

// BATCH: 16-byte slab loads a lane keeps in flight: 8 when no lane of the launch has more than eight slabs to sum (the host
// checks): the kernel then fits 64 registers and eight work-groups share a CU - at ML-1M all 1600 are resident at once
template <int BATCH>
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
    if (jb.lanes == 4) tail_mat<4, BATCH>(a, jb, bid, tsh);
    else if (jb.lanes == 2) tail_mat<2, BATCH>(a, jb, bid, tsh);
    else tail_mat<1, BATCH>(a, jb, bid, tsh);
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
    // a work-group takes 2048 consecutive elements of one row (the host cuts rows into such pieces): no index division here -
    // an emulated 64-bit division per element was most of this job
    const int ppr = (jb.cols + TAIL_THREADS * 8 - 1) / (TAIL_THREADS * 8);   // pieces per row
    const int r = bid / ppr, c0 = (bid - r * ppr) * (TAIL_THREADS * 8);
    const float* src = a.p + jb.flat_off + (int64_t)r * jb.flat_ld + c0;
    float* dst = jb.red + (int64_t)r * jb.red_ld + c0;
    const int n = min(TAIL_THREADS * 8, jb.cols - c0);
    float vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) vv[u] = tid + u * TAIL_THREADS < n ? src[tid + u * TAIL_THREADS] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (tid + u * TAIL_THREADS < n) dst[tid + u * TAIL_THREADS] = vv[u];
    return;
  }

  // ---- TJ_SCALAR: thousands of partials (one per dgrad work-group and application): every thread takes up to 16, all in flight
  const float w0 = a.p[jb.flat_off], m0 = a.update ? a.m[jb.flat_off] : 0.f, v0 = a.update ? a.v[jb.flat_off] : 0.f;
  float s = 0.f;
  for (int k = 0; k < jb.nslabs; ++k) {   // application k: `inner` partials, sixteen per thread in flight
    const float* src = jb.src + (size_t)k * jb.slab_stride;
    for (int i0 = 0; i0 < jb.inner; i0 += TAIL_THREADS * 16) {
      float vv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = i0 + u * TAIL_THREADS + tid;
        vv[u] = i < jb.inner ? src[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += vv[u];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((tid & 63) == 0) tsh[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) tail_apply_pre(a, jb.flat_off, (tsh[0] + tsh[1]) + (tsh[2] + tsh[3]), w0, m0, v0);
}

// ---- second launch: the embedding path.  Work-groups [0, nA): 8 rows x 32 columns of W0e; [nA, nA + nB): 8 x 8 tiles of
// emb_layer.weight; then 16 entries of emb_layer.bias each.
constexpr int TE_RB = 8, TE_JT = 32;     // W0e work-groups: rows, columns
constexpr int TE_WT = 8, TE_WC = 256;    // emb_layer.weight work-groups: tile edge, rows of the contraction per chunk
__host__ __device__ inline int tail_emb_blocks_a(int W, int T) { return ((W + TE_RB - 1) / TE_RB) * ((T + TE_JT - 1) / TE_JT); }
__host__ __device__ inline int tail_emb_blocks_b(int T) { return ((T + TE_WT - 1) / TE_WT) * ((T + TE_WT - 1) / TE_WT); }
__host__ __device__ inline int tail_emb_blocks_c(int T) { return (T + 15) / 16; }
__host__ __device__ inline size_t tail_emb_lds_floats(int T, int TP) {
  const size_t a_ = (size_t)TE_JT * (T + 2) + (size_t)TE_RB * TP + TE_RB + TE_JT;
  const size_t b_ = (size_t)2 * TE_WC * (TE_WT + 2);
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
    const int ldws = (T + 1) | 1;                     // odd: the 32 lanes of a row read 32 different banks
    float* WeS = Ms + TE_RB * TP;                     // [32][ldws]
    float* db0S = WeS + TE_JT * ldws;                 // [8]
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
    // We rows j0 .. j0 + 31: wave w takes rows w, w + 4, .. (eight of them), its lanes the columns i = lane, lane + 64, ..
    // (coalesced, and no index division: an emulated division per element was most of this work-group's time)
    {
      const int wv = tid >> 6, ln = tid & 63;
      for (int i0 = 0; i0 < T; i0 += 128) {
        float vv[8][2];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int jr = wv + 4 * rr, i = i0 + ln + 64 * h;
            vv[rr][h] = (j0 + jr < T && i < T) ? snapWe[(size_t)(j0 + jr) * T + i] : 0.f;
          }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int jr = wv + 4 * rr, i = i0 + ln + 64 * h;
            if (i < T) WeS[jr * ldws + i] = vv[rr][h];
          }
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
    const float* wr = WeS + jj * ldws;
    int i = 0;
    for (; i + 1 < T; i += 2) { s0 = fmaf(mr[i], wr[i], s0); s1 = fmaf(mr[i + 1], wr[i + 1], s1); }
    if (i < T) s0 = fmaf(mr[i], wr[i], s0);
    if (own) {
      const float wn = tail_apply_pre(a, fi, (s0 + s1) + db0S[r] * beS[jj], ow, om, ov);
      if (a.update && a.W0eP) a.W0eP[(size_t)w * a.TPe + j] = wn;
    }
    return;
  }
  bid -= nA;
  if (bid < nB) {
    // d emb_layer.weight[j][i] = sum_w W0e[w][j] * M[w][i]: an 8 x 8 tile per work-group (what a work-group costs is the bytes it
    // pulls, W x (8 + 8) floats: with 16 x 16 tiles the 106 KB of W = 830 were 14 us, the whole launch); the contraction in
    // chunks of 256 rows through LDS (the next chunk's loads in flight while this one is multiplied).  Sixteen groups of 16 lanes
    // take rows g, g + 16, .. of a chunk for ALL 64 outputs, 2 x 2 of them per lane (two 8-byte LDS reads per four
    // multiply-adds); the groups' sums meet in LDS, in a fixed order.
    const int nt = (T + TE_WT - 1) / TE_WT;
    const int j0 = (bid / nt) * TE_WT, i0 = (bid % nt) * TE_WT;
    const int jj = (tid >> 3) & 7, ii = tid & 7;            // thread tid < 64: its OUTPUT (after the groups' sums have met)
    const bool own = tid < TE_WT * TE_WT && j0 + jj < T && i0 + ii < T;
    const int64_t fi = a.off_we + (int64_t)(own ? j0 + jj : 0) * T + (own ? i0 + ii : 0);
    const float ow = own ? a.p[fi] : 0.f, om = (own && a.update) ? a.m[fi] : 0.f, ov = (own && a.update) ? a.v[fi] : 0.f;
    constexpr int PER = TE_WC * TE_WT / 256;   // 8 floats of each operand per thread and chunk: (row f / 8, column f % 8)
    constexpr int LDT = TE_WT + 2;             // LDS row stride: even (8-byte reads), 10 floats
    float xa[PER], xb[PER];
    auto load = [&](int wc0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = u * 256 + tid, wr_ = wc0 + (f >> 3), cc = f & 7;
        xa[u] = (wr_ < a.W && j0 + cc < T) ? snapW0e[(size_t)wr_ * T + j0 + cc] : 0.f;
        xb[u] = (wr_ < a.W && i0 + cc < T) ? a.Mred[(size_t)wr_ * TP + i0 + cc] : 0.f;
      }
    };
    const int grp = tid >> 4, jq = (tid >> 2) & 3, iq = tid & 3;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    load(0);
    float* As = esh;
    float* Bs = As + TE_WC * LDT;
    for (int wc0 = 0; wc0 < a.W; wc0 += TE_WC) {
      if (wc0 > 0) lds_barrier();   // every wave is done with the previous chunk
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = u * 256 + tid;
        As[(f >> 3) * LDT + (f & 7)] = xa[u];
        Bs[(f >> 3) * LDT + (f & 7)] = xb[u];
      }
      if (wc0 + TE_WC < a.W) load(wc0 + TE_WC);
      lds_barrier();     // (LDS only: the next chunk's loads stay in flight)
#pragma unroll 8
      for (int q = grp; q < TE_WC; q += 16) {
        const float2 av = *reinterpret_cast<const float2*>(As + q * LDT + 2 * jq);
        const float2 bv = *reinterpret_cast<const float2*>(Bs + q * LDT + 2 * iq);
        acc[0][0] = fmaf(av.x, bv.x, acc[0][0]); acc[0][1] = fmaf(av.x, bv.y, acc[0][1]);
        acc[1][0] = fmaf(av.y, bv.x, acc[1][0]); acc[1][1] = fmaf(av.y, bv.y, acc[1][1]);
      }
    }
    lds_barrier();   // every wave is done with the operand buffers: they become the groups' partial sums [16][8][8]
    float* red = esh;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) red[grp * 64 + (2 * jq + x) * 8 + 2 * iq + y] = acc[x][y];
    lds_barrier();
    if (own) {
      float s4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) s4[k] = (red[(4 * k) * 64 + tid] + red[(4 * k + 1) * 64 + tid]) + (red[(4 * k + 2) * 64 + tid] + red[(4 * k + 3) * 64 + tid]);
      const float wn = tail_apply_pre(a, fi, (s4[0] + s4[1]) + (s4[2] + s4[3]), ow, om, ov);
      if (a.update && a.WeP) a.WeP[(size_t)(j0 + jj) * a.TPe + i0 + ii] = wn;
    }
    return;
  }
  // d emb_layer.bias[j] = sum_w W0e[w][j] * db0[w]: 16 columns per work-group, thread -> (column tid % 16, slice tid / 16 of the
  // rows: w = slice, slice + 16, ..): up to 32 rows per thread in flight at once (a single thread per column walking all W rows
  // in batches was a chain of W / 16 dependent round trips: 11 us at W = 340, 25 at 830 - the whole launch)
  bid -= nB;
  {
    const int jj = tid & 15, sl = tid >> 4, j = bid * 16 + jj;
    const bool jok = j < T;
    const int64_t fi = a.off_be + (jok ? j : 0);
    const float ow = a.p[fi], om = a.update ? a.m[fi] : 0.f, ov = a.update ? a.v[fi] : 0.f;
    float s0 = 0.f, s1 = 0.f;
    for (int w0 = sl; w0 < a.W; w0 += 16 * 32) {
      float xv[32], dv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int w = w0 + 16 * u;
        xv[u] = (w < a.W && jok) ? snapW0e[(size_t)w * T + j] : 0.f;
        dv[u] = w < a.W ? a.g[a.off_b0 + w] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 32; u += 2) { s0 = fmaf(xv[u], dv[u], s0); s1 = fmaf(xv[u + 1], dv[u + 1], s1); }
    }
    float* red = esh;   // [16 slices][16 columns]
    red[tid] = s0 + s1;
    lds_barrier();
    if (tid < 16 && jok) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += red[16 * q + tid];
      tail_apply_pre(a, fi, s, ow, om, ov);
    }
  }
}

}  // namespace sdrm
