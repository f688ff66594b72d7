// EXPERIMENT (round 4, not adopted; kept as a record): the mid-width nets' full-resolution sampling loop in ONE persistent launch.
// Wired in, it needed three small hooks that were reverted with it: bufres.h `bload4c<AUX>` (raw buffer load with cache-policy bits),
// gemm.h `load_tile<.., AUX>` + `gemm_body<.., COHA>(p, bid, logical_in)` (A loads with sc1 = 16, the caller's tile index), and in
// sdrm_hip.hip a launch from sdrm_sample_begin behind `sample_persist_fits()` (PHILOX, full resolution, L == W, one row chain,
// 384 .. 528 work-groups of the 64x64 tile), counters + abort flag in a small device buffer, the abort flag read at sdrm_sample_end.
// Parity-green on the sampling tests (54 passed).  Measured (MI355X, ML-1M net, 78 steps, rocprofv3):
//   n = 5429 (85 row tiles: five XCDs get 11 row tiles = 66 work-groups for 32 CUs, two CUs with three): 4680 us = 60.0 us per step
//   n = 5120 (80 row tiles, 60 work-groups per XCD, balanced):                                            3659 us = 46.9 us per step
//   per-layer path at n = 5120: 13.9 + 13.9 + 14.4 + 6.2 (k_reverse_update) = 48.4 us per step + gaps (49.6 at n = 5429)
// A phase of the persistent kernel costs 15.6 us against 14.4 us for a launch: back-to-back launches already overlap most of a
// kernel boundary (the next launch's ramp with the previous one's drain), the row-tile barrier waits for the slowest of six tiles,
// and the tile body's prologue is paid either way.  A row tile's six work-groups must share an XCD, and 510 tiles do not divide
// into eight XCD shares of at most 64: the headline size loses 20 %.  Not adopted.
// The full-resolution reverse-sampling loop of a mid-width eps-net (sample_ddpm, train_SDRM.py:50-59) in ONE persistent launch
// (round 4): every step's H + 2 forward GEMMs and the DDPM reverse update (denoise_add_noise, :20-25, inside the out layer's
// epilogue, EPI_TANH_REV) for all T steps, with no kernel boundary in between.
//
// The per-layer path pays per reverse step three launches of ~14.4 us for 8.6 us of matrix-pipe need plus a stand-alone
// k_reverse_update (6.2 us): a kernel boundary is ~4.5 us on this chip and each launch has its own ramp, prologue and drain.
// What couples the launches is narrow: layer k + 1 of row tile R needs only the column tiles of row tile R of layer k.  Round 1
// measured that hand-shake at agent scope - 4.75 us per phase, more than a boundary - and dropped the idea (DESIGN.md section 4d
// (x)).  Inside ONE XCD it is cheap (tools/rowsync_probe.hip, profiles/r04_rowsync_probe.txt: 0.84 us): the L2 of an XCD is
// coherent for its own CUs, so
//   * row tile R lives on XCD R % 8, all its column tiles (work-group b runs on XCD b & 7; slot b / 8 of that XCD = (row tile
//     index on the XCD, column tile)): nothing has to be visible beyond that L2;
//   * a work-group computes the SAME tile (R, c) of every layer (the layers of a net with L == W have the same tiling), with
//     gemm.h's tile body as it is - only its A operand, which other work-groups of this launch wrote, is loaded with sc1
//     (served by the L2, not by the CU's L1, where a line of an earlier step may sit);
//   * behind each layer: every thread's stores are acknowledged by the L2 (s_waitcnt vmcnt(0) + barrier), one thread adds 1 to
//     the row tile's counter with a work-group-scope atomic (executed in that L2) and polls it with sc1 loads until the row
//     tile's column tiles have all signed this phase - a barrier among the row tile's work-groups only.
// All work-groups are resident at once (two per CU at 5429 rows); every wait is a bounded spin, a timeout raises an abort flag
// that every work-group polls and the host reads at sdrm_sample_end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.h"

namespace sdrm {

struct SamplePersistArgs {
  GemmArgs l0, lh, lo;          // layer 0, the shared hidden layer (A / C of its first application), the out layer (+ the reverse update)
  size_t pre_stride;            // between the activation buffers of consecutive layers
  const float* B0tab; int ldtab;   // [T + 1][ldtab]: layer 0's bias row of step i
  const float* rev;             // [3][T + 1]: c1, sqrt(alpha), sqrt(beta) per step
  int T, H, i_start;
  int row_tiles, tiles_n;
  unsigned* cnt;                // [row_tiles][32]: phases signed per row tile (one counter per 128-byte line)
  unsigned* abort_;
};

__device__ __forceinline__ unsigned sp_poll(const unsigned* p) {   // the counter as the XCD's L2 holds it
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <class Cfg>
__global__ __launch_bounds__(NTHREADS, 2) void k_sample_persist(const SamplePersistArgs P) {
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = slot / P.tiles_n;
  const int R = x + 8 * rt, c = slot - rt * P.tiles_n;
  if (R >= P.row_tiles) return;
  const int logical = R * P.tiles_n + c;
  const int tid = threadIdx.x;
  unsigned* my = P.cnt + 32 * (size_t)R;
  unsigned phase = 0;
  __shared__ int go;
  // a barrier among the work-groups of this row tile, behind a phase whose output the next phase reads
  auto row_tile_barrier = [&]() -> bool {
    ++phase;
    __syncthreads();   // (s_waitcnt vmcnt(0) in front of it: every thread's stores of this phase are acknowledged)
    if (tid == 0) {
      __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned target = (unsigned)P.tiles_n * phase;
      int ok = 0;
      for (unsigned spins = 0; spins < (1u << 20); ++spins) {
        if (sp_poll(my) >= target) { ok = 1; break; }
        if ((spins & 1023u) == 1023u && __hip_atomic_load(P.abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      if (!ok) __hip_atomic_store(P.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      go = ok;
    }
    __syncthreads();
    return go != 0;
  };
  const int n1 = P.T + 1;
  for (int i = P.i_start; i >= 1; --i) {
    {
      GemmArgs a = P.l0;
      a.bias = P.B0tab + (size_t)i * P.ldtab;
      gemm_body<Cfg, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_BIAS_PRELU, 1>(a, 0, logical);
    }
    if (!row_tile_barrier()) return;
    for (int h = 0; h < P.H; ++h) {
      GemmArgs a = P.lh;
      a.A = P.lh.A + (size_t)h * P.pre_stride;
      a.C = P.lh.C + (size_t)h * P.pre_stride;
      gemm_body<Cfg, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_BIAS_PRELU, 1>(a, 0, logical);
      if (!row_tile_barrier()) return;
    }
    {
      GemmArgs a = P.lo;
      a.rev_step = i;
      a.rev_c1 = P.rev[i]; a.rev_sqrt_alpha = P.rev[n1 + i]; a.rev_sqrt_beta = P.rev[2 * n1 + i];
      gemm_body<Cfg, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_TANH_REV, 1>(a, 0, logical);
    }
    if (!row_tile_barrier()) return;
  }
}

}  // namespace sdrm
