// Reverse-sampling steps of a mid-width eps-net without kernel boundaries between the layers (sample_ddpm, train_SDRM.py:50-59;
// denoise_add_noise, :20-25), for the row counts of a multi-GPU shard: n <= ~1400 rows is a chain of three dependent launches of
// 4 - 6 us per reverse step for 0.16 - 0.3 GFLOP - each launch at the floor of a kernel boundary (~4.5 us on this chip).
//
// ONE launch runs `count` reverse steps: every step's H + 2 forward GEMMs on the 32x32x32 tile (gemm.h's tile body as it is) and
// the DDPM reverse update inside the out layer's epilogue (EPI_TANH_REV: the sampler state X and the next step's dropped-out input
// in place).  What couples consecutive layers is narrow - layer k + 1 of row tile R needs the column tiles of row tile R of layer
// k, nothing else - and the chip's topology makes that hand-over cheap (round 4: tools/rowsync_probe.hip, 0.84 us; round 5: the
// same mechanism as csrc/rows48.h):
//   * row tile R lives on XCD R % 8 with all its column tiles (work-group b runs on XCD b & 7 - checked by sdrm_create; slot
//     b / 8 of that XCD = (row tile index on the XCD, column tile)), so nothing has to be visible beyond that XCD's L2;
//   * a work-group computes the SAME tile (R, c) of every layer (a net with L == W has one tiling for all layers); only its A
//     operand, which other work-groups of this launch wrote, is loaded with sc1 (served by the L2, never by the CU's L1);
//   * behind each layer: the work-group's stores are acknowledged by the L2 (s_waitcnt vmcnt(0)), one thread bumps the row tile's
//     counter with a work-group-scope atomic (executed in that L2) and polls it with sc1 loads until all column tiles of the row
//     tile have signed this phase - a barrier among the row tile's work-groups only.
// Every work-group of the launch is resident at once (the host checks tiles <= 2 per CU), every wait is bounded by the wall
// clock, a timeout raises the handle's host-visible abort word (reported by sdrm_sample_end).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.h"
#include "rows48.h"

namespace sdrm {

struct SamplePersistArgs {
  GemmArgs l0, lh, lo;          // layer 0, the shared hidden layer (A / C of its first application), the out layer (+ the reverse update)
  size_t pre_stride;            // between the activation buffers of consecutive layers
  const float* B0tab; int ldtab;   // [T + 1][ldtab]: layer 0's bias row of step i
  const float* rev;             // [3][T + 1]: c1, sqrt(alpha), sqrt(beta) per step
  int T, H, i_first, count;     // steps i_first, i_first - 1, .. (count of them, not below 1)
  int row_tiles, tiles_n;
  unsigned* cnt;                // [row_tiles][32]: phases signed per row tile (one counter per 128-byte line)
  unsigned base;                // what the counters stood at when this launch was issued
  unsigned* abort_;
};

template <class Cfg>
__global__ __launch_bounds__(NTHREADS, 2) void k_sample_persist(const SamplePersistArgs P) {
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = slot / P.tiles_n;
  const int R = x + 8 * rt, c = slot - rt * P.tiles_n;
  if (R >= P.row_tiles) return;
  const int logical = R * P.tiles_n + c;
  unsigned* my = P.cnt + 32 * (size_t)R;
  unsigned phase = 0;
  __shared__ int go;
  const int n1 = P.T + 1;
  const int i_last = max(P.i_first - P.count + 1, 1);
  for (int i = P.i_first; i >= i_last; --i) {
    {
      GemmArgs a = P.l0;
      a.bias = P.B0tab + (size_t)i * P.ldtab;
      gemm_body<Cfg, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_BIAS_PRELU, BUF_SC1>(a, 0, logical);
    }
    if (!r48_group_barrier(my, P.base + (unsigned)P.tiles_n * ++phase, P.abort_, &go)) return;
    for (int h = 0; h < P.H; ++h) {
      GemmArgs a = P.lh;
      a.A = P.lh.A + (size_t)h * P.pre_stride;
      a.C = P.lh.C + (size_t)h * P.pre_stride;
      gemm_body<Cfg, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_BIAS_PRELU, BUF_SC1>(a, 0, logical);
      if (!r48_group_barrier(my, P.base + (unsigned)P.tiles_n * ++phase, P.abort_, &go)) return;
    }
    {
      GemmArgs a = P.lo;
      a.rev_step = i;
      a.rev_c1 = P.rev[i]; a.rev_sqrt_alpha = P.rev[n1 + i]; a.rev_sqrt_beta = P.rev[2 * n1 + i];
      gemm_body<Cfg, LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_TANH_REV, BUF_SC1>(a, 0, logical);
    }
    // (the last step of the launch needs no hand-shake: the kernel boundary follows - but the counters must stay in step)
    if (!r48_group_barrier(my, P.base + (unsigned)P.tiles_n * ++phase, P.abort_, &go)) return;
  }
}

}  // namespace sdrm
