// EXPERIMENT (round 4, not adopted; kept as a record): the narrow nets' BACKWARD on 4-row MFMA units, 8 users per work-group.
// It was wired in behind csrc/skinny_fwd4.h (same SkStepArgs, slab sets = work-groups, SK_SLAB_SETS = 256 in sdrm_create) and is
// parity-green on the whole GPU suite incl. the random-shape test; at ADM (B = 850) it runs 20.5 us against 22.0 us for
// k_skinny_bwd (16 users per work-group), but its 108 slab sets cost k_tail 1.6 us (5.8 -> 7.4): no net gain.
// Cycles of one work-group (s_memtime stamps): weights 3.8 k, fold 3.3 k, requests 4.8 k, seeds 3.2 k, chain 6.9 k (1.15 k per
// layer), slope reduce + barrier 2.1 k, dpre0 sum + bias sums 1.6 k, weight-gradient tiles 8.5 k + 3.1 k of barrier wait
// (63 tile products of 6 MFMAs on six waves over four SIMDs), M 1.9 k.
// To build it: append to csrc/skinny_fwd4.h inside namespace sdrm.
// ================================================================================================ backward on 4-row units
// k_skinny_bwd (skinny_step.h) with the forward's ownership: a WAVE owns 4 stacked rows - one pass of 4 users - through the whole
// dgrad chain, no barrier between layers; a work-group is 8 USERS (6 waves, 24 rows).  Layer numbering here: layer k = 0 .. H
// produces pre[k] (0: dnn.0, 1 .. H: the shared hidden layer), layer H + 1 is the out layer.  Two sets of [24][SK4_XS] LDS tiles:
//   DT(k), k = 0 .. H + 1: the gradient of layer k's output (DT(H + 1) = the loss seeds times tanh', DT(k) = dpre[k]);
//   XK(k), k = 0 .. H + 1: layer k's input as the forward stored it: XK(0) = U (dropped-out latents), XK(k + 1) = pre[k], RAW -
//     PReLU is applied where it is read (the chain needs the raw value: PReLU' and the slope gradient).
// Chain: DT(k) = (DT(k + 1) * W_{k+1}) .* PReLU'(pre[k]) - srcA = one row of the [out][in] weight copy (lane = input column: no
// transposed image needed), srcB = the wave's 4 gradients of that output column.  Behind ONE barrier every weight gradient is
// dW_k = DT(k)^T * act(XK(k)) over the work-group's 24 rows (16x16x4 tiles, the shared layer's H applications summed in
// registers), bias gradients are column sums of the DT tiles, M (tail.h) comes from dpre0 summed over the passes and the users'
// time-embedding rows.  One slab set per work-group, as k_skinny_bwd leaves them.
constexpr int SK4B_USERS = 8, SK4B_WAVES = 6, SK4B_THREADS = 64 * SK4B_WAVES, SK4B_ROWS = 24, SK4B_MAXH = 6;

__host__ __device__ inline size_t sk4_bwd_lds_floats(int H, int TPs) {
  return (size_t)2 * (H + 2) * SK4B_ROWS * SK4_XS + 8 * SK4_XS + 8 * (size_t)(TPs + 4) + 16 + 16;
}

// one 16 x 16 tile of a weight gradient over ROWS stacked rows: dW[n0 + .][k0 + .] = sum_r D[r][n0 + .] * f(X[r][k0 + .]), f = PReLU
// (act) or the identity; C layout: acc[r] = dW[n0 + 4 lq + r][k0 + li]
template <int ROWS>
__device__ __forceinline__ f32x4 sk4_wgrad_tile(const float* __restrict__ D, int ldd, int n0, const float* __restrict__ X, int ldx, int k0,
                                                bool act, float sl, int li, int lq, f32x4 acc) {
  float dv[ROWS / 4], xv[ROWS / 4];
#pragma unroll
  for (int g = 0; g < ROWS / 4; ++g) { dv[g] = D[(4 * g + lq) * ldd + n0 + li]; xv[g] = X[(4 * g + lq) * ldx + k0 + li]; }
  if (act) {
#pragma unroll
    for (int g = 0; g < ROWS / 4; ++g) xv[g] = prelu_f(xv[g], sl);
  }
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < ROWS / 4; g += 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[g], xv[g], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[g + 1], xv[g + 1], acc1, 0, 0, 0);
  }
  return acc + acc1;
}

template <int NL, int NW>
__global__ __launch_bounds__(SK4B_THREADS) void k_skinny_bwd4(const SkStepArgs a) {
  constexpr int LPk = 16 * NL, WPk = 16 * NW, XS = SK4_XS, R = SK4B_ROWS, NTHR = SK4B_THREADS;
  extern __shared__ __attribute__((aligned(16))) float sk4b[];
  const int nt = a.H + 2;
  float* DTb = sk4b;                            // [nt][R][XS]
  float* XKb = DTb + (size_t)nt * R * XS;       // [nt][R][XS]
  float* D3 = XKb + (size_t)nt * R * XS;        // [8][XS]: dpre0 summed over the passes
  float* Te = D3 + 8 * XS;                      // [8][TPs + 4]: temb rows of the group's users
  float* red = Te + 8 * (a.TPs + 4);            // [6 waves][2]: slope partials (layer 0, hidden total)
  double* shd = reinterpret_cast<double*>(red + 16);   // [16] (8-byte aligned: every piece above is a multiple of 2 floats)
  const int ldte = a.TPs + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pass = wave >> 1, unit = wave & 1;
  const int li = lane & 15, lq = lane >> 4;
  const int b4 = lane >> 2, l4 = lane & 3, c4 = 4 * b4;
  const bool wcols = c4 < WPk, lcols = c4 < LPk;
  const int cw = wcols ? c4 : 0, cl = lcols ? c4 : 0;
  const int wlw = lane < WPk ? lane : 0;
  const int r = 4 * wave + l4;                  // this lane's row of every tile
  const int s = blockIdx.x;                     // this work-group's slab set
  auto DT = [&](int k) __attribute__((always_inline)) { return DTb + (size_t)k * R * XS; };
  auto XK = [&](int k) __attribute__((always_inline)) { return XKb + (size_t)k * R * XS; };

  // first requests of the kernel (their values are waited for in issue order): the loss partials of the fold, the timesteps of the
  // first group's users - then the srcA operands of the dgrads: rows of the [out][in] copies, lane = input column
  double fv[4] = {0, 0, 0, 0};
  if (!a.sums) {
    const int i0 = min(64 * (wave & 3) + lane, a.NP - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) fv[j] = a.loss_part[4 * (size_t)i0 + j];
  }
  int tu_next = a.tdev[min(SK4B_USERS * (int)blockIdx.x + ((tid >> 5) & 7), a.B - 1)];
  float wor[LPk], whr[WPk];
  {
    const float* po = a.Woc + wlw;
    const float* ph = (a.H > 0 ? a.Whc : a.Woc) + wlw;   // (H == 0: never used, any valid rows)
#pragma unroll
    for (int o = 0; o < LPk; ++o) { wor[o] = *po; po += a.WPs; }
#pragma unroll
    for (int o = 0; o < WPk; ++o) { whr[o] = *ph; ph += a.WPs; }
  }
  const float slope0 = *a.slope0, slopeh = a.H > 0 ? *a.slopeh : 0.f;

  // ---- the five sums: given (sharded step, after the all-reduce) or folded from the forward's partials (k_skinny_bwd's fold:
  // the summation tree of k_loss_sums)
  double s0, s1, s2, s3, N;
  if (a.sums) {
    s0 = a.sums[0]; s1 = a.sums[1]; s2 = a.sums[2]; s3 = a.sums[3]; N = a.sums[4];
  } else {
    // (wave vw < 4 plays k_loss_sums' wave vw; the four results are added in that kernel's order)
    if (wave < 4 && 64 * wave < a.NP) {
      double v[4];
      const int i0 = 64 * wave + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i0 < a.NP ? fv[j] : 0.0;
      for (int i = i0 + 256; i < a.NP; i += 256)
        for (int j = 0; j < 4; ++j) v[j] += a.loss_part[4 * (size_t)i + j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += __shfl_down(v[j], off, 64);
      if (lane == 0)
        for (int j = 0; j < 4; ++j) shd[4 * wave + j] = v[j];
    }
    lds_barrier();
    {
      double tot[4] = {0, 0, 0, 0};
      for (int vw = 0; vw < 4 && 64 * vw < a.NP; ++vw)
        for (int j = 0; j < 4; ++j) tot[j] += shd[4 * vw + j];
      s0 = tot[0]; s1 = tot[1]; s2 = tot[2]; s3 = tot[3]; N = a.count;
    }
  }
  const double A = s0 / N, Cc = s1 / N, Rbar = s2 / N;
  const double V = (N > 1.0) ? (s3 - N * Rbar * Rbar) / (N - 1.0) : __builtin_nan("");
  const double den = 1e-8 + V;
  const double kk = 0.5 / den;
  const float cD = (float)(2.0 * kk / N);
  const float cV = (float)(-(0.5 * (A + Cc) / (den * den)) * 2.0 / (N - 1.0));
  const float rbar = (float)Rbar;
  if (s == 0 && tid == 0 && a.loss) *a.loss = (float)(0.5 * (A + Cc) / den);
  const int G8 = 2 * a.G;   // groups of 8 users: every row of every 16-user group
  bool first = true;
  for (int g = blockIdx.x; g < G8; g += gridDim.x, first = false) {
    const int u0 = SK4B_USERS * g;
    const int usr = u0 + 4 * unit + l4;
    const bool uok = usr < a.B;
    const size_t gbase = (size_t)SK_ROWS * (u0 / SK_USERS) + (u0 % SK_USERS) + 4 * unit + l4;   // + 16 pass: this lane's stacked row
    const size_t myrow = gbase + 16 * pass;
    lds_barrier();   // the previous group is done with every tile
    // ---- requests first, unconditional at clamped indices: the forward's outputs and x0 (seeds), every layer's pre-activations,
    // layer 0's input, the users' timesteps
    const int tu = tu_next;   // thread -> (user (tid / 32) % 8, quad tid % 32) of the temb rows: requested a group ahead (another read's
                              // address waits for it)
    tu_next = a.tdev[min(u0 + SK4B_USERS * (int)gridDim.x + ((tid >> 5) & 7), a.B - 1)];
    f32x4 yq[3];
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) yq[pp] = *reinterpret_cast<const f32x4*>(a.Y + (gbase + 16 * pp) * a.LPs + cl);
    float xv[4];
    {
      const size_t xb = (size_t)min(usr, a.B - 1) * a.L;
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[e] = a.x0[xb + min(c4 + e, a.L - 1)];
    }
    f32x4 pq[SK4B_MAXH + 1];
#pragma unroll
    for (int k = 0; k <= SK4B_MAXH; ++k) pq[k] = *reinterpret_cast<const f32x4*>(a.pre + (size_t)min(k, a.H) * a.pre_stride + myrow * a.WPs + cw);
    const f32x4 uq = *reinterpret_cast<const f32x4*>(a.U + myrow * a.K0 + cl);
    const f32x4 tev = *reinterpret_cast<const f32x4*>(a.tembP + (size_t)tu * a.TPs + 4 * min(tid & 31, a.TPs / 4 - 1));   // (used behind the chain)
    // ---- seeds (App. A.5) times tanh' -> DT(H + 1); the layer inputs -> XK
    {
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (uok && c4 + e < a.L) {
          const float P = yq[0][e], S = yq[1][e], Q = yq[2][e];
          const float Rr = P - xv[e];
          constexpr float IMU2 = 1.f / MU2;   // (a multiply, as in the wide path's seeds: an IEEE division is ten instructions)
          const float D = (Q - S) * IMU2 - Rr;
          const float gD = cD * D;
          const float gC = cD * (Rr - S);
          const float gV = cV * (Rr - rbar);
          const float gP = (-gD + gC + gV) * (1.f - P * P);
          const float gQ = (gD * IMU2) * (1.f - Q * Q);
          const float gS = (-gD * IMU2 - gC) * (1.f - S * S);
          dq[e] = pass == 0 ? gP : (pass == 1 ? gS : gQ);
        }
      }
      if (lcols) {
        *reinterpret_cast<f32x4*>(DT(a.H + 1) + r * XS + c4) = dq;
        *reinterpret_cast<f32x4*>(XK(0) + r * XS + c4) = uq;
      }
      if (wcols) {
#pragma unroll
        for (int k = 0; k <= SK4B_MAXH; ++k)
          if (k <= a.H) *reinterpret_cast<f32x4*>(XK(k + 1) + r * XS + c4) = pq[k];
      }
    }
    if (tid < 256) {
      const int ur = tid >> 5, q = tid & 31;
      if (4 * q < a.TPs) *reinterpret_cast<f32x4*>(Te + ur * ldte + 4 * q) = u0 + ur < a.B ? tev : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- the chain, this wave's four rows, no barrier: DT(k) = (DT(k + 1) * W_{k+1}) .* PReLU'(pre[k])
    float ss0 = 0.f, ssh = 0.f;
    for (int k = a.H; k >= 0; --k) {
      const float* in = DT(k + 1) + 4 * wave * XS;
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(XK(k + 1) + r * XS + cw);   // raw pre[k]: requested in front of the layer's MFMAs
      const f32x4 acc = k == a.H ? sk4_layer<LPk / 4>(wor, in, lane) : sk4_layer<WPk / 4>(whr, in, lane);
      float ss = 0.f;
      if (wcols) {
        const float sl = k > 0 ? slopeh : slope0;
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = p4[e], v = acc[e];
          const bool pos = p > 0.f;
          d[e] = pos ? v : sl * v;
          ss += pos ? 0.f : v * p;
        }
        *reinterpret_cast<f32x4*>(DT(k) + r * XS + c4) = d;
      }
      if (k > 0) ssh += ss; else ss0 = ss;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ss0 += __shfl_down(ss0, off, 64); ssh += __shfl_down(ssh, off, 64); }
    if (lane == 0) { red[2 * wave] = ss0; red[2 * wave + 1] = ssh; }
    lds_barrier();
    // ---- behind the one barrier: dpre0 summed over the passes, bias gradients, every weight gradient of the 24 rows
    for (int f = tid; f < SK4B_USERS * WPk; f += NTHR) {
      const int ur = f / WPk, c = f - ur * WPk;
      D3[ur * XS + c] = (DT(0)[ur * XS + c] + DT(0)[(8 + ur) * XS + c]) + DT(0)[(16 + ur) * XS + c];
    }
    if (tid < 3 * 64) {
      const int kind = tid >> 6, c = tid & 63;   // 0: out layer, 1: the shared hidden layer (all applications), 2: layer 0
      float t = 0.f;
      if (kind == 0) {
        if (c < LPk) for (int q = 0; q < R; ++q) t += DT(a.H + 1)[q * XS + c];
      } else if (kind == 1) {
        if (c < WPk) for (int k = a.H; k >= 1; --k) for (int q = 0; q < R; ++q) t += DT(k)[q * XS + c];
      } else {
        if (c < WPk) for (int q = 0; q < R; ++q) t += DT(0)[q * XS + c];
      }
      float* dst = kind == 0 ? (c < a.LPs ? a.dbOs + (size_t)s * a.LPs + c : nullptr)
                             : (c < a.WPs ? (kind == 1 ? a.dbHs : a.db0s) + (size_t)s * a.WPs + c : nullptr);
      if (dst && (kind != 1 || a.H >= 1)) *dst = first ? t : *dst + t;
    }
    {
      constexpr int NO = NL * NW, NH = NW * NW;
      const float slH = a.H > 0 ? slopeh : slope0;   // the slope of the layer whose output feeds the out layer
      // work units: an out-layer or layer-0 tile is one product, a hidden tile H of them.  The hidden tiles are dealt first, round
      // robin, then the single ones to whoever has the least (dealt by index alone a wave gets up to 13 units at ADM, average 10.5)
      auto single = [&](int t1) __attribute__((always_inline)) {
        if (t1 < NO) {   // out layer: dWo[n < L][k < W]
          const int nt_ = t1 / NW, kt = t1 - nt_ * NW;
          const f32x4 acc = sk4_wgrad_tile<R>(DT(a.H + 1), XS, 16 * nt_, XK(a.H + 1), XS, 16 * kt, true, slH, li, lq, f32x4{0.f, 0.f, 0.f, 0.f});
          sk_slab_tile(a.slabO + (size_t)s * a.LPs * a.WPs, a.WPs, 16 * nt_, 16 * kt, li, lq, acc, first);
        } else {         // layer 0: dW0[n < W][k < L], input U as it is
          const int t2 = t1 - NO, nt_ = t2 / NL, kt = t2 - nt_ * NL;
          const f32x4 acc = sk4_wgrad_tile<R>(DT(0), XS, 16 * nt_, XK(0), XS, 16 * kt, false, 0.f, li, lq, f32x4{0.f, 0.f, 0.f, 0.f});
          sk_slab_tile(a.slab0 + (size_t)s * a.WPs * a.K0, a.K0, 16 * nt_, 16 * kt, li, lq, acc, first);
        }
      };
      int units = 0;
      if (a.H >= 1) {   // the shared hidden layer: its H applications summed (Q1)
        for (int t2 = wave; t2 < NH; t2 += SK4B_WAVES, units += a.H) {
          const int nt_ = t2 / NW, kt = t2 - nt_ * NW;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          for (int k = a.H; k >= 1; --k)
            acc = sk4_wgrad_tile<R>(DT(k), XS, 16 * nt_, XK(k), XS, 16 * kt, true, k - 1 > 0 ? slopeh : slope0, li, lq, acc);
          sk_slab_tile(a.slabH + (size_t)s * a.WPs * a.WPs, a.WPs, 16 * nt_, 16 * kt, li, lq, acc, first);
        }
      }
      // single tiles: first the waves with one hidden tile fewer catch up (H singles each), then everybody in turn
      {
        const int extra = a.H >= 1 ? NH % SK4B_WAVES : 0, nlight = SK4B_WAVES - extra;
        const int nA = min(nlight * (extra ? a.H : 0), 2 * NO);
        if (wave >= extra)
          for (int t1 = wave - extra; t1 < nA; t1 += nlight) single(t1);
        for (int t1 = nA + wave; t1 < 2 * NO; t1 += SK4B_WAVES) single(t1);
      }
    }
    lds_barrier();   // D3 (and Te) complete
    // ---- M[n < W][i < T] = sum over the group's users of D3[u][n] * temb[t_u][i], into the trailing columns of the layer-0 slab
    {
      const int TT = a.TPs / 16;
      for (int tl = wave; tl < NW * TT; tl += SK4B_WAVES) {
        const int nt_ = tl / TT, it = tl - nt_ * TT;
        const f32x4 acc = sk4_wgrad_tile<SK4B_USERS>(D3, XS, 16 * nt_, Te, ldte, 16 * it, false, 0.f, li, lq, f32x4{0.f, 0.f, 0.f, 0.f});
        sk_slab_tile(a.slab0 + (size_t)s * a.WPs * a.K0, a.K0, 16 * nt_, a.LPs + 16 * it, li, lq, acc, first);
      }
    }
    if (tid <= a.H) {   // application slots of alpha_part: [0] layer 0, [1] the hidden layer's total, [2 ..] nothing
      float sum = 0.f;
      if (tid < 2)
        for (int w = 0; w < SK4B_WAVES; ++w) sum += red[2 * w + tid];
      float* dst = a.alpha_part + (size_t)tid * a.alpha_part_stride + s;
      if (first) *dst = sum; else *dst += sum;
    }
  }
}

}  // namespace sdrm
