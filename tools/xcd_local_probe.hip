// Calibration probe for DESIGN.md section 8 (round 4, "next" item 4): the one structure for small batches not yet measured - an
// XCD-LOCAL row-owned forward.  Each XCD owns an eighth of the stacked rows through all layers, its 32 CUs split a layer's output
// columns, and between layers the activations are exchanged through THAT XCD's L2 (coherent inside an XCD: no write-back /
// invalidate of the kind a device-scope fence costs on 8 non-coherent L2s, 39 us in tools/barrier_probe.hip).
// 256 work-groups, one per CU; work-group b runs on XCD b & 7 (profiles/r02_wgrad_one_round_and_strips.txt), so the 32 work-groups
// {x, x + 8, ..} share XCD x.  Per phase (= layer) a work-group writes its piece of the activation tile (ROWS x 352 floats per
// XCD: 1/32 of it), signals its XCD's counter, waits for all 32, then reads the WHOLE tile (what the next layer's operand is)
// and checks every value.
//   Columns: the barrier alone; + the stores; + the reads as sc1 loads (served by the XCD's L2: a line read in an earlier phase may
//   sit stale in the CU's L1); + the reads as plain loads; the same data movement with one kernel launch per phase (what the engine
//   does today).  The counter is incremented by work-group-scope atomics (executed in the XCD's L2) and polled by sc1 loads: polled
//   by sc0 loads - or by a fetch_add of 0, which the compiler folds into an atomic load - some work-groups keep seeing a stale
//   count (63 of 64 while others have gone on to 65) and the barrier times out.
// Every wait is a bounded spin; a timeout raises an abort flag all work-groups poll.
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcd_local_probe tools/xcd_local_probe.hip && tools/xcd_local_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int COLS = 352, WG_PER_XCD = 32, NWG = 256;

__device__ __forceinline__ float tile_value(int phase, int xcd, int i) { return (float)(phase * 5 + xcd) + (float)(i & 255) * 0.00390625f; }

// eight global loads with sc1 (served by the XCD's L2, not by this CU's L1), all in flight before the one wait
__device__ __forceinline__ void load_l2x8(const float* p, int stride_bytes, f32x4 (&v)[8]) {
  const char* q = reinterpret_cast<const char*>(p);
  const char* q1 = q + stride_bytes; const char* q2 = q1 + stride_bytes; const char* q3 = q2 + stride_bytes;
  const char* q4 = q3 + stride_bytes; const char* q5 = q4 + stride_bytes; const char* q6 = q5 + stride_bytes; const char* q7 = q6 + stride_bytes;
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(q), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7)
      : "memory");
}

// the counter as the XCD's L2 holds it
__device__ __forceinline__ unsigned poll_l2(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(256) void k_xcd(float* buf, unsigned* cnt, unsigned* abort_, int rows, int phases, int mode, int* bad,
                                              unsigned long long* cyc) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, tid = threadIdx.x;
  const int tile = rows * COLS, piece = tile / WG_PER_XCD;   // floats
  int errors = 0;
  __shared__ int go;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int p = 0; p < phases; ++p) {
    float* t = buf + ((size_t)(p & 1) * 8 + xcd) * tile;
    if (mode & 1)
      for (int i = 4 * tid; i < piece; i += 1024) {
        const int e = idx * piece + i;
        *reinterpret_cast<f32x4*>(t + e) = f32x4{tile_value(p, xcd, e), tile_value(p, xcd, e + 1), tile_value(p, xcd, e + 2), tile_value(p, xcd, e + 3)};
      }
    __syncthreads();   // (s_waitcnt vmcnt(0) in front of the barrier: this work-group's stores are acknowledged by L2)
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt + 32 * xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned target = (unsigned)WG_PER_XCD * (unsigned)(p + 1);
      int ok = 0;
      for (int spin = 0; spin < 200000; ++spin) {
        if (poll_l2(cnt + 32 * xcd) >= target) { ok = 1; break; }
        if ((spin & 255) == 255 && __hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      if (!ok) __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      go = ok;
    }
    __syncthreads();
    if (!go) { if (tid == 0) atomicAdd(bad, 1000000 + p); return; }
    if (mode & 4)
      for (int i = 4 * tid; i < tile; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(t + i);
        for (int e = 0; e < 4; ++e) errors += v[e] != tile_value(p, xcd, i + e);
      }
    if (mode & 2)
      for (int i = 4 * tid; i < tile; i += 8 * 1024) {   // (tile = rows x 352 floats is a multiple of 8192 floats for rows % 256 == 0 only:
        f32x4 v[8];                                       //  the tail reads clamp to the tile's last 16 bytes)
        const int last = tile - 4;
        const float* base = t + min(i, last);
        load_l2x8(base, 4096, v);
        for (int u = 0; u < 8; ++u) {
          const int ii = i + 1024 * u;
          if (ii <= last) for (int e = 0; e < 4; ++e) errors += v[u][e] != tile_value(p, xcd, ii + e);
        }
      }
  }
  if (errors) atomicAdd(bad, errors);
  if (tid == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

// the same data movement, one launch per phase (stores in launch p, loads in launch p + 1: the kernel boundary is the barrier)
__global__ __launch_bounds__(256) void k_phase(float* buf, int rows, int p, int* bad) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, tid = threadIdx.x;
  const int tile = rows * COLS, piece = tile / WG_PER_XCD;
  int errors = 0;
  if (p > 0) {
    const float* t = buf + ((size_t)((p - 1) & 1) * 8 + xcd) * tile;
    for (int i = 4 * tid; i < tile; i += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(t + i);
      for (int e = 0; e < 4; ++e) errors += v[e] != tile_value(p - 1, xcd, i + e);
    }
  }
  float* t = buf + ((size_t)(p & 1) * 8 + xcd) * tile;
  for (int i = 4 * tid; i < piece; i += 1024) {
    const int e = idx * piece + i;
    *reinterpret_cast<f32x4*>(t + e) = f32x4{tile_value(p, xcd, e), tile_value(p, xcd, e + 1), tile_value(p, xcd, e + 2), tile_value(p, xcd, e + 3)};
  }
  if (errors) atomicAdd(bad, errors);
}

int main() {
  const int phases = 40;
  float* buf; unsigned *cnt, *abort_; int* bad; unsigned long long* cyc;
  hipMalloc(&buf, (size_t)2 * 8 * 1024 * COLS * 4); hipMalloc(&cnt, 8 * 32 * 4); hipMalloc(&abort_, 4); hipMalloc(&bad, 4); hipMalloc(&cyc, NWG * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("256 work-groups (32 per XCD), %d phases; us per phase.  rows = stacked rows per XCD (tile = rows x 352 floats)\n", phases);
  for (int rows : {64, 128, 384}) {
    const int modes[5] = {0, 1, 3, 5, -1};   // barrier alone; + stores; + stores and sc1 loads; + stores and plain loads; a launch per phase
    float us[5]; int nbad[5];
    for (int m = 0; m < 5; ++m) {
      for (int rep = 0; rep < 2; ++rep) {   // first repetition: warm-up
        hipMemset(cnt, 0, 8 * 32 * 4); hipMemset(abort_, 0, 4); hipMemset(bad, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (modes[m] >= 0) hipLaunchKernelGGL(k_xcd, dim3(NWG), dim3(256), 0, 0, buf, cnt, abort_, rows, phases, modes[m], bad, cyc);
        else for (int p = 0; p <= phases; ++p) hipLaunchKernelGGL(k_phase, dim3(NWG), dim3(256), 0, 0, buf, rows, p, bad);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        us[m] = ms * 1e3f / phases;
        hipMemcpy(&nbad[m], bad, 4, hipMemcpyDeviceToHost);
      }
    }
    printf("rows %4d (%3d KB per XCD):  barrier alone %6.2f (%d)   + stores %6.2f (%d)   + sc1 loads %6.2f (%d)   + plain loads %6.2f (%d)   a launch per phase %6.2f (%d)\n",
           rows, rows * COLS * 4 / 1024, us[0], nbad[0], us[1], nbad[1], us[2], nbad[2], us[3], nbad[3], us[4], nbad[4]);
  }
  return 0;
}
