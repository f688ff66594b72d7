// Calibration probe for DESIGN.md section 8, item 2: a sampling step's layers as phases of ONE persistent launch,
// synchronised per ROW TILE instead of per grid.  510 work-groups = 85 row tiles x 6 column tiles (the sampling GEMM's
// grid at 5429 rows); the six work-groups of a row tile sit on one XCD (the engine's xcd_remap).  Per phase a work-group
// writes its 64 x 64 tile (16 KB), signals the row tile's counter, waits until all six have signalled, then reads the six
// tiles of its row (96 KB, what the next layer's A operand is) and checks every value.  Against: the same data
// movement as one launch per phase.  Every wait is a bounded spin; a timeout aborts all work-groups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/rowsync_probe tools/rowsync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROWS = 85, COLS = 6, NB = ROWS * COLS, TILE = 64 * 64;   // floats per tile

__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}
__device__ __forceinline__ float tile_value(int phase, int R, int c, int i) { return (float)(phase * 7 + R * 3 + c) + (float)(i & 15) * 0.0625f; }

// mode 0: handshake only.  mode 1: data through agent-scope (L2-level, sc1) dword stores and loads, relaxed counters.
// mode 2: plain 16-byte stores and loads, agent-scope release fence before the signal, acquire fence after the wait.
__global__ __launch_bounds__(256) void k_rowsync(float* buf, unsigned int* cnt, unsigned int* abort_, int phases, int mode, int* bad) {
  const int logical = xcd_remap(blockIdx.x, NB);
  const int R = logical / COLS, c = logical - R * COLS;
  const int tid = threadIdx.x;
  int errors = 0;
  __shared__ int go;
  for (int p = 0; p < phases; ++p) {
    float* mine = buf + ((size_t)(p & 1) * NB + (size_t)R * COLS + c) * TILE;
    if (mode == 1) {
      for (int i = tid; i < TILE; i += 256) __hip_atomic_store(mine + i, tile_value(p, R, c, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 2) {
      for (int i = 4 * tid; i < TILE; i += 1024)
        *reinterpret_cast<float4*>(mine + i) = make_float4(tile_value(p, R, c, i), tile_value(p, R, c, i + 1), tile_value(p, R, c, i + 2), tile_value(p, R, c, i + 3));
    }
    __syncthreads();   // every thread's stores are acknowledged (s_waitcnt vmcnt(0) before the barrier)
    if (tid == 0) {
      if (mode == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(cnt + R, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = (unsigned int)COLS * (unsigned int)(p + 1);
      unsigned int spins = 0;
      int ok = 1;
      while (__hip_atomic_load(cnt + R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1u << 20) || __hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      go = ok;
    }
    __syncthreads();
    if (!go) break;   // uniform: every thread of the work-group leaves
    if (mode != 0) {
      const float* row = buf + ((size_t)(p & 1) * NB + (size_t)R * COLS) * TILE;
      if (mode == 1) {
        for (int i = tid; i < COLS * TILE; i += 256) {
          const float v = __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          errors += v != tile_value(p, R, i / TILE, i % TILE);
        }
      } else {
        for (int i = 4 * tid; i < COLS * TILE; i += 1024) {
          const float4 v = *reinterpret_cast<const float4*>(row + i);
          const int cc = i / TILE, ii = i % TILE;
          errors += (v.x != tile_value(p, R, cc, ii)) + (v.y != tile_value(p, R, cc, ii + 1)) + (v.z != tile_value(p, R, cc, ii + 2)) +
                    (v.w != tile_value(p, R, cc, ii + 3));
        }
      }
    }
  }
  if (errors) atomicAdd(bad, errors);
}

// Round 4: the same hand-over THROUGH ONE XCD's L2.  Row tile R lives on XCD R % 8 - all six of its work-groups (work-group b runs on
// XCD b & 7; slot b / 8 of that XCD = (row tile index on the XCD, column tile)) - so nothing has to be coherent beyond that L2:
// the counter is incremented by work-group-scope atomics (executed in the L2), polled by an sc1 load; the tiles are written by
// plain stores (acknowledged by the L2 before the signal) and read by sc1 loads (not from the CU's L1, where a line of an
// earlier phase may sit), eight 16-byte loads in flight per thread.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned poll_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void load_sc1x8(const float* p, f32x4v (&v)[8]) {   // p, p + 4 KiB, ..
  const char* q = reinterpret_cast<const char*>(p);
  const char *q1 = q + 4096, *q2 = q + 8192, *q3 = q + 12288, *q4 = q + 16384, *q5 = q + 20480, *q6 = q + 24576, *q7 = q + 28672;
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(q), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7)
      : "memory");
}
constexpr int SLOTS = ((ROWS + 7) / 8) * COLS;   // work-groups per XCD (the XCDs with a row tile fewer leave six idle)

__global__ __launch_bounds__(256) void k_rowsync_l2(float* buf, unsigned int* cnt, unsigned int* abort_, int phases, int data, int* bad) {
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int R = x + 8 * (slot / COLS), c = slot % COLS;
  if (R >= ROWS) return;
  const int tid = threadIdx.x;
  int errors = 0;
  __shared__ int go;
  for (int p = 0; p < phases; ++p) {
    float* mine = buf + ((size_t)(p & 1) * NB + (size_t)R * COLS + c) * TILE;
    if (data)
      for (int i = 4 * tid; i < TILE; i += 1024)
        *reinterpret_cast<float4*>(mine + i) = make_float4(tile_value(p, R, c, i), tile_value(p, R, c, i + 1), tile_value(p, R, c, i + 2), tile_value(p, R, c, i + 3));
    __syncthreads();   // every thread's stores are acknowledged (s_waitcnt vmcnt(0) before the barrier)
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt + 32 * R, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned int target = (unsigned int)COLS * (unsigned int)(p + 1);
      int ok = 0;
      for (unsigned int spins = 0; spins < (1u << 18); ++spins) {
        if (poll_sc1(cnt + 32 * R) >= target) { ok = 1; break; }
        if ((spins & 255u) == 255u && __hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      if (!ok) __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      go = ok;
    }
    __syncthreads();
    if (!go) break;
    if (data) {
      const float* row = buf + ((size_t)(p & 1) * NB + (size_t)R * COLS) * TILE;
      for (int i = 4 * tid; i < COLS * TILE; i += 8 * 1024) {   // COLS * TILE = 24576 floats = 3 x 8192
        f32x4v v[8];
        load_sc1x8(row + i, v);
        for (int u = 0; u < 8; ++u) {
          const int ii = i + 1024 * u, cc = ii / TILE, jj = ii % TILE;
          for (int e = 0; e < 4; ++e) errors += v[u][e] != tile_value(p, R, cc, jj + e);
        }
      }
    }
  }
  if (errors) atomicAdd(bad, errors);
}

// the same data movement as one launch per phase: write the tile of phase p, check the row's tiles of phase p - 1
__global__ __launch_bounds__(256) void k_phase(float* buf, int p, int* bad) {
  const int logical = xcd_remap(blockIdx.x, NB);
  const int R = logical / COLS, c = logical - R * COLS;
  const int tid = threadIdx.x;
  int errors = 0;
  if (p > 0) {
    const float* row = buf + ((size_t)((p - 1) & 1) * NB + (size_t)R * COLS) * TILE;
    for (int i = 4 * tid; i < COLS * TILE; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      const int cc = i / TILE, ii = i % TILE;
      errors += (v.x != tile_value(p - 1, R, cc, ii)) + (v.y != tile_value(p - 1, R, cc, ii + 1)) + (v.z != tile_value(p - 1, R, cc, ii + 2)) +
                (v.w != tile_value(p - 1, R, cc, ii + 3));
    }
  }
  float* mine = buf + ((size_t)(p & 1) * NB + (size_t)R * COLS + c) * TILE;
  for (int i = 4 * tid; i < TILE; i += 1024)
    *reinterpret_cast<float4*>(mine + i) = make_float4(tile_value(p, R, c, i), tile_value(p, R, c, i + 1), tile_value(p, R, c, i + 2), tile_value(p, R, c, i + 3));
  if (errors) atomicAdd(bad, errors);
}

int main() {
  float* buf; unsigned int *cnt, *abort_; int* bad;
  hipMalloc(&buf, (size_t)2 * NB * TILE * 4); hipMalloc(&cnt, ROWS * 4); hipMalloc(&abort_, 4); hipMalloc(&bad, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int phases = 300;
  const char* names[3] = {"handshake only", "agent-scope dword stores / loads (sc1), relaxed counter", "plain 16-byte stores / loads, release + acquire fences"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    int hbad = 0; unsigned int habort = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(cnt, 0, ROWS * 4); hipMemset(abort_, 0, 4); hipMemset(bad, 0, 4);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_rowsync, dim3(NB), dim3(256), 0, 0, buf, cnt, abort_, phases, mode, bad);
      hipEventRecord(e1, 0);
      hipError_t rc = hipDeviceSynchronize();
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&habort, abort_, 4, hipMemcpyDeviceToHost);
      if (rc != hipSuccess || habort) { printf("mode %d: rc %s abort %u\n", mode, hipGetErrorString(rc), habort); break; }
    }
    printf("persistent, row-tile sync, %-58s: %6.2f us per phase   wrong values %d   timed out %u\n", names[mode], best * 1e3 / phases, hbad, habort);
  }
  {
    unsigned int* cnt2; hipMalloc(&cnt2, ROWS * 32 * 4);
    const char* n2[2] = {"handshake only, through the row tile's XCD's L2 (round 4)", "plain stores / sc1 loads through that L2, work-group-scope counter"};
    for (int data = 0; data < 2; ++data) {
      float best = 1e30f;
      int hbad = 0; unsigned int habort = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(cnt2, 0, ROWS * 32 * 4); hipMemset(abort_, 0, 4); hipMemset(bad, 0, 4);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rowsync_l2, dim3(8 * SLOTS), dim3(256), 0, 0, buf, cnt2, abort_, phases, data, bad);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&habort, abort_, 4, hipMemcpyDeviceToHost);
        if (habort) break;
      }
      printf("persistent, row-tile sync, %-66s: %6.2f us per phase   wrong values %d   timed out %u\n", n2[data], best * 1e3 / phases, hbad, habort);
    }
  }
  {
    float best = 1e30f;
    int hbad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(bad, 0, 4);
      hipEventRecord(e0, 0);
      for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(NB), dim3(256), 0, 0, buf, p, bad);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    }
    printf("one launch per phase (plain 16-byte stores / loads)                                     : %6.2f us per phase   wrong values %d\n",
           best * 1e3 / phases, hbad);
  }
  return 0;
}
