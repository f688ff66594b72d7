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
