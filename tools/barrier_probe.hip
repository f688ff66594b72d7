// Calibration probe: what a dependent kernel boundary costs on MI355X against a grid-wide barrier inside one
// persistent launch (all 8 XCDs), with a cross-XCD data exchange at every barrier to check visibility.
//   hipcc --offload-arch=gfx950 -O3 -o tools/barrier_probe tools/barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_step(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (i + 4096 * 13) % n;          // read what a far-away block (other XCD) wrote in the previous launch
  out[i] = in[j] + 1.0f;
}

struct Bar { unsigned int count; unsigned int gen; unsigned int abort_; };

// sense-reversing grid barrier with a bounded spin: returns false when it gave up (abort flag set for everyone)
__device__ bool grid_barrier(Bar* b, unsigned int nblocks, unsigned int& local_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();   // release: this block's stores are device-visible
    const unsigned int target = local_gen + 1;
    const unsigned int arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == nblocks) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned int spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
        if (++spins > (1u << 22) || __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __threadfence();   // acquire
    local_gen = target;
  }
  __shared__ int okk;
  if (threadIdx.x == 0) okk = ok ? 1 : 0;
  __syncthreads();
  return okk != 0;
}

// the lean form: no cache-wide fences; the data that crosses the barrier is stored and loaded with agent scope
// (write-through / L2-bypassing accesses), the barrier itself is relaxed agent-scope atomics
__device__ bool grid_barrier_lean(Bar* b, unsigned int nblocks, unsigned int& local_gen) {
  __syncthreads();   // every thread's agent-scope stores have been acknowledged (vmcnt(0) before s_barrier)
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned int target = local_gen + 1;
    const unsigned int arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == nblocks) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned int spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
        if (++spins > (1u << 22) || __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
      }
    }
    local_gen = target;
  }
  __shared__ int okk2;
  if (threadIdx.x == 0) okk2 = ok ? 1 : 0;
  __syncthreads();
  return okk2 != 0;
}

// mode 0: barriers only.  mode 1: + exchange: plain store, then agent-scope (sc1) load of a far block's value
__global__ __launch_bounds__(256) void k_persistent(float* a, float* b, int n, int steps, Bar* bar, int mode, int* bad) {
  unsigned int gen = 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (i + 4096 * 13) % n;
  float* in = a; float* out = b;
  for (int s = 0; s < steps; ++s) {
    if (mode == 1) {
      const float v = __hip_atomic_load(in + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (float)s) atomicAdd(bad, 1);
      out[i] = v + 1.0f;
    } else if (mode == 3) {
      const float v = __hip_atomic_load(in + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (float)s) atomicAdd(bad, 1);
      __hip_atomic_store(out + i, v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (mode >= 2) { if (!grid_barrier_lean(bar, gridDim.x, gen)) return; }
    else if (!grid_barrier(bar, gridDim.x, gen)) return;
    float* t = in; in = out; out = t;
  }
}

int main() {
  const int blocks = 512, n = blocks * 256, steps = 2000;
  float *a, *b; Bar* bar; int* bad;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&bar, sizeof(Bar)); hipMalloc(&bad, 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  // dependent launches
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(a, 0, n * 4);
    hipEventRecord(e0);
    float *in = a, *out = b;
    for (int s = 0; s < steps; ++s) { k_step<<<blocks, 256>>>(in, out, n); float* t = in; in = out; out = t; }
    hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(n); hipMemcpy(h.data(), (steps % 2) ? b : a, n * 4, hipMemcpyDeviceToHost);
    printf("dependent launches: %d x %d blocks: %.2f us per launch (value check %s)\n", steps, blocks, ms * 1e3 / steps,
           h[12345] == (float)steps ? "ok" : "BAD");
  }
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(bar, 0, sizeof(Bar)); hipMemset(bad, 0, 4);
      hipEventRecord(e0);
      k_persistent<<<blocks, 256>>>(a, b, n, steps, bar, mode, bad);
      hipEventRecord(e1);
      hipError_t rc = hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
      Bar hb; int hbad; hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
      printf("persistent, mode %d: %d grid barriers over %d blocks: %.2f us per barrier (rc %d, abort %u, gen %u, stale reads %d)\n",
             mode, steps, blocks, ms * 1e3 / steps, (int)rc, hb.abort_, hb.gen, hbad);
    }
  }
  return 0;
}
