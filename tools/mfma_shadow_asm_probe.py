#!/usr/bin/env python3
"""What does an instruction cost beside fp32 MFMAs?  ISA-level probe: the K-step of the row-owned kernels in miniature - one wave
per SIMD, 132 v_mfma_f32_16x16x4_f32 on 33 accumulator quads per trip - written as ONE inline-asm block with fixed registers,
so nothing the compiler does (accumulator copies, address arithmetic, waitcnt placement) is in the measurement.  Variants add
global wave-loads (consumed by the MFMAs of the next trip or not), LDS reads, LDS-DMA loads, stores, VALU instructions, s_nop,
and different register banks for the A / B operands.

    python3 tools/mfma_shadow_asm_probe.py            # writes tools/_gen/mfma_shadow_asm_probe.hip and builds the binary beside it
    tools/_gen/mfma_shadow_asm_probe                  # on the GPU: cycles per trip (4224 = the MFMAs alone)

Findings (profiles/r03_mfma_shadow_asm_probe.txt): memory instructions behind an MFMA are free (11 consumed 1 KiB wave-loads per
trip: 1.02 x the MFMAs alone, even one per MFMA), s_nop is free, operand banks do not matter; a VALU instruction costs 4 - 9
cycles (it runs on the lanes the fp32 MFMA uses).  The compiled probes that said otherwise (tools/wave_occupancy_probe.hip) were
measuring v_accvgpr copies the register allocator had put into their loops."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_gen")


def mfma(t, a, b):
    return f"v_mfma_f32_16x16x4_f32 a[{4 * t}:{4 * t + 3}], v{a}, v{b}, a[{4 * t}:{4 * t + 3}]"


def body(pieces=None, a_of=lambda e, t: 32 + e, b_of=lambda e, t: 64 + 4 * (t % 11) + e, top="s_waitcnt vmcnt(0) lgkmcnt(0)"):
    pieces = pieces or {}
    lines = [top]
    for e in range(4):
        for t in range(33):
            lines.append(mfma(t, a_of(e, t), b_of(e, t)))
            lines.extend(pieces.get(e * 33 + t, []))
    return "\\n".join(lines)


def spread(n, mk, start=0, span=132, every=None):
    d = {}
    for i in range(n):
        s = start + (i * every if every else (i * span) // n)
        d.setdefault(s, []).extend(mk(i))
    return d


def merge(*ds):
    out = {}
    for d in ds:
        for k, v in d.items():
            out.setdefault(k, []).extend(v)
    return out


fm = lambda i: [f"v_fmac_f32 v{160 + i % 16}, v{176 + i % 8}, v{184 + i % 8}"]
tri = lambda i: [f"v_min_f32 v{160 + i % 16}, 0, v{176 + i % 8}", f"v_max_f32 v{192 + i % 16}, 0, v{176 + i % 8}",
                 f"v_fmac_f32 v{192 + i % 16}, v184, v{160 + i % 16}"]
a64 = lambda i: [f"v_lshl_add_u64 v[{208 + 2 * (i % 8)}:{209 + 2 * (i % 8)}], v[224:225], 0, v[226:227]"]
gl_con = lambda p: [f"global_load_dwordx4 v[{64 + 4 * p}:{67 + 4 * p}], v{4 + p}, %2"]
gl_un = lambda p: [f"global_load_dwordx4 v[{112 + 4 * p}:{115 + 4 * p}], v{4 + p}, %2"]
gst = lambda p: [f"global_store_dwordx4 v{4 + p}, v[{112 + 4 * (p % 4)}:{115 + 4 * (p % 4)}], %2 offset:2048"]
dsr = lambda p: [f"ds_read_b128 v[{128 + 4 * p}:{131 + 4 * p}], v{4 + p}"]
ds_con = lambda p: [f"ds_read_b128 v[{64 + 4 * p}:{67 + 4 * p}], v{4 + p}"]
gl_lds = lambda p: [f"s_mov_b32 m0, {p * 1024}", f"global_load_lds_dwordx4 v{4 + p}, %2"]
nop = lambda i: ["s_nop 0"]

VARIANTS = [
    ("132 MFMAs alone", body()),
    ("A, B operands constant", body(a_of=lambda e, t: 32, b_of=lambda e, t: 64)),
    ("A and B in the same register bank", body(a_of=lambda e, t: 32 + e)),
    ("A one bank above B", body(a_of=lambda e, t: 33 + e)),
    ("A two banks above B", body(a_of=lambda e, t: 34 + e)),
    ("+ 11 global wave-loads, not consumed, one per 12 MFMAs", body(spread(11, gl_un))),
    ("+ 11 global wave-loads consumed by the next trip, one per 12", body(spread(11, gl_con))),
    ("+ the same, one per 4 MFMAs from the top", body(spread(11, gl_con, every=4))),
    ("+ the same, one per MFMA from the top", body(spread(11, gl_con, every=1))),
    ("+ 11 ds_read_b128 consumed by the next trip", body(spread(11, ds_con))),
    ("+ 11 LDS-DMA wave-loads (global_load_lds_dwordx4)", body(spread(11, gl_lds))),
    ("+ 11 loads, 3 stores, 3 LDS reads", body(merge(spread(11, gl_con), spread(3, gst, 5), spread(3, dsr, 9)))),
    ("+ 132 s_nop 0", body(spread(132, nop))),
    ("+ 11 v_lshl_add_u64", body(spread(11, a64))),
    ("+ 36 v_fmac_f32, spread", body(spread(36, fm))),
    ("+ 12 PReLU triples (min, max, fmac)", body(spread(12, tri))),
    ("+ 36 PReLU triples", body(spread(36, tri))),
    ("+ loads, stores, LDS reads, 12 triples, 11 add_u64", body(merge(spread(11, gl_con), spread(3, gst, 5), spread(3, dsr, 9), spread(12, tri, 2), spread(11, a64, 1)))),
    ("+ the same with 36 triples", body(merge(spread(11, gl_con), spread(3, gst, 5), spread(3, dsr, 9), spread(36, tri, 2), spread(11, a64, 1)))),
]

CLOB_V = ", ".join(f'"v{i}"' for i in list(range(4, 15)) + list(range(32, 36)) + list(range(64, 228)))
CLOB_A = ", ".join(f'"a{i}"' for i in range(132))
INIT = "\\n".join(["v_mov_b32 v4, %3"] + [f"v_add_u32 v{5 + i}, {4096 * (i + 1)}, v4" for i in range(10)] +
                  [f"v_mov_b32 v{r}, 0" for r in list(range(32, 36)) + list(range(64, 108)) + [224, 225, 226, 227]])

SRC_HEAD = r'''// generated by tools/mfma_shadow_asm_probe.py - do not edit
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
'''

KERNEL = r'''
__global__ __launch_bounds__(256, 1) void k_%(idx)d(const float* w, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.f;
  __syncthreads();
  float r;
  unsigned lane16 = threadIdx.x * 16u;
  asm volatile("%(init)s\ns_mov_b32 s40, %%1\n1:\n%(body)s\ns_sub_u32 s40, s40, 1\ns_cmp_lg_u32 s40, 0\ns_cbranch_scc1 1b\n"
               "s_waitcnt vmcnt(0) lgkmcnt(0)\ns_nop 7\ns_nop 7\nv_accvgpr_read_b32 %%0, a0\n"
               : "=v"(r) : "s"(iters), "s"(w), "v"(lane16) : "s40", "scc", "m0", "memory", %(clob_v)s, %(clob_a)s);
  out[blockIdx.x * 256 + threadIdx.x] = r + lds[threadIdx.x];
}
'''

MAIN = r'''
template <typename F> int timeit(const char* name, F f, const float* w, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f<<<256, 256>>>(w, out, iters);
  CHECK(hipEventRecord(e0, 0));
  f<<<256, 256>>>(w, out, iters);
  CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%%-62s %%6.0f cycles per trip at 2.4 GHz (the MFMAs alone: 4224)\n", name, ms * 1e-3 * 2.4e9 / iters);
  return 0;
}
int main() {
  float *out, *w; CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&w, 1 << 20)); CHECK(hipMemset(w, 0, 1 << 20));
  printf("# one wave per SIMD, 256 work-groups, 132 v_mfma_f32_16x16x4_f32 per trip, one inline-asm block with fixed registers\n");
%(calls)s
  return 0;
}
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    src = SRC_HEAD
    calls = []
    for idx, (name, b) in enumerate(VARIANTS):
        src += KERNEL % dict(idx=idx, init=INIT, body=b, clob_v=CLOB_V, clob_a=CLOB_A)
        calls.append(f'  if (timeit("{name}", k_{idx}, w, out)) return 1;')
    src += MAIN % dict(calls="\n".join(calls))
    path = os.path.join(OUT, "mfma_shadow_asm_probe.hip")
    with open(path, "w") as f:
        f.write(src)
    exe = os.path.join(OUT, "mfma_shadow_asm_probe")
    res = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-o", exe, path], capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stdout + res.stderr)
        return 1
    print(exe)
    return 0


if __name__ == "__main__":
    sys.exit(main())
