"""ISA lint of the row-owned kernels (csrc/rowchain.h, csrc/dgrad_rows.h): their MFMAs are inline asm with the accumulator tied
in place (the register allocator otherwise renames accumulators inside the K loops and copies them back by the hundred), so the
compiler's hazard recogniser does not see them.  The sources guard the two places where compiler-made VALU code meets the
accumulators (rc_acc_begin / rc_acc_settle); this test cross-compiles the kernels for gfx950 (no GPU needed) and checks the
generated code for any other:

  * no accumulator copy or write (v_accvgpr_*) inside a K loop, no scratch access inside a K loop;
  * an accumulator write by VALU (v_accvgpr_write / v_accvgpr_mov) is never closer than two wait states to the next MFMA;
  * the first accumulator read (v_accvgpr_read) after an MFMA is separated from it by at least 18 wait states of s_nop
    (v_mfma_f32_16x16x4_f32: 8 passes; the CDNA3 ISA guide asks for 11 before a VALU read of the result);
  * no VALU instruction inside the K loops of the dgrad kernels at all (operands by raw buffer loads with scalar offsets).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "sdrm_amd", "csrc")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    pytest.skip("hipcc not available")


def _compile(instantiations: str) -> str:
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.hip")
        with open(src, "w") as f:
            f.write(f'#include "{CSRC}/rows48.h"\n' + instantiations)
        out = os.path.join(d, "k.s")
        res = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        return open(out).read()


def _kernels(asm: str):
    """name -> list of instruction strings (labels kept as 'LABEL name')."""
    out, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        t = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            cur.append("LABEL " + t.split(":")[0])
        elif line.startswith("\t") and t and not t.startswith(";") and not t.startswith("."):
            cur.append(t.split(";")[0].strip())
            if t.startswith("s_endpgm"):
                cur = None
    return out


def _loops(ins):
    """(start, end) index ranges of the basic-block loops: a label and the first backward branch to it."""
    labels = {x.split()[1]: i for i, x in enumerate(ins) if x.startswith("LABEL ")}
    loops = []
    for i, x in enumerate(ins):
        m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", x)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    return loops


def _wait_states(x):
    m = re.match(r"s_nop (\d+)", x)
    return int(m.group(1)) + 1 if m else (0 if x.startswith("LABEL") else 1)


VALU_OK_IN_LOOP = ("v_mfma",)


@pytest.mark.parametrize("ct", [4, 5, 7, 11])
def test_row_owned_kernels_isa(ct):
    light = "true" if ct == 11 else "false"   # the compact last K-step (340 = 21 * 16 + 4) on the headline width, plain on the other
    asm = _compile(f"template __global__ void sdrm::k_dgrad_chain<{ct}, {light}>(const sdrm::DgradChainArgs);\n"
                   f"template __global__ void sdrm::k_dgrad_rows<{ct}, {light}>(const sdrm::DgradRowsArgs);\n"
                   f"template __global__ void sdrm::k_row_fwd<{ct}, {light}>(const sdrm::RowChainArgs);\n"
                   # the same step on 48-row work-groups (csrc/rows48.h): the same asm MFMAs through the same K-step templates
                   f"template __global__ void sdrm::k_rows48_fwd<{ct}, {light}>(const sdrm::RowChainArgs);\n"
                   f"template __global__ void sdrm::k_rows48_dgrad_chain<{ct}, {light}>(const sdrm::DgradChain48Args);\n"
                   # ... and with two work-groups per row group (column-split, the form the size rule takes for 1281 .. 2048 users)
                   f"template __global__ void sdrm::k_rows48_fwd<{ct}, {light}, 2>(const sdrm::RowChainArgs);\n"
                   f"template __global__ void sdrm::k_rows48_dgrad_chain<{ct}, {light}, 2>(const sdrm::DgradChain48Args);\n"
                   # ... and the shared-tile form (4 q + 2 column tiles: the waves of a pair split one tile's K-steps)
                   + (f"template __global__ void sdrm::k_rows48_fwd<{ct}, {light}, 1, true>(const sdrm::RowChainArgs);\n"
                      f"template __global__ void sdrm::k_rows48_dgrad_chain<{ct}, {light}, 1, true>(const sdrm::DgradChain48Args);\n" if ct % 2 else ""))
    ks = _kernels(asm)
    names = {"chain": [n for n in ks if "k_dgrad_chain" in n and "rows48" not in n], "rows": [n for n in ks if "k_dgrad_rows" in n],
             "fwd": [n for n in ks if "k_row_fwd" in n], "fwd48": [n for n in ks if "k_rows48_fwd" in n and "ELi2E" not in n and "ELi1ELb1E" not in n],
             "chain48": [n for n in ks if "k_rows48_dgrad_chain" in n and "ELi2E" not in n and "ELi1ELb1E" not in n],
             "fwd48x2": [n for n in ks if "k_rows48_fwd" in n and "ELi2E" in n],
             "chain48x2": [n for n in ks if "k_rows48_dgrad_chain" in n and "ELi2E" in n]}
    if ct % 2:
        names["fwd48s"] = [n for n in ks if "k_rows48_fwd" in n and "ELi1ELb1E" in n]
        names["chain48s"] = [n for n in ks if "k_rows48_dgrad_chain" in n and "ELi1ELb1E" in n]
    assert all(len(v) == 1 for v in names.values()), names
    for kind, (name,) in names.items():
        ins = ks[name]
        mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
        assert len(mf) >= (100 if "48" not in kind else 12), (kind, len(mf))
        # K loops: the loops that hold MFMAs
        loops = _loops(ins)
        inner = [(a, b) for a, b in loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in loops)]
        # (the layer loops have barriers, a K loop has none)
        kloops = [(a, b) for a, b in inner if sum(1 for x in ins[a:b] if x.startswith("v_mfma")) >= 24 and "s_barrier" not in ins[a:b]]
        # (a K loop of a single trip is straight-line code: the narrowest nets of the dgrad kernels)
        assert kloops or (not kind.startswith("fwd") and ct <= 7), kind   # (up to 14 K-steps the dgrads' main loop is a single trip of four)
        for a, b in kloops:
            body = ins[a:b]
            assert not [x for x in body if x.startswith("v_accvgpr")], (kind, "accumulator copies inside a K loop")
            assert not [x for x in body if x.startswith("scratch_")], (kind, "scratch access inside a K loop")
            if not kind.startswith("fwd"):
                valu = [x for x in body if x.startswith("v_") and not x.startswith(VALU_OK_IN_LOOP)]
                assert not valu, (kind, "VALU inside a K loop", valu[:4])
            else:
                # the forward keeps a handful per pair of K-steps: LDS addresses of the A fragment reads and of the stream chunks
                valu = [x for x in body if x.startswith("v_") and not x.startswith(VALU_OK_IN_LOOP)]
                assert len(valu) <= 16, (kind, len(valu), valu[:6])
        # hazards around the asm MFMAs
        for i in mf:
            ws = 0
            for j in range(i - 1, max(i - 4, -1), -1):
                x = ins[j]
                if x.startswith(("v_accvgpr_write", "v_accvgpr_mov")):
                    assert ws >= 2, (kind, "VALU write of an accumulator %d wait states before an MFMA" % ws, ins[j:i + 1])
                ws += _wait_states(x)
                if ws >= 2:
                    break
        # accumulator reads: along every path out of an MFMA (branches followed), 18 wait states of s_nop come first
        labels = {x.split()[1]: i for i, x in enumerate(ins) if x.startswith("LABEL ")}

        def walk(i, nops, budget, seen):
            while i < len(ins) and budget > 0 and nops < 18:
                x = ins[i]
                if x.startswith("v_mfma") or x.startswith("s_endpgm"):
                    return
                assert not x.startswith("v_accvgpr_read"), (kind, "accumulator read %d nop states after an MFMA" % nops, ins[max(0, i - 6):i + 1])
                m = re.match(r"s_(c?branch)\w* (\.LBB\d+_\d+)", x)
                if m:
                    tgt = labels[m.group(2)]
                    if (tgt, nops) not in seen:
                        seen.add((tgt, nops))
                        walk(tgt, nops, budget - 1, seen)
                    if m.group(1) == "branch":
                        return
                nops += _wait_states(x) if x.startswith("s_nop") else 0
                i += 1
                budget -= 1

        for i in mf:
            if not ins[i + 1].startswith("v_mfma"):
                walk(i + 1, 0, 96, set())


def test_fragment_packing_is_a_bijection():
    """wfrag_index (csrc/elementwise.h) maps the (n, k) of a padded [NP][KP] weight onto the fragment-packed copy the row-owned
    kernels load from - with the compact last K-step (k = 16 klast + j in lane group j, component 0) it must still hit every
    slot exactly once, and rc_light_klast must pick that form exactly when the last padded K-step holds one to four real k.
    Host code, compiled with hipcc and run here."""
    src = r'''
#include <cstdio>
#include <vector>
#include "%s/elementwise.h"
using namespace sdrm;
int main() {
  const int cases[][3] = {{340, 352, 21}, {337, 352, 21}, {180, 192, 11}, {100, 128, -1}, {136, 160, -1}, {352, 352, -1}, {341, 352, -1},
                          {336, 352, -1}, {20, 32, 1}, {16, 32, -1}};
  for (auto& c : cases) {
    const int K = c[0], KP = c[1], want = c[2];
    const int kl = rc_light_klast(K, KP);
    if (kl != want) { printf("rc_light_klast(%%d, %%d) = %%d, expected %%d\n", K, KP, kl, want); return 1; }
    const int NP = KP, NCT = NP / 16;
    std::vector<int> hit((size_t)NP * KP, 0);
    for (int n = 0; n < NP; ++n)
      for (int k = 0; k < KP; ++k) {
        const size_t i = wfrag_index(n, k, NCT, kl);
        if (i >= hit.size()) { printf("index out of range\n"); return 1; }
        ++hit[i];
        // the layout contract of the kernels: tile (k / 16, n / 16) is 256 consecutive floats, lane = 16 * group + n %% 16
        const size_t tile = i / 256, lane = (i %% 256) / 4, comp = i %% 4;
        const int kk = k & 15;
        const bool compact = (k >> 4) == kl;
        const int group = compact ? (kk & 3) : (kk >> 2), e = compact ? (kk >> 2) : (kk & 3);
        if (tile != (size_t)(k >> 4) * NCT + (n >> 4) || lane != (size_t)(16 * group + (n & 15)) || comp != (size_t)e) { printf("layout\n"); return 1; }
      }
    for (int v : hit) if (v != 1) { printf("not a bijection at K = %%d\n", K); return 1; }
  }
  printf("ok\n");
  return 0;
}
''' % CSRC
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "h.hip")
        with open(path, "w") as f:
            f.write(src)
        exe = os.path.join(d, "h")
        res = subprocess.run([_hipcc(), "-O1", "-std=c++17", "--offload-arch=gfx950", "-o", exe, path], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        run = subprocess.run([exe], capture_output=True, text=True)
        assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr


def test_strips_kernel_loop_is_clean():
    """k_wgrad_strips (csrc/wgrad2.h) uses the MFMA builtin: the K loop must hold no accumulator copies (the allocator has been seen
    to rename accumulators inside such loops - 124 v_accvgpr_mov per three K-steps in the 16-wide experiment), no scratch, and only
    a handful of VALU instructions (its staging addresses are buffer resources + scalar offsets)."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.hip")
        with open(src, "w") as f:
            f.write(f'#include "{CSRC}/wgrad2.h"\ntemplate __global__ void sdrm::k_wgrad_strips<11>(const sdrm::Wg2Args);\n')
        out = os.path.join(d, "k.s")
        res = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        asm = open(out).read()
    ks = _kernels(asm)
    (name,) = [n for n in ks if "k_wgrad_strips" in n]
    ins = ks[name]
    loops = _loops(ins)
    inner = [(a, b) for a, b in loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in loops)]
    kloops = [(a, b) for a, b in inner if sum(1 for x in ins[a:b] if x.startswith("v_mfma")) >= 88]
    assert kloops
    for a, b in kloops:
        body = ins[a:b]
        nm = sum(1 for x in body if x.startswith("v_mfma"))
        assert sum(1 for x in body if x.startswith(("v_accvgpr_mov", "v_accvgpr_write"))) == 0
        assert not [x for x in body if x.startswith("scratch_")]
        valu = [x for x in body if x.startswith("v_") and not x.startswith(("v_mfma", "v_accvgpr_read"))]
        assert len(valu) * 8 <= nm, (len(valu), nm, valu[:6])


def test_narrow_forward_layer_reads_precede_its_mfmas():
    """k_skinny_fwd4 (csrc/skinny_fwd4.h): a hidden layer of a wave is twelve ds_read_b128 and 48 v_mfma_f32_4x4x1_16b_f32.  Left
    to itself the scheduler pairs every read with its four MFMAs - twelve exposed LDS round trips per layer (1277 cycles measured
    against 711) - so the source pins the reads in front (sched_group_barrier): here the hidden-layer loop must hold no scratch
    access, and at most two LDS reads may follow the first MFMA of the loop body."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.hip")
        with open(src, "w") as f:
            f.write(f'#include "{CSRC}/skinny_fwd4.h"\ntemplate __global__ void sdrm::k_skinny_fwd4<3, 3>(const sdrm::SkStepArgs);\n')
        out = os.path.join(d, "k.s")
        res = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        asm = open(out).read()
    ks = _kernels(asm)
    (name,) = [n for n in ks if "k_skinny_fwd4" in n]
    ins = ks[name]
    assert not [x for x in ins if x.startswith("scratch_")]
    assert sum(1 for x in ins if x.startswith("v_mfma_f32_4x4x1")) == 3 * 48
    loops = _loops(ins)
    hidden = [(a, b) for a, b in loops if sum(1 for x in ins[a:b] if x.startswith("v_mfma_f32_4x4x1")) == 48]
    assert hidden, "the hidden-layer loop was not found"
    for a, b in hidden:
        body = ins[a:b]
        first = next(i for i, x in enumerate(body) if x.startswith("v_mfma_f32_4x4x1"))
        assert sum(1 for x in body[first:] if x.startswith("ds_read_b128")) <= 2, [x for x in body[first:] if x.startswith("ds_read")]
