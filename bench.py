#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec (train + sample) on ML-1M-shaped latents.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts its N ranks itself, sdrm_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3]; SURVEY.md §8d "C3/C4"): eps-net L=W=340, T=78, H=1
(README ML-1M/MLP hyper-parameters), synthetic N(0,1) latents, global train batch B=8192 users,
15 epochs x 1 batch, then full-resolution reverse sampling of 5429 users for 78 steps.  One "step"
is one pass of the hot path over one batch: either a train step (q_sample + 3 eps-net forwards +
score-matching loss + backward + Adam; train_SDRM.py:326-337) or a reverse-sampling step (eps-net
forward + denoise_add_noise; train_SDRM.py:56-59).  The K timed steps walk the job's own 15:78
train:sample mix cyclically.  Randomness is the engine's Philox mode.  N>1: the batch and the sampled
users are sharded over ranks (strong scaling: the job is fixed), RCCL all-reduce of 5 loss scalars and
of the flat gradient per train step, no communication while sampling.

Timing protocol (warm-up, then the median of 5 repeats): an untimed pre-heat of one whole job cycle
(93 steps: clocks, caches, every kernel variant of the mix loaded) plus the --warmup steps, then the --steps window is
timed FIVE times back to back, each window bracketed by barrier + device sync on both sides and reduced with MAX over
ranks; `value` / `ms_per_step` are the MEDIAN window, `window_min_ms` / `window_max_ms` the extremes.  A separate,
untimed-for-throughput pass records HIP events around every GEMM launch (on the launch stream) for `roofline`; it runs
at least two job cycles so that the dominant class is averaged over >= 30 launches whatever --steps is.

Prints ONE JSON line on rank 0."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from sdrm_amd import synth  # noqa: E402

WL = dict(name="ML-1M/MLP shaped synthetic latents (C3/C4)", L=340, W=340, T=78, H=1, B=8192, n_sample=5429,
          epochs=15, batches_per_epoch=1, lr=9.8e-5, nd=1.0)
PEAK_FP32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak


def is_train(k: int, n_train: int, cycle: int) -> bool:
    """Bresenham spread of n_train train steps over a cycle of `cycle` steps."""
    j = k % cycle
    return ((j + 1) * n_train) // cycle > (j * n_train) // cycle


def train_flops(B, L, T, H):      # 6*B*(2T^2 + 3TL + (3H+5)L^2)
    return 6.0 * B * (2 * T * T + 3 * T * L + (3 * H + 5) * L * L)


def sample_flops(n, L, T, H):     # n * 2*(T^2 + T*L + (H+2)*L^2)
    return 2.0 * n * (T * T + T * L + (H + 2) * L * L)


class Job:
    """Walks the job's step mix on one rank."""

    def __init__(self, engine, trainer, x0_local, row0, n_local, srow0, wl, seed=1234):
        self.e, self.tr, self.x0, self.row0 = engine, trainer, x0_local, row0
        self.n_local, self.srow0, self.wl, self.seed = n_local, srow0, wl, seed
        self.n_train = wl["epochs"] * wl["batches_per_epoch"]
        self.cycle = self.n_train + wl["T"]
        self.k = 0
        self.train_count = 0
        self.sampling = False
        self.call_id = 0
        self.handshake_timeouts = 0
        self.left = 0

    def guarded(self, call, retry):
        """The launches that synchronise work-groups through an XCD's L2 (csrc/rows48.h, csrc/sample_persist.h) wait with a bound; a
        timed-out hand-shake is reported by the NEXT call, before it does anything, and the engine has then switched that path off for
        good.  A throughput run goes on (the call is made again; an invalid sampling result is dropped) and says so on its line
        (`handshake_timeouts`) instead of losing the whole measurement; anything else is raised."""
        try:
            return call()
        except Exception as ex:   # noqa: BLE001
            if "timed out" not in str(ex) or self.handshake_timeouts >= 3:
                raise
            self.handshake_timeouts += 1
            print(f"bench.py: {ex} - continuing without that path", file=sys.stderr)
            return call() if retry else None

    def lr(self):
        ep = (self.train_count // self.wl["batches_per_epoch"]) % self.wl["epochs"]
        return self.wl["lr"] * (1 - ep / self.wl["epochs"])      # train_SDRM.py:316

    def advance(self, budget):
        """The next train step, or the walk's next run of consecutive sampling steps - at most `budget`, never past the end of the
        sampling call - in ONE sdrm_sample_steps call (the reference's sampler is one loop over all T steps, train_SDRM.py:50-61; a call
        per step is host time, and one rank of eight is bound by the host's enqueue rate).  Returns (steps taken, train steps taken)."""
        wl = self.wl
        if is_train(self.k, self.n_train, self.cycle):
            self.step()
            return 1, 1
        if not self.sampling:
            self.e.sample_begin(self.n_local, nd=wl["nd"], seed=self.seed, call_id=self.call_id, row0=self.srow0)
            self.sampling = True
            self.call_id += 1
            self.left = wl["T"]
        m = 1
        while m < min(budget, self.left) and not is_train(self.k + m, self.n_train, self.cycle):
            m += 1
        self.left = self.e.sample_steps(m)
        if self.left == 0:
            self.guarded(self.e.sample_end, retry=False)
            self.sampling = False
        self.k += m
        return m, 0

    def step(self):
        wl = self.wl
        if is_train(self.k, self.n_train, self.cycle):
            self.guarded(lambda: self.tr.train_step(self.x0, self.lr(), row0=self.row0, step=self.train_count, seed=self.seed, nd=wl["nd"]),
                         retry=True)
            self.train_count += 1
            kind = "train"
        else:
            if not self.sampling:
                self.e.sample_begin(self.n_local, nd=wl["nd"], seed=self.seed, call_id=self.call_id, row0=self.srow0)
                self.sampling = True
                self.call_id += 1
                self.left = wl["T"]
            self.left = self.e.sample_steps(1)
            if self.left == 0:
                self.guarded(self.e.sample_end, retry=False)
                self.sampling = False
            kind = "sample"
        self.k += 1
        return kind


def host_cores() -> int:
    """Cores this process may actually use: the cgroup CPU quota if there is one (the GPU box exposes
    256 logical CPUs but grants 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(wl, min_seconds=10.0, max_seconds=30.0):
    """The CPU oracle (torch CPU ops, all host cores) on the same step mix: whole job cycles until at least
    `min_seconds` of CPU work have been timed, or the prefix of the walk that fits `max_seconds` on a slow host."""
    from oracle import sdrm_oracle as orc
    L, W, T, H, B, n = wl["L"], wl["W"], wl["T"], wl["H"], wl["B"], wl["n_sample"]
    cores = host_cores()
    torch.set_num_threads(cores)
    o = orc.Oracle(L, W, T, H, synth.init_params(L, W, T, H, seed=1))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0))
    g = torch.Generator().manual_seed(0)
    n_train = wl["epochs"] * wl["batches_per_epoch"]
    cycle = n_train + T
    x = torch.randn(n, L, generator=g)
    done = {"train": 0, "sample": 0}
    i = T
    t0 = time.perf_counter()
    k = 0
    while True:
        if is_train(k, n_train, cycle):
            eps = torch.randn(B, L, generator=g) * wl["nd"]
            t = torch.randint(1, T + 1, (B,), generator=g)
            keeps = [(torch.rand(B, L, generator=g) < 0.5).float() for _ in range(3)]
            o.train_step(x0, eps, t, keeps, wl["lr"])
            done["train"] += 1
        else:
            keep = (torch.rand(n, L, generator=g) < 0.5).float()
            z = torch.randn(n, L, generator=g) * wl["nd"] if i > 1 else torch.zeros(n, L)
            eps_hat = o.forward(x, torch.full((n,), i, dtype=torch.int64), keep)
            x = orc.reverse_update(x, eps_hat, z, i, o.beta, o.alpha, o.alphabar)
            i = i - 1 if i > 1 else T
            done["sample"] += 1
        k += 1
        el = time.perf_counter() - t0
        if (k % cycle == 0 and el >= min_seconds) or (el > max_seconds and done["train"] >= 2 and done["sample"] >= 10):
            break
    dt = time.perf_counter() - t0
    steps = done["train"] + done["sample"]
    return {"value": steps / dt, "unit": "denoising-steps/s", "cores": cores, "kind": "port",
            "sample": f"{done['train']} train steps (B={B}) + {done['sample']} reverse-sampling steps (n={n}) of the "
                      f"same 15:78 mix ({k / cycle:.2f} job cycles), oracle/sdrm_oracle.py on torch CPU ops, {cores} threads, {dt:.1f} s"}


OTHER = {   # BASELINE.json configs besides the benched one (README hyper-parameters)
    "ML-100k/SVD  B=550 L=830 T=83 H=2 n=843": dict(L=830, W=830, T=83, H=2, B=550, n=843),
    "ML-1M/MLP    B=160 L=340 T=78 H=1 n=5429 (README batch)": dict(L=340, W=340, T=78, H=1, B=160, n=5429),
    # the batch sweep of C3 (160, 512, 2048, 8192): train step only - the sampled rows are the same
    "ML-1M/MLP    B=512 L=340 T=78 H=1 (batch sweep)": dict(L=340, W=340, T=78, H=1, B=512, n=0),
    "ML-1M/MLP    B=2048 L=340 T=78 H=1 (batch sweep)": dict(L=340, W=340, T=78, H=1, B=2048, n=0),
    "ML-1M/MLP    B=4096 L=340 T=78 H=1 (batch sweep; the 2-GPU shard of the 8192 batch)": dict(L=340, W=340, T=78, H=1, B=4096, n=0),
    "ADM/NeuMF    B=850 L=40 T=93 H=5 n=9558": dict(L=40, W=40, T=93, H=5, B=850, n=9558),
}


def cpu_config_rates(c, seconds=2.0):
    """The CPU oracle (torch CPU ops, all granted host cores) on one of the other configs: train steps for about `seconds`,
    then full-resolution reverse steps for about `seconds` - the same-box CPU figure beside each GPU figure of
    `other_configs` (BASELINE.json config 2 is literally "HIP denoiser vs CPU baseline")."""
    from oracle import sdrm_oracle as orc
    L, W, T, H, B, n = c["L"], c["W"], c["T"], c["H"], c["B"], c["n"]
    cores = host_cores()
    torch.set_num_threads(cores)
    o = orc.Oracle(L, W, T, H, synth.init_params(L, W, T, H, seed=1))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0))
    g = torch.Generator().manual_seed(0)

    def train():
        eps = torch.randn(B, L, generator=g)
        t = torch.randint(1, T + 1, (B,), generator=g)
        keeps = [(torch.rand(B, L, generator=g) < 0.5).float() for _ in range(3)]
        o.train_step(x0, eps, t, keeps, 1e-5)

    out = {"cores": cores}
    train()   # warm-up
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or k < 2:
        train()
        k += 1
    out["train_steps_per_s"] = round(k / (time.perf_counter() - t0), 2)
    if n > 0:
        x = torch.randn(n, L, generator=g)
        k, i, t0 = 0, T, time.perf_counter()
        while time.perf_counter() - t0 < seconds or k < 2:
            keep = (torch.rand(n, L, generator=g) < 0.5).float()
            z = torch.randn(n, L, generator=g) if i > 1 else torch.zeros(n, L)
            x = orc.reverse_update(x, o.forward(x, torch.full((n,), i, dtype=torch.int64), keep), z, i, o.beta, o.alpha, o.alphabar)
            i = i - 1 if i > 1 else T
            k += 1
        out["sample_steps_per_s"] = round(k / (time.perf_counter() - t0), 2)
    return out


HBM_ACHIEVABLE_TBS = 6.3   # achievable HBM stream rate the byte roofline is priced against


def train_bytes(B, L, W, T, H):   # section 8d minimum: x0 + eps in, 6 passes over the P parameters (p, m, v read + write); PHILOX: no masks
    return 2.0 * B * L * 4 + 6.0 * synth.param_count(L, W, T, H) * 4


def sample_bytes(n, L):           # PHILOX mode: x read + x write
    return 2.0 * n * L * 4


def other_configs(with_cpu=True):
    """The other BASELINE.json configs (parity-test cases, not the headline): steps/s, kernel launches per step and the
    section-8d roofline time max(flops / 157.3 TF, bytes / 6.3 TB/s) beside the measured step time.  These are latency
    bound (2-164 us of roofline time in 3-12 launches), so `frac` says how far the launch chain is from the arithmetic.
    with_cpu: the CPU oracle on the same shapes and the same box beside each figure (`cpu_steps_per_s`, `cpu_cores`)."""
    from sdrm_amd.engine import Engine
    res = {}
    for name, c in OTHER.items():
        L, W, T, H, B, n = c["L"], c["W"], c["T"], c["H"], c["B"], c["n"]
        eng = Engine(L, W, T, H, max_rows=max(B, n))
        eng.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
        x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
        entry = {}
        for kind in (("train", "sample_full", "sample_multires") if n > 0 else ("train",)):
            if kind == "train":
                fn, reps = (lambda: eng.train_step(x0, 1e-5, seed=1, step=3)), 100
                roof_us = max(train_flops(B, L, T, H) / (PEAK_FP32_TFLOPS * 1e12), train_bytes(B, L, W, T, H) / (HBM_ACHIEVABLE_TBS * 1e12)) * 1e6
            else:
                multires = kind == "sample_multires"
                eng.sample_begin(n, seed=2, call_id=1, multires=multires)

                def fn():
                    if eng.sample_steps(1) == 0:
                        eng.sample_end()
                        eng.sample_begin(n, seed=2, call_id=1, multires=multires)
                reps = 2 * T
                rows = n / 2.0 if multires else n     # multi-resolution: on average half the rows are active per step
                roof_us = max(sample_flops(rows, L, T, H) / (PEAK_FP32_TFLOPS * 1e12), sample_bytes(rows, L) / (HBM_ACHIEVABLE_TBS * 1e12)) * 1e6
            windows = []
            for w in range(6):       # first window = warm-up, then the median of 5
                torch.cuda.synchronize()
                l0, t0 = eng.launch_count(), time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                windows.append((time.perf_counter() - t0, eng.launch_count() - l0))
            dt = float(np.median([w[0] for w in windows[1:]]))
            step_us = dt / reps * 1e6
            entry[kind] = {"steps_per_s": round(reps / dt, 1), "step_us": round(step_us, 2),
                           "launches_per_step": round(windows[-1][1] / reps, 2),
                           "roofline_us": round(roof_us, 2), "frac": round(roof_us / step_us, 4)}
            if kind != "train":
                while eng.sample_steps(T) != 0:
                    pass
                eng.sample_end()
        eng.close()
        if with_cpu:
            cpu = cpu_config_rates(c)
            entry["train"]["cpu_steps_per_s"] = cpu["train_steps_per_s"]
            entry["train"]["speedup_vs_cpu"] = round(entry["train"]["steps_per_s"] / cpu["train_steps_per_s"], 1)
            if "sample_full" in entry:
                entry["sample_full"]["cpu_steps_per_s"] = cpu["sample_steps_per_s"]
                entry["sample_full"]["speedup_vs_cpu"] = round(entry["sample_full"]["steps_per_s"] / cpu["sample_steps_per_s"], 1)
            entry["cpu_cores"] = cpu["cores"]
            entry["cpu_kind"] = "port (oracle/sdrm_oracle.py on torch CPU ops, about 2 s per leg)"
        res[name] = entry
    return res


def _profile_file(pattern: str):
    """Newest committed profile of that kind (profiles/rNN_*; names sort by round)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", pattern)))
    return files[-1] if files else None


def kernel_needle(kernel_class: str):
    """What identifies the kernel of a profile class inside a rocprofv3 kernel name: the template arguments of the GEMM classes
    ("gemm_kernel<0,0,1,0,1>" -> ", 0, 0, 1, 0, 1>("), or the name of the stand-alone kernels ("k_wgrad_strips")."""
    import re
    m = re.search(r"<(\d+),(\d+),(\d+),(\d+),(\d+)>", kernel_class)
    if m:
        return ", " + ", ".join(m.groups()) + ">("
    m = re.search(r"\b(k_[a-z0-9_]+)", kernel_class)
    return "::" + m.group(1) + "<" if m else None


def pmc_traffic(kernel_class: str, lib_hash: str):
    """HBM bytes per launch of the dominant kernel REPLAYED from the committed PMC passes (profiles/*pmc_traffic.json,
    made by tools/pmc_summary.py from two separate `rocprofv3 --pmc` runs of this same command; gfx950 FETCH_SIZE
    correction applied there) - counters cannot be read inside an unprofiled run.  Returns (bytes or None, provenance):
    the value is dropped when the file was collected from other kernel sources than the library loaded now."""
    path = _profile_file("*pmc_traffic.json")
    needle = kernel_needle(kernel_class)
    if not path or not needle:
        return None, {"file": None}
    data = json.load(open(path))
    src = {"file": os.path.basename(path), "git_head": data.get("git_head"), "source_hash": (data.get("source_hash") or "")[:16] or None,
           "matches_loaded_library": bool(data.get("source_hash")) and data.get("source_hash") == lib_hash}
    if not src["matches_loaded_library"]:
        return None, src
    for name, v in data.get("kernels", {}).items():
        if needle in name:
            # the same template serves train launches (24576 rows: the largest grid) and sampling launches
            rows = sorted(v.get("by_grid", []), key=lambda r: r["grid_size"])
            if rows:
                pick = rows[0] if kernel_class.startswith("sample") else rows[-1]
                return pick["hbm_bytes_per_launch"], src
            return v["hbm_bytes_per_launch_mean"], src
    return None, src


def pmc_mfma_busy(kernel_class: str, lib_hash: str):
    """Matrix-pipe busy cycles per SIMD and launch of the dominant kernel, replayed from the committed SQ counter pass
    (profiles/*pmc_sq.json, made by tools/pmc_sq.py from `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ...` of this command);
    None when that pass belongs to other kernel sources."""
    path = _profile_file("*pmc_sq.json")
    needle = kernel_needle(kernel_class)
    if not path or not needle:
        return None
    data = json.load(open(path))
    if not data.get("source_hash") or data.get("source_hash") != lib_hash:
        return None
    for name, rows in data.get("kernels", {}).items():
        if needle in name and rows:
            rows = sorted(rows, key=lambda r: r["grid_size"])
            return (rows[0] if kernel_class.startswith("sample") else rows[-1])["mfma_busy_per_simd_cycles"]
    return None


def pick_windows(K: int, cycle: int, at_least: int = 5, at_most: int = 24) -> int:
    """Number of timed K-step windows: the smallest count >= at_least whose total is closest to whole job cycles."""
    best, best_err = at_least, None
    for w in range(max(1, at_least), max(at_least, at_most) + 1):
        tot = w * K
        err = abs(tot - round(tot / cycle) * cycle) / tot
        if best_err is None or err < best_err - 1e-12:
            best, best_err = w, err
    return best


def combine_windows(window_ms, window_trains, K):
    """(seconds per K-step window, {"train": t, "sample": s} average step counts of a window): median duration within
    each train-step-count class, classes weighted by their number of windows."""
    classes = {}
    for ms, n_tr in zip(window_ms, window_trains):
        classes.setdefault(n_tr, []).append(ms)
    n = len(window_ms)
    ms = sum(len(v) * float(np.median(v)) for v in classes.values()) / n
    trains = sum(len(v) * k for k, v in classes.items()) / n
    return ms * 1e-3, {"train": trains, "sample": K - trains}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=186)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra legs on the other BASELINE.json configs (steps/s, launches per step, roofline time)")
    ap.add_argument("--windows", type=int, default=5, help="least number of timed repeats of the --steps window")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--exchange", default="rccl-abi", choices=["rccl-abi", "torch"],
                    help="N > 1: collectives issued inside libsdrm_hip.so over RCCL (default) or by torch.distributed between the phases")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="N = 1 only: run the train steps through sdrm_train_step_sharded over a ONE-rank RCCL communicator (every "
                         "collective of the N > 1 path is really issued); a rehearsal of that path, not the headline number")
    ap.add_argument("--rehearse-shard", type=int, default=0, metavar="N",
                    help="N = 1 only: give this process the rows ONE rank of an N-GPU run would have (B/N users, n/N sampled rows) and run "
                         "them through the one-rank exchange (implies --rehearse-exchange): that rank's compute plus the launch side of "
                         "the collectives under the bench protocol, no link time - a projection aid, never the headline number")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: put every rank on cuda:0 (a 1-GPU box), implies a non-RCCL backend")
    ap.add_argument("--stub-engine", action="store_true",
                    help="tests only (tests/test_bench_launch.py): run the whole protocol - self-launch, rendezvous, windows, MAX over "
                         "ranks, the JSON line - on the CPU over gloo with tests/bench_stub.py in place of the engine; measures nothing")
    ap.add_argument("--comm-init-timeout", type=float, default=180.0,
                    help="N > 1: seconds a rank may spend creating its RCCL communicator before the run exits non-zero instead of hanging")
    args = ap.parse_args()

    # Started as a plain command with --gpus N > 1: this process has not touched the GPU yet - start the N ranks as a child
    # launcher (sdrm_amd/launch.py), relay rank 0's line (the children inherit stdout) and its exit code.
    from sdrm_amd import launch
    if args.gpus > 1 and not launch.inside_launcher():
        sys.exit(launch.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, stdout=REAL_STDOUT))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    stub = args.stub_engine
    if stub and os.environ.get("SDRM_BENCH_TEST_FAIL_RANK") == str(rank):
        raise SystemExit(f"bench.py: rank {rank} asked to fail (test hook of tests/test_bench_launch.py)")
    if stub:
        args.backend, args.exchange, args.no_cpu_baseline, args.no_other_configs = "gloo", "torch", True, True
    dev = "cpu" if stub else "cuda"
    device_sync = (lambda: None) if stub else torch.cuda.synchronize
    if not stub:
        torch.cuda.set_device(0 if args.share_gpu else local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    if stub and os.environ.get("SDRM_BENCH_TEST_FAIL_RANK_LATE") == str(rank):
        # (test hook: a rank that dies AFTER the rendezvous - its peers are then inside collectives; the launcher must end them)
        raise SystemExit(f"bench.py: rank {rank} asked to fail after the rendezvous (test hook of tests/test_bench_launch.py)")
    if stub:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from bench_stub import StubEngine as Engine
    else:
        from sdrm_amd.engine import Engine
    from sdrm_amd.parallel import RcclTrainer, ShardedTrainer, shard_rows

    wl = WL
    L, W, T, H, B, n = wl["L"], wl["W"], wl["T"], wl["H"], wl["B"], wl["n_sample"]
    row0, rows = shard_rows(B, rank, world)
    srow0, n_local = shard_rows(n, rank, world)
    if world == 1 and args.rehearse_shard > 1:
        args.rehearse_exchange = True
        row0, rows = shard_rows(B, 0, args.rehearse_shard)
        srow0, n_local = shard_rows(n, 0, args.rehearse_shard)
    eng = Engine(L, W, T, H, max_rows=max(rows, n_local))
    eng.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)[row0:row0 + rows]).to(dev)
    # N > 1 over RCCL: the exchange is issued by the library itself (sdrm_train_step_sharded); --exchange torch keeps the
    # torch.distributed collectives between the three phases (the only form a gloo rehearsal can run)
    use_abi = world > 1 and args.exchange == "rccl-abi" and args.backend == "nccl"
    trainer, exchange_used = None, ("none" if world == 1 else "torch.distributed between the phases")
    if use_abi:
        # Every rank must end up on the same exchange, and ncclCommInitRank is collective: a rank that fails BEFORE it would
        # leave its peers blocked inside it.  So the ranks agree first, over the torch process group, on what each can check
        # locally (librccl resolves, the engine sits on this rank's device); only then do all of them create the communicator,
        # or none.  A watchdog turns a communicator that never comes up into a non-zero exit instead of a hang.
        why = ""
        try:
            local_ok = bool(Engine.comm_available()) and torch.cuda.current_device() == (0 if args.share_gpu else local_rank)
            if not local_ok:
                why = "librccl could not be loaded" if not Engine.comm_available() else "engine not on this rank's device"
        except Exception as ex:
            local_ok, why = False, f"{type(ex).__name__}: {ex}"
        ok = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.cpu()) == 1:
            import threading

            def give_up():
                print(f"bench.py: rank {rank}: the RCCL communicator did not come up within {args.comm_init_timeout:.0f} s", file=sys.stderr)
                os._exit(3)

            watchdog = threading.Timer(args.comm_init_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
            trainer = RcclTrainer(eng, rank, world)   # an exception here ends this rank non-zero; its peers' watchdogs end them
            watchdog.cancel()
            exchange_used = "RCCL inside libsdrm_hip.so (sdrm_train_step_sharded)"
        else:
            exchange_used = "torch.distributed between the phases (a rank cannot use the in-library exchange" + (": " + why if why else "") + ")"
            print("bench.py: " + exchange_used, file=sys.stderr)
    if trainer is None and world == 1 and args.rehearse_exchange:
        trainer = RcclTrainer(eng, 0, 1)
        exchange_used = "REHEARSAL: RCCL inside libsdrm_hip.so over a one-rank communicator" + (
            f", rows of one rank of {args.rehearse_shard} ({rows} users, {n_local} sampled rows)" if args.rehearse_shard > 1 else "")
    if trainer is None:
        trainer = ShardedTrainer(eng, rank, world)
    job = Job(eng, trainer, x0, row0, n_local, srow0, wl)

    def barrier():
        device_sync()
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        device_sync()

    # ---- pre-heat: one whole job cycle whatever --warmup says (a 5-step warm-up leaves the first timed train steps
    # cold: round 1's driver line read 20 % under the warm numbers), then the caller's warm-up steps
    n_cycle = wl["epochs"] * wl["batches_per_epoch"] + T
    for _ in range(n_cycle + args.warmup):
        job.step()
    # ---- the --steps window, timed several times back to back along the job's cyclic walk; each window: barrier + sync,
    # EXACTLY K steps, barrier + sync, MAX over ranks.  A window shorter than a job cycle holds a whole number of train
    # steps (20 steps: 3 or 4, i.e. 15 % or 20 % against the job's 16.1 %), so windows are only comparable within their
    # train-step count: the figure reported is the MEDIAN window of each composition class, the classes weighted by how
    # often the walk produces them, over a number of windows that covers whole job cycles - robust like a median,
    # unbiased in the train:sample mix whatever --steps is.
    n_windows = pick_windows(args.steps, n_cycle, args.windows)
    window_ms, window_trains, enqueue_ms = [], [], []
    for w in range(n_windows):
        barrier()
        t0 = time.perf_counter()
        n_tr = 0
        done = 0
        while done < args.steps:                      # EXACTLY K steps: a run of sampling steps never crosses the window's end
            m, tr = job.advance(args.steps - done)
            done += m
            n_tr += tr
        enqueue_ms.append((time.perf_counter() - t0) * 1e3)   # the host's share: the loop without the synchronisation behind it
        barrier()
        dt_w = time.perf_counter() - t0
        tmax = torch.tensor([dt_w], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        window_ms.append(float(tmax.cpu()) * 1e3)
        window_trains.append(n_tr)
    dt, kinds = combine_windows(window_ms, window_trains, args.steps)

    sampler_chains = int(getattr(eng, "sampler_chains", 1))   # (concurrent row chains of the sampling call, by size)
    # The event passes time the sampler on ONE chain: intervals of launches that share the chip overlap (and the library would run the
    # chains one after the other under an event profile: half-size launches that cannot fill the chip alone) - either way not a
    # per-kernel figure.  One chain is the same kernels on full-size launches; the timed windows above ran `sampler_chains`.
    if sampler_chains > 1 and hasattr(eng, "debug_set"):
        while job.k % job.cycle != 0:      # to the start of a job cycle: no sampling call open
            job.step()
        eng.debug_set(chains=1)
    # ---- second, untimed-for-throughput pass with HIP events around every GEMM launch: at least two job cycles, so the
    # dominant class (one batched weight-gradient launch per train step) is averaged over >= 30 launches
    prof_steps = max(args.steps, 2 * n_cycle)
    eng.profile_begin(capacity=prof_steps * 16)
    l0 = eng.launch_count()
    prof_kinds = {"train": 0, "sample": 0}
    for _ in range(prof_steps):
        prof_kinds[job.step()] += 1
    launches_total = eng.launch_count() - l0
    prof = eng.profile_end()
    # ---- third pass: the dominant class alone.  Every bracketed launch puts two marker packets on the stream and the markers of
    # neighbouring launches inflate each other's intervals (3-10 us per launch: a class of 312 short launches gains a millisecond);
    # with only one class bracketed its interval is the kernel plus one dispatch gap - the figure rocprofv3 --kernel-trace agrees
    # with.  The three largest classes of the pass above are each timed alone; the dominant class is the largest of THOSE totals.
    prof_dom, dom_name = None, None
    if prof:
        alone = {}
        for cand, _ in sorted(prof.items(), key=lambda kv: -kv[1][0])[:3]:
            eng.profile_begin(capacity=prof_steps * 4, only=cand)
            for _ in range(prof_steps):
                job.step()
            got = eng.profile_end().get(cand)
            if got:
                alone[cand] = got
        if alone:
            dom_name = max(alone.items(), key=lambda kv: kv[1][0])[0]
            prof_dom = alone[dom_name]
        eng.profile_begin(capacity=1)          # (all classes again for any later caller)
        eng.profile_end()
    if sampler_chains > 1 and hasattr(eng, "debug_set"):
        eng.debug_set(chains=-1)

    # ---- separate train-only / sample-only rates (extra information, not `value`)
    def rate(fn, reps):
        barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return reps / (time.perf_counter() - t)

    if job.sampling:                      # drain a sampling call left open by the cyclic walk
        while eng.sample_steps(1) != 0:
            pass
        eng.sample_end()
        job.sampling = False
    train_rate = rate(lambda: trainer.train_step(x0, wl["lr"], row0=row0, step=7, seed=99, nd=wl["nd"]), 20)
    eng.sample_begin(n_local, nd=wl["nd"], seed=5, call_id=777, row0=srow0)
    sample_rate = rate(lambda: eng.sample_steps(1), 60)

    # what every rank ran on: its device, its rows, the communicator it sees (a SCALE record then shows that RCCL saw N ranks)
    comm_n, comm_r = eng.comm_info()
    mine = {"rank": rank, "device": None if stub else torch.cuda.current_device(), "train_rows": rows, "sample_rows": n_local,
            "comm_nranks": comm_n, "comm_rank": comm_r, "gradient_buckets": int(getattr(trainer, "buckets", 1))}
    rank_info = [mine]
    if world > 1:
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)

    # the in-library exchange was agreed on: every rank's communicator must then span the whole job - a record that says
    # "RCCL" while some rank ran alone is worse than no record
    if use_abi and trainer is not None and isinstance(trainer, RcclTrainer):
        bad = [r for r in rank_info if r is None or r.get("comm_nranks") != world]
        if bad:
            if rank == 0:
                print(f"bench.py: the RCCL communicator does not span the job ({world} ranks): {bad}", file=sys.stderr)
            sys.exit(4)

    if rank == 0:
        dom = (dom_name, prof[dom_name]) if (prof and dom_name in prof) else (max(prof.items(), key=lambda kv: kv[1][0]) if prof else None)
        roof = None
        if dom:
            name, (ms_all, launches_all, flops_all) = dom
            ms, launches, flops = prof_dom if prof_dom else (ms_all, launches_all, flops_all)
            achieved = flops / (ms * 1e-3) / 1e12
            lib_hash = eng.lib.sdrm_source_hash().decode()
            traffic, src = pmc_traffic(name, lib_hash)
            roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_FP32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                    "traffic_measured_in_run": False, "traffic_source": src,
                    "avg_launch_us": round(ms * 1e3 / launches, 2), "launches": launches,
                    "event_pass_steps": prof_steps,
                    "avg_launch_us_all_classes_bracketed": round(ms_all * 1e3 / launches_all, 2),
                    "timing_note": "achieved / avg_launch_us: HIP events around the launches of this class only (its own pass; the three "
                                   "largest classes are each timed that way and the largest total names the dominant one); "
                                   "all_kernels: every GEMM class bracketed in one pass, each interval 3-10 us high from the neighbours' markers"
                                   + (f"; sampling classes: timed on one row chain (full-size launches) - the timed windows run {sampler_chains} "
                                      "concurrent chains of half-size launches, whose intervals overlap" if sampler_chains > 1 else ""),
                    "mfma_busy_cycles_per_simd": pmc_mfma_busy(name, lib_hash),
                    "mfma_util_note": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs per launch (PMC pass); divide by avg_launch_us x "
                                      "shader clock (~2.1 GHz in kernels this short, tools/mfma_probe.hip) for the pipe utilisation",
                    "algorithmic_flops_per_launch": flops / launches,
                    "all_kernels": {k: {"ms": round(v[0], 3), "launches": v[1],
                                        "tflops": round(v[2] / (v[0] * 1e-3) / 1e12, 2)} for k, v in prof.items()}}
        n_train = wl["epochs"] * wl["batches_per_epoch"]
        job_flops = (kinds["train"] * train_flops(B, L, T, H) + kinds["sample"] * sample_flops(n, L, T, H))
        out = {
            "metric": "denoising-steps/sec (train+sample) on ML-1M latents", "value": round(args.steps / dt, 2),
            "unit": "denoising-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "windows": len(window_ms), "window_ms": [round(v, 3) for v in window_ms], "window_train_steps": window_trains,
            "window_min_ms": round(min(window_ms), 3), "window_max_ms": round(max(window_ms), 3),
            "host_enqueue_frac": round(sum(enqueue_ms) / sum(window_ms), 3),   # rank 0: time spent queueing the windows' steps / their wall time
            "window_rule": "median per train-step-count class, classes weighted by frequency; windows span whole job cycles",
            "preheat_steps": n_cycle,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "latent": L, "width": W, "timesteps": T, "hidden_layers": H,
                       "global_batch": B, "n_sample": n, "step_mix": f"{n_train} train : {T} sample per job cycle",
                       "timed_train_steps": round(kinds["train"], 3), "timed_sample_steps": round(kinds["sample"], 3),
                       "rng": "philox4x32-10 on device", "sampler_row_chains": sampler_chains,
                       "parallelism": f"user-sharded dp{world}",
                       "collectives": ((exchange_used + f" [{args.backend}]"
                                        + f": all-reduce of 5 f64 loss sums + flat f32 gradient in {mine['gradient_buckets']} bucket(s) per train step")
                                       if (world > 1 or args.rehearse_exchange) else "none")},
            "exchange_used": exchange_used, "ranks": rank_info, "handshake_timeouts": job.handshake_timeouts,
            "whole_job_tflops": round(job_flops / dt / 1e12, 2),
            "train_steps_per_s": round(train_rate, 2), "sample_steps_per_s": round(sample_rate, 2),
            "launches_per_step": round(launches_total / prof_steps, 2),
            "library_source_hash": eng.lib.sdrm_source_hash().decode()[:16],
            "roofline": roof,
        }
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps(out) + "\n").encode())   # the ONE line of this command's stdout
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    # stdout carries the JSON line and nothing else: libraries that print from C (RCCL's version banner at communicator
    # creation goes to fd 1) are sent to stderr for the life of the process; the line itself is written to the saved descriptor
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main()
