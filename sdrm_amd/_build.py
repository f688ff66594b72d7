"""Builds libsdrm_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension machinery).

The shared object sits next to this file so that it travels with the repository snapshot to the GPU
box and shows up as an in-tree native library in the loaded-module record."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
SOURCES = ["sdrm_hip.hip"]
HEADERS = ["gemm.h", "skinny_train.h", "elementwise.h", "philox.h", "skinny.h", "select.h", "rank.h", "feed.h", os.path.join("..", "..", "include", "sdrm_hip.h")]
LIB_PATH = os.path.join(HERE, "libsdrm_hip.so")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsdrm_hip.so")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(SRC_DIR, s) for s in SOURCES] + [os.path.normpath(os.path.join(SRC_DIR, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > built for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", LIB_PATH + ".tmp"] + [os.path.join(SRC_DIR, s) for s in SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose and (res.stdout or res.stderr):
        print(res.stdout + res.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
