"""Builds libsdrm_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension machinery).

The shared object sits next to this file so that it travels with the repository snapshot to the GPU
box and shows up as an in-tree native library in the loaded-module record.

The binary is bound to its sources: a SHA-256 over every file under csrc/ and include/ is compiled in
(`-DSDRM_SOURCE_HASH`, returned by sdrm_source_hash()) and `is_stale()` compares it with the hash of the
sources as they are now - file times play no part (a snapshot copy does not keep them)."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
INC_DIR = os.path.normpath(os.path.join(HERE, "..", "include"))
SOURCES = ["sdrm_hip.hip"]
LIB_PATH = os.path.join(HERE, "libsdrm_hip.so")
_MARKER = b"SDRM_SOURCE_HASH="


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsdrm_hip.so")


def source_files() -> list[str]:
    """Everything the library is compiled from: csrc/* and include/*.h (sorted, so the hash is stable)."""
    files = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR) if f.endswith((".h", ".hip"))]
    files += [os.path.join(INC_DIR, f) for f in os.listdir(INC_DIR) if f.endswith(".h")]
    return sorted(files, key=os.path.basename)


def source_hash() -> str:
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def binary_hash(path: str | None = None) -> str | None:
    """The source hash compiled into the library at `path` (default LIB_PATH), read from the file (no dlopen); None if absent."""
    try:
        with open(path or LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(re.escape(_MARKER) + rb"([0-9a-f]{64})", blob)
    return m.group(1).decode() if m else None


def is_stale() -> bool:
    return binary_hash() != source_hash()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compiles the library when it is missing or stale (or `force`).  Safe against concurrent callers - the ranks of a
    multi-process run that all find a stale binary: one file lock around the build, the staleness re-checked once the lock is
    held (the rank that waited finds the fresh binary and compiles nothing), a temp file of its own per process, and an atomic
    rename, so no process ever loads a half-written file."""
    if not force and not is_stale():
        return LIB_PATH
    import fcntl
    import tempfile
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return LIB_PATH
            digest = source_hash()
            fd, tmp = tempfile.mkstemp(prefix="libsdrm_hip.", suffix=f".{os.getpid()}.tmp", dir=HERE)
            os.close(fd)
            try:
                cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
                       "-Wno-unused-function", f'-DSDRM_SOURCE_HASH="{digest}"', "-o", tmp]
                cmd += [os.path.join(SRC_DIR, s) for s in SOURCES] + ["-ldl"]
                res = subprocess.run(cmd, capture_output=True, text=True)
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
                if verbose and (res.stdout or res.stderr):
                    print(res.stdout + res.stderr)
                os.chmod(tmp, 0o755)
                os.replace(tmp, LIB_PATH)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
            if binary_hash() != digest:
                raise RuntimeError("libsdrm_hip.so was built but does not carry its source hash")
            return LIB_PATH
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
