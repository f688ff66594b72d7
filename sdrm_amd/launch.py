"""Starting the ranks of a one-node run: one process per GPU, `torch.distributed` rendezvous on 127.0.0.1.

`launch_ranks` is what `python bench.py --gpus N` uses when it is started as a plain command: the parent - which must not have
touched the GPU yet - starts `python -m torch.distributed.run ...` as a CHILD process (never exec: a process that has
initialised HIP must not be replaced), lets it inherit stdout / stderr (rank 0 prints the JSON line) and returns its exit
code.  Already inside a launcher (RANK / WORLD_SIZE set) nothing is started."""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def inside_launcher(env=None) -> bool:
    env = os.environ if env is None else env
    return "RANK" in env or "WORLD_SIZE" in env or "LOCAL_RANK" in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_command(script: str, argv, nprocs: int, port: int):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nprocs)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script, *argv]


def launch_ranks(script: str, argv, nprocs: int, env=None, timeout=None, stdout=None) -> int:
    """Runs `script argv` as `nprocs` ranks; returns the launcher's exit code (non-zero if any rank failed).  `stdout`: the file
    descriptor the ranks inherit as their stdout (default: this process's)."""
    if nprocs < 1:
        raise ValueError("nprocs must be >= 1")
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = rank_command(script, list(argv), nprocs, free_port())
    try:
        return subprocess.run(cmd, env=e, timeout=timeout, stdout=stdout).returncode
    except subprocess.TimeoutExpired:
        print(f"launch_ranks: {nprocs} ranks of {os.path.basename(script)} did not finish within {timeout} s", file=sys.stderr)
        return 124
