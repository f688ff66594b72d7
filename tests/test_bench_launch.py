"""`python bench.py --gpus N` as a plain command: the parent starts its N ranks itself (sdrm_amd/launch.py)
before it touches the GPU, relays rank 0's JSON line and the exit code.  Covered here without hardware: the real bench.py, the
real launcher and rendezvous (gloo on 127.0.0.1), a stand-in engine (tests/bench_stub.py)."""
import json
import os
import subprocess
import sys

import pytest

from sdrm_amd import launch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=REPO)


def json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_rank_command_is_the_drivers_form():
    cmd = launch.rank_command("bench.py", ["--gpus", "4"], 4, 29511)
    assert cmd[1:8] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port"]
    assert cmd[8:] == ["29511", "bench.py", "--gpus", "4"]
    assert not launch.inside_launcher({}) and launch.inside_launcher({"RANK": "0"}) and launch.inside_launcher({"WORLD_SIZE": "2"})


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_bench_as_a_plain_command(gpus):
    r = run_bench("--gpus", str(gpus), "--steps", "12", "--warmup", "2", "--windows", "2", "--stub-engine")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["n_gpus"] == gpus and d["steps"] == 12 and d["warmup"] == 2 and d["value"] > 0
    assert [x["rank"] for x in d["ranks"]] == list(range(gpus))
    assert sum(x["train_rows"] for x in d["ranks"]) == d["config"]["global_batch"]
    assert sum(x["sample_rows"] for x in d["ranks"]) == d["config"]["n_sample"]
    assert ("torch.distributed" in d["exchange_used"]) == (gpus > 1)


def test_a_failing_rank_ends_the_command_nonzero():
    # --gpus that the launcher does not match is refused inside the ranks: the parent must relay the failure
    r = run_bench("--gpus", "2", "--steps", "4", "--stub-engine", env_extra={"SDRM_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("hook", ["SDRM_BENCH_TEST_FAIL_RANK", "SDRM_BENCH_TEST_FAIL_RANK_LATE"])
def test_rank_5_of_8_failing_ends_the_command_nonzero_in_time(hook):
    """the first real 8-rank run must not be the first time the launcher sees eight ranks.  Rank 5 dies before the
    rendezvous (its peers wait in init_process_group) or after it (its peers sit in the first collective): either way the
    command ends non-zero, without a JSON line, long before the 180 s communicator watchdog."""
    import time
    t0 = time.time()
    r = run_bench("--gpus", "8", "--steps", "4", "--warmup", "1", "--windows", "1", "--stub-engine", env_extra={hook: "5"}, timeout=170)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank 5 asked to fail" in r.stderr
    assert time.time() - t0 < 150
