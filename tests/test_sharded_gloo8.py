"""The 8-rank job on CPU: 8 processes over gloo, the product's `ShardedTrainer` / `shard_rows` driving the
oracle-backed stand-in of tests/test_sharded_gloo.py - so that the first real 8-GPU run is not the first time anything here sees
eight ranks.  Global batch 8192 (even: 1024 users per rank, BASELINE config C4) and a ragged one (8187 = 8 * 1023 + 3), both
gradient-bucket forms: the three-step trajectory equals the single process to 1e-5 and the replicas stay bit-identical.  Sampling:
5429 users (5429 % 8 = 5) sharded with no communication: the ranks' rows, concatenated, ARE the single-process sample (Philox is
keyed by the global row).  The reference is single-device (new functionality)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import philox_ref as pr  # noqa: E402
from sdrm_amd import synth  # noqa: E402
from sdrm_amd.parallel import ShardedTrainer, shard_rows  # noqa: E402
from test_sharded_gloo import LR, ND, SEED, H, L, OraclePhases, T, W, _free_port  # noqa: E402

WORLD = 8
N_SAMPLE = 5429


def _worker(rank, port, init, x0, q, overlap, n_sample):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(1)
    eng = OraclePhases(init)
    tr = ShardedTrainer(eng, rank, WORLD, device="cpu", n_params=eng.P, overlap=overlap)
    B = x0.shape[0]
    r0, rows = shard_rows(B, rank, WORLD)
    losses = []
    for step in range(3):
        loss = tr.train_step(torch.from_numpy(x0[r0:r0 + rows]), LR * (1 - step / 3), row0=r0, step=step, seed=SEED, nd=ND)
        losses.append(float(loss))
    lat = None
    if n_sample:
        # this rank's users of the sampling call, no collective: randoms keyed by (seed, call id, GLOBAL row)
        s0, n_local = shard_rows(n_sample, rank, WORLD)
        xT, z, keep, _ = pr.sample_randoms(SEED, 3, s0, n_local, L, T, ND, False)
        lat = (s0, eng.o.sample(xT, z, keep).numpy())
    q.put((rank, eng.o.flat(eng.names), losses, (r0, rows), lat))
    dist.barrier()
    dist.destroy_process_group()


def _run(B, overlap, n_sample=0):
    init = synth.init_params(L, W, T, H, seed=13)
    x0 = synth.synth_latents(B, L, seed=14)
    single = OraclePhases(init)
    ref_losses = [float(single.train_step(torch.from_numpy(x0), LR * (1 - s / 3), seed=SEED, step=s, nd=ND)) for s in range(3)]
    ref = single.o.flat(single.names)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, init, x0, q, overlap, n_sample)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in results] == list(range(WORLD))
    # the row partition: contiguous, disjoint, complete, balanced to one row
    spans = [r[3] for r in results]
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] == B
    assert max(s[1] for s in spans) - min(s[1] for s in spans) <= 1
    for rank, flat, losses, _, _ in results:
        assert np.sqrt(((flat - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-5, rank
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
    for rank, flat, _, _, _ in results[1:]:
        assert np.array_equal(flat, results[0][1]), rank          # replicas stay bit-identical
    return single, results


@pytest.mark.timeout(400)
@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("B", [8192, 8187])
def test_eight_rank_gloo_matches_single_process(B, overlap):
    _run(B, overlap)


@pytest.mark.timeout(400)
def test_eight_rank_sampling_is_the_single_process_sample():
    single, results = _run(64, False, n_sample=N_SAMPLE)
    starts = [r[4][0] for r in results]
    sizes = [r[4][1].shape[0] for r in results]
    assert starts == list(np.cumsum([0] + sizes[:-1])) and sum(sizes) == N_SAMPLE and sorted(set(sizes)) == [678, 679]
    got = np.concatenate([r[4][1] for r in results])
    xT, z, keep, _ = pr.sample_randoms(SEED, 3, 0, N_SAMPLE, L, T, ND, False)
    # the replicas' parameters after the three steps are bit-identical (checked in _run), so any rank's net is THE net; the
    # single-process oracle has walked the same trajectory to 1e-5 - sample with rank 0's parameters through a fresh oracle
    o = OraclePhases(synth.init_params(L, W, T, H, seed=13))
    off = 0
    shapes = synth.param_shapes(L, W, T, H)
    for n in o.names:
        k = int(np.prod(shapes[n]))
        o.o.p[n] = torch.from_numpy(results[0][1][off:off + k].reshape(shapes[n]).copy())
        off += k
    want = o.o.sample(xT, z, keep).numpy()
    assert np.array_equal(got, want)                               # rows are independent: sharding changes no bit
