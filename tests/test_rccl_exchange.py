"""The exchange of the user-sharded train step as the library issues it: RCCL called from inside libsdrm_hip.so
(include/sdrm_hip.h: sdrm_comm_unique_id / sdrm_comm_init_rank / sdrm_allreduce_init / sdrm_train_step_sharded).

A one-GPU box can hold one RCCL rank per device, so these tests run a communicator of ONE rank: every RCCL call of the
step is really made (unique id, communicator, three all-reduces per step on two streams, the event hand-offs), only the
wire is trivial.  The arithmetic of N > 1 ranks is covered by tests/test_sharded_gloo.py (two processes) and
test_hip_parity.py::test_sharded_equals_single.  Needs a real MI355X: `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest
import torch

from sdrm_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("buckets", [1, 2])
@pytest.mark.parametrize("dims", [(340, 340, 78, 1, 1024), (40, 40, 93, 5, 107), (96, 72, 10, 0, 33),
                                  # the shard sizes of the 8192 batch that take the 48-row row-owned step (csrc/rows48.h): 2048 users
                                  # (4 GPUs) with two work-groups per row group, 4096 users (2 GPUs) with one
                                  (340, 340, 78, 1, 2048), (340, 340, 78, 1, 4096)])
def test_one_rank_rccl_step_equals_train_step(engine_cls, dims, buckets):
    """sdrm_train_step_sharded over a 1-rank RCCL communicator == sdrm_train_step, bit for bit, over three steps
    (loss, gradient, parameters, Adam moments), in both exchange forms: one all-reduce of the gradient after the backward
    (default), or the bucketed backward with the first bucket's all-reduce on the auxiliary stream - the in-place
    all-reduces on the internal gradient and the stream hand-offs change nothing."""
    L, W, T, H, B = dims
    # behind a row-owned forward the one-call backward takes the strip-owned weight gradients, the two-call (bucketed) one the tile
    # launches: the same sums in another order - there the two steps agree to fp32 summation order, not bit for bit
    exact = not (buckets == 2 and B > 1280)
    init = synth.flatten_params(synth.init_params(L, W, T, H, seed=61), H)
    x0 = synth.synth_latents(B, L, seed=62)
    a, b = engine_cls(L, W, T, H, B), engine_cls(L, W, T, H, B).debug_set(gradient_buckets=buckets)
    a.set_params(init); b.set_params(init)
    assert b.comm_info() == (0, -1)
    b.comm_init_rank(1, 0, engine_cls.comm_unique_id())
    assert b.comm_info() == (1, 0)
    for step in range(3):
        lr = 1e-3 * (1 - step / 3)
        la = float(a.train_step(x0, lr, seed=9, step=step, nd=0.9).cpu())
        lb = float(b.train_step_sharded(x0, lr, row0=0, seed=9, step=step, nd=0.9).cpu())
        assert la == lb, (step, la, lb)
        if exact:
            assert bool((a.get_grads() == b.get_grads()).all()), step
            assert bool((a.get_params() == b.get_params()).all()), step
        else:
            ga, gb = a.get_grads(), b.get_grads()
            assert float(((ga - gb).norm() / ga.norm()).cpu()) <= 1e-5, step
            pa, pb = a.get_params(), b.get_params()
            assert float(((pa - pb).norm() / pa.norm()).cpu()) <= 1e-5, step
    (ma, va, ta), (mb, vb, tb) = a.get_adam_state(), b.get_adam_state()
    assert ta == tb == 3
    if exact:
        assert bool((ma == mb).all()) and bool((va == vb).all())
    a.close(); b.close()


def test_adopted_communicator_and_errors(engine_cls):
    """sdrm_allreduce_init adopts a ncclComm_t made by the caller's own RCCL code (here: made through a first engine) and
    does not destroy it; a step without a communicator, or a second communicator on one engine, is an error."""
    from sdrm_amd.engine import SdrmError
    L, W, T, H, B = 48, 48, 7, 1, 20
    init = synth.flatten_params(synth.init_params(L, W, T, H, seed=63), H)
    x0 = synth.synth_latents(B, L, seed=64)
    e = engine_cls(L, W, T, H, B)
    e.set_params(init)
    with pytest.raises(SdrmError, match="no communicator"):
        e.train_step_sharded(x0, 1e-3, seed=1, step=0)
    e.comm_init_rank(1, 0, engine_cls.comm_unique_id())
    with pytest.raises(SdrmError, match="already has a communicator"):
        e.comm_init_rank(1, 0, engine_cls.comm_unique_id())
    loss = float(e.train_step_sharded(x0, 1e-3, seed=1, step=0).cpu())
    ref = engine_cls(L, W, T, H, B)
    ref.set_params(init)
    assert loss == float(ref.train_step(x0, 1e-3, seed=1, step=0).cpu())
    # a second engine adopts the first one's communicator (as a caller with its own RCCL code would hand one in)
    other = engine_cls(L, W, T, H, B)
    other.set_params(init)
    comm = e.lib.sdrm_debug_comm_handle(e._h)
    assert comm
    assert other.lib.sdrm_allreduce_init(other._h, C.c_void_p(comm), None) == 0 and other.comm_info() == (1, 0)
    assert float(other.train_step_sharded(x0, 1e-3, seed=1, step=0).cpu()) == loss
    assert bool((other.get_params() == e.get_params()).all())
    other.close()                                    # must NOT destroy the adopted communicator ...
    l2 = float(e.train_step_sharded(x0, 1e-3, seed=1, step=1).cpu())   # ... its owner still uses it
    assert l2 == float(ref.train_step(x0, 1e-3, seed=1, step=1).cpu())
    assert e.lib.sdrm_comm_destroy(e._h) == 0 and e.comm_info() == (0, -1)
    e.close(); ref.close()


def test_explicit_randoms_through_the_sharded_step(engine_cls):
    L, W, T, H, B = 100, 100, 12, 2, 64
    init = synth.flatten_params(synth.init_params(L, W, T, H, seed=65), H)
    x0 = synth.synth_latents(B, L, seed=66)
    eps, t, keep = synth.synth_train_randoms(B, L, T, 1.0, seed=67)
    a, b = engine_cls(L, W, T, H, B), engine_cls(L, W, T, H, B)
    a.set_params(init); b.set_params(init)
    b.comm_init_rank(1, 0, engine_cls.comm_unique_id())
    la = float(a.train_step(x0, 1e-3, noise=eps, t=t, keep=keep).cpu())
    lb = float(b.train_step_sharded(x0, 1e-3, noise=eps, t=t, keep=keep).cpu())
    assert la == lb and bool((a.get_params() == b.get_params()).all())
    a.close(); b.close()


# ---------------------------------------------------------------------------------------------------------------------
# Two REAL ranks over RCCL: needs two devices, so it skips itself on a one-GPU box and runs wherever a multi-GPU driver
# runs `pytest -m gpu` (the first N > 1 execution of sdrm_comm_init_rank / sdrm_train_step_sharded
# should be a test, not the scaling bench).
def _rccl_worker(rank, world, port, init_flat, x0, dims, q):
    import os

    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # carries the 128-byte unique id only
    from sdrm_amd.engine import Engine
    from sdrm_amd.parallel import RcclTrainer, shard_rows
    gl, gw, gt, gh, gb = dims
    eng = Engine(gl, gw, gt, gh, max_rows=gb)
    eng.set_params(init_flat)
    tr = RcclTrainer(eng, rank, world)
    info = eng.comm_info()
    r0, rows = shard_rows(gb, rank, world)
    losses = []
    for step in range(3):
        loss = tr.train_step(torch.from_numpy(x0[r0:r0 + rows]).cuda(), 2e-3 * (1 - step / 3), row0=r0, step=step, seed=4242, nd=0.9)
        losses.append(float(loss.cpu()))
    q.put((rank, info, eng.get_params().cpu().numpy(), losses))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dims", [(340, 340, 78, 1, 2048), (40, 40, 93, 5, 107)])
def test_two_rank_rccl_step_equals_one_rank(dims):
    """Two processes, two devices, the library's own communicator (sdrm_comm_unique_id -> sdrm_comm_init_rank) and
    sdrm_train_step_sharded: comm_info() == (2, rank) on both, the three-step trajectory equals the single-engine
    sdrm_train_step to 1e-5 (fp32 summation order of the two half-batches), and the replicas stay bit-identical."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import socket

    import torch.multiprocessing as mp

    from sdrm_amd.engine import Engine
    gl, gw, gt, gh, gb = dims
    init_flat = synth.flatten_params(synth.init_params(gl, gw, gt, gh, seed=13), gh)
    x0 = synth.synth_latents(gb, gl, seed=14)
    single = Engine(gl, gw, gt, gh, max_rows=gb)
    single.set_params(init_flat)
    ref_losses = [float(single.train_step(x0, 2e-3 * (1 - s / 3), seed=4242, step=s, nd=0.9).cpu()) for s in range(3)]
    ref = single.get_params().cpu().numpy()
    single.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, init_flat, x0, dims, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, info, flat, losses in results:
        assert tuple(info) == (2, rank)
        assert np.sqrt(((flat - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-5, rank
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    assert np.array_equal(results[0][2], results[1][2])
