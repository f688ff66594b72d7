"""User-sharded (data-parallel) train step: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device; this is new functionality specified in SURVEY.md
§8e.  Rows (users) of the global batch are partitioned contiguously over ranks.  Two exchanges per
step, nothing else:

  1. all-reduce(sum) of 5 float64 loss scalars {sum D^2, sum (R-S)^2, sum R, sum R^2, count} after the
     three forwards: var(R) and both mse means of `score_matching_loss` (train_SDRM.py:196-198) are
     over the GLOBAL batch, and the gradient flows through them (Q6).
  2. all-reduce(sum) of the flat gradient [P] fp32 after the backward (default), or in two buckets - the first one
     (embedding + layer 0) exchanged while the upper layers' weight gradients are still being computed, the second right
     after: the hand-offs of that form cost ~40 us per step by themselves (tools/exchange_probe.py), so it pays only for
     gradients whose all-reduce takes longer than that; every rank then applies the identical Adam update
     (train_SDRM.py:337), so parameters stay replicated without a broadcast.

Randomness is Philox keyed by the GLOBAL row index, so G ranks draw exactly what one rank draws.
Sampling shards users with no communication at all.

Two drivers of the same step:
  * `ShardedTrainer`: the three phases of include/sdrm_hip.h with `torch.distributed` collectives between them.  It only
    needs an object with `train_forward / train_backward / adam_step`; the CPU tests drive it with an oracle-backed
    stand-in over gloo, and with the real Engine on a shared GPU when one is present.
  * `RcclTrainer`: `sdrm_train_step_sharded` - the collectives are issued by the library itself over RCCL (what a
    non-Python caller uses; one C call per step, no Python between the phases).  `torch.distributed` is used once, to
    ship the 128-byte RCCL unique id from rank 0."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous balanced partition: first (n_rows % world) ranks get one extra row."""
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


class ShardedTrainer:
    def __init__(self, engine, rank: int = 0, world: int = 1, group=None, device=None, n_params=None, overlap=False):
        self.engine, self.rank, self.world, self.group, self.overlap = engine, rank, world, group, overlap
        self.buckets = 2 if overlap else 1
        dev = device if device is not None else getattr(engine, "device", "cpu")
        P = n_params if n_params is not None else engine.P
        self.sums = torch.zeros(8, dtype=torch.float64, device=dev)
        self.grad = torch.zeros(P, dtype=torch.float32, device=dev)

    def train_step(self, x0_local, lr, row0=0, step=0, seed=0, nd=1.0, explicit=None):
        """x0_local: this rank's rows [rows, L].  explicit = (noise, t, keep) for this rank's rows,
        or None for Philox(seed, step, row0 + r, c).  Returns the device scalar holding the GLOBAL loss."""
        e = self.engine
        if self.world == 1:
            if explicit is None:
                return e.train_step(x0_local, lr, seed=seed, step=step, nd=nd)
            noise, t, keep = explicit
            return e.train_step(x0_local, lr, noise=noise, t=t, keep=keep, nd=nd)
        if explicit is None:
            e.train_forward(x0_local, seed=seed, step=step, nd=nd, row0=row0, sums=self.sums)
        else:
            noise, t, keep = explicit
            e.train_forward(x0_local, noise=noise, t=t, keep=keep, nd=nd, row0=row0, sums=self.sums)
        dist.all_reduce(self.sums, op=dist.ReduceOp.SUM, group=self.group)
        if self.overlap and hasattr(e, "train_backward_begin"):
            # bucketed exchange: the first bucket (embedding + layer 0; final once the layer-0 weight gradient and
            # the embedding backward are done) is all-reduced on the collective's own stream while the upper
            # layers' weight gradients (two thirds of the wgrad flops) are computed
            (o0, n0), (o1, n1) = e.grad_buckets()
            loss = e.train_backward_begin(sums=self.sums, grad=self.grad)
            work = dist.all_reduce(self.grad[o0:o0 + n0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            e.train_backward_finish(grad=self.grad)
            dist.all_reduce(self.grad[o1:o1 + n1], op=dist.ReduceOp.SUM, group=self.group)
            work.wait()
        else:
            loss = e.train_backward(sums=self.sums, grad=self.grad)
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)
        e.adam_step(lr, grad=self.grad)
        return loss


class RcclTrainer:
    """Same `train_step` signature as ShardedTrainer; the exchange lives in libsdrm_hip.so (sdrm_train_step_sharded)."""

    def __init__(self, engine, rank: int = 0, world: int = 1, unique_id: bytes | None = None, group=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.buckets = int(getattr(engine, "gradient_buckets", 1))   # what sdrm_train_step_sharded will do (1 unless set otherwise)
        if unique_id is None:
            box = [None]
            if rank == 0:
                try:   # a failure here must still reach the broadcast, or the other ranks wait for it for ever
                    box[0] = type(engine).comm_unique_id()
                except Exception as ex:
                    box[0] = ex
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            if isinstance(box[0], Exception):
                raise box[0]
            unique_id = box[0]
        engine.comm_init_rank(world, rank, unique_id)

    def train_step(self, x0_local, lr, row0=0, step=0, seed=0, nd=1.0, explicit=None):
        if explicit is None:
            return self.engine.train_step_sharded(x0_local, lr, row0=row0, seed=seed, step=step, nd=nd)
        noise, t, keep = explicit
        return self.engine.train_step_sharded(x0_local, lr, row0=row0, noise=noise, t=t, keep=keep, nd=nd)
