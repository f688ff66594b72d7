"""Stand-in for sdrm_amd.engine.Engine used by `bench.py --stub-engine` (tests/test_bench_launch.py): the methods the bench
protocol calls, on CPU tensors, computing nothing.  It lets the self-launch path of `python bench.py --gpus N`, the rendezvous,
the barrier / MAX-over-ranks timing and the JSON line run on a box without a GPU; it measures nothing."""
import time
import types

import torch


class StubEngine:
    P = 16

    def __init__(self, L, W, T, H, max_rows):
        self.L, self.T, self.device = L, T, "cpu"
        self.lib = types.SimpleNamespace(sdrm_source_hash=lambda: b"stub-engine")
        self._left, self._launches = 0, 0

    def set_params(self, flat):
        pass

    def _work(self):
        self._launches += 1
        time.sleep(2e-4)

    # one-call step (N = 1) and the three phases ShardedTrainer drives (N > 1)
    def train_step(self, x0, lr, **kw):
        self._work()
        return torch.zeros(())

    def train_forward(self, x0, sums=None, **kw):
        self._work()
        if sums is not None:
            sums[:5] = 1.0

    def train_backward(self, sums=None, grad=None):
        if grad is not None:
            grad.fill_(1.0)
        return torch.zeros(())

    def adam_step(self, lr, grad=None):
        pass

    def sample_begin(self, n, **kw):
        self._left = self.T

    def sample_steps(self, k):
        self._work()
        self._left = max(0, self._left - k)
        return self._left

    def sample_end(self):
        self._left = 0
        return torch.zeros((1, self.L))

    def profile_begin(self, **kw):
        pass

    def profile_end(self):
        return {}

    def launch_count(self):
        return self._launches

    def comm_info(self):
        return 0, -1
