"""Python handle over the C ABI: owns one `sdrm_engine` and passes torch device pointers through.

torch is used only for device memory and streams (plumbing); every numeric step runs in
libsdrm_hip.so.  Reference lines are into /root/reference/train_SDRM.py."""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _lib, synth


class SdrmError(RuntimeError):
    pass


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ParamSpan:
    """What torch.as_tensor needs to wrap foreign device memory: the engine's flat parameter vector."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class Engine:
    """eps-predictor SDRM(N_ITEMS=L, EMB_DIM=T, LATENT_DIM=W, n_hidden_layers=H) (:86-95) with its
    Adam state (:309) and DDPM schedule (:296-303) resident on one MI355X."""

    def __init__(self, L, W, T, H, max_rows, device=None):
        if not torch.cuda.is_available():
            raise SdrmError("sdrm_amd needs a ROCm device: no GPU is visible and there is no CPU fallback")
        self.lib = _lib.load()
        self.L, self.W, self.T, self.H, self.max_rows = int(L), int(W), int(T), int(H), int(max_rows)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self._h = C.c_void_p()
        rc = self.lib.sdrm_create(self.L, self.W, self.T, self.H, self.max_rows, self.device.index, C.byref(self._h))
        if rc != 0:
            msg = self.lib.sdrm_last_error(self._h).decode() if self._h else ""
            if self._h:
                self.lib.sdrm_destroy(self._h)
                self._h = C.c_void_p()
            raise SdrmError(f"sdrm_create failed: {_lib.STATUS.get(rc, rc)} {msg}")
        self.P = int(self.lib.sdrm_param_count(self._h))
        assert self.P == synth.param_count(self.L, self.W, self.T, self.H)
        self._sums = torch.zeros(8, dtype=torch.float64, device=self.device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.gradient_buckets = 2 if os.environ.get("SDRM_AR_BUCKETS", "1").strip() == "2" else 1   # of train_step_sharded
        self._keepalive = None
        self._n_views = 0                 # spans handed out by params_view() that some tensor's storage still holds
        self._close_pending = False

    # ------------------------------------------------------------------ plumbing
    def close(self):
        """Frees the engine - unless tensors returned by `params_view()` (an SDRM's `parameters()`) are still alive: they alias the
        master vector sdrm_destroy would free, so the handle then stays allocated (and unused) until the last of them is gone.
        Such a tensor of a closed engine keeps reading the parameters as they were at close()."""
        if not getattr(self, "_h", None):
            return
        if self._n_views > 0:
            self._close_pending = True
            return
        torch.cuda.synchronize(self.device)
        self.lib.sdrm_destroy(self._h)
        self._h = C.c_void_p()
        self._close_pending = False

    @property
    def closed(self):
        return not getattr(self, "_h", None)

    @staticmethod
    def _view_released(eng):
        eng._n_views -= 1
        if eng._close_pending and eng._n_views <= 0:
            eng.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SdrmError(f"{what}: {_lib.STATUS.get(rc, rc)}: {self.lib.sdrm_last_error(self._h).decode()}")

    def _dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=dtype)
        else:
            t = torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype)
        return t.contiguous()

    def debug_set(self, tile=None, skinny=None, fused_reverse=None, chains=None, nt32_rows=None, nt32_rows_train=None,
                  gradient_buckets=None, rowchain=None, wgrad_strips=None, dgrad_rows=None, rows48=None, rows48_split=None, sample_persist=None, rows48_share=None):
        """Test / tuning hooks of THIS engine (include/sdrm_hip_debug.h): force a GEMM tile shape (-1 = automatic),
        switch the narrow-net kernels, the fused reverse update, the sampler row chains, the 32x32-tile row thresholds,
        the number of gradient all-reduces of the sharded step (1 or 2), the row-owned train forward (0 never, 1 by
        size, 2 whenever the net allows), the strip-owned weight gradients and the row-owned input gradients behind it."""
        if rowchain is not None:
            self._check(self.lib.sdrm_debug_set_rowchain(self._h, int(rowchain)), "sdrm_debug_set_rowchain")
        if rows48_share is not None:   # the shared-tile form of the 48-row kernels (1 on, 0 the plain form)
            self._check(self.lib.sdrm_debug_set_rows48_share(self._h, int(rows48_share)), "sdrm_debug_set_rows48_share")
        if sample_persist is not None:   # reverse steps in one launch (csrc/sample_persist.h): 0 never, 1 by size, 2 whenever it fits
            self._check(self.lib.sdrm_debug_set_sample_persist(self._h, int(sample_persist)), "sdrm_debug_set_sample_persist")
        if rows48_split is not None:   # column-split row groups of that step: 0 never, 1 by size, 2 / 4 work-groups per group
            self._check(self.lib.sdrm_debug_set_rows48_split(self._h, int(rows48_split)), "sdrm_debug_set_rows48_split")
        if rows48 is not None:   # the same step on 48-row work-groups (csrc/rows48.h): 0 never, 1 by size, 2 whenever the net allows
            self._check(self.lib.sdrm_debug_set_rows48(self._h, int(rows48)), "sdrm_debug_set_rows48")
        if wgrad_strips is not None:
            self._check(self.lib.sdrm_debug_set_wgrad_strips(self._h, int(bool(wgrad_strips))), "sdrm_debug_set_wgrad_strips")
        if dgrad_rows is not None:
            self._check(self.lib.sdrm_debug_set_dgrad_rows(self._h, int(dgrad_rows)), "sdrm_debug_set_dgrad_rows")
        if gradient_buckets is not None:
            self._check(self.lib.sdrm_debug_set_gradient_buckets(self._h, int(gradient_buckets)), "sdrm_debug_set_gradient_buckets")
            self.gradient_buckets = int(gradient_buckets)
        if tile is not None:
            self._check(self.lib.sdrm_debug_set_tile(self._h, int(tile)), "sdrm_debug_set_tile")
        if skinny is not None:
            self._check(self.lib.sdrm_debug_set_skinny(self._h, int(skinny)), "sdrm_debug_set_skinny")
        if fused_reverse is not None:
            self._check(self.lib.sdrm_debug_set_fused_reverse(self._h, int(fused_reverse)), "sdrm_debug_set_fused_reverse")
        if chains is not None:
            self._check(self.lib.sdrm_debug_set_chains(self._h, int(chains)), "sdrm_debug_set_chains")
        if nt32_rows is not None or nt32_rows_train is not None:
            self._check(self.lib.sdrm_debug_set_nt32_rows(self._h, -1 if nt32_rows is None else int(nt32_rows),
                                                          -1 if nt32_rows_train is None else int(nt32_rows_train)),
                        "sdrm_debug_set_nt32_rows")
        return self

    @property
    def sampler_chains(self):
        """Row chains of the sampling call in progress / of the last one (csrc/sdrm_hip.hip: chains_for)."""
        return int(self.lib.sdrm_debug_chains(self._h))

    @property
    def rows48_split_available(self):
        """True when column-split row groups (csrc/rows48.h) may be taken: the net qualifies and the chip maps block b to XCD b & 7."""
        return bool(self.lib.sdrm_debug_rows48_split_available(self._h))

    @property
    def rowchain_available(self):
        """True when this engine's shape qualifies for the row-owned train forward (csrc/rowchain.h)."""
        return bool(self.lib.sdrm_debug_rowchain_available(self._h))

    # ------------------------------------------------------------------ parameters
    def set_params(self, flat):
        flat = self._dev(flat, torch.float32).reshape(-1)
        if flat.numel() != self.P:
            raise SdrmError(f"set_params: expected {self.P} floats, got {flat.numel()}")
        self._check(self.lib.sdrm_set_params(self._h, _ptr(flat), _stream()), "sdrm_set_params")
        self._keepalive = flat

    def _flat_out(self, fn, name):
        out = torch.empty(self.P, dtype=torch.float32, device=self.device)
        self._check(fn(self._h, _ptr(out), _stream()), name)
        return out

    def get_params(self):
        return self._flat_out(self.lib.sdrm_get_params, "sdrm_get_params")

    def params_view(self):
        """The live parameters in place (sdrm_params_ptr): a [P] float32 tensor that ALIASES the engine's master vector - it follows
        every train step without a copy.  Read it; do not write through it (the kernels read compute copies of it)."""
        span = _ParamSpan(int(self.lib.sdrm_params_ptr(self._h)), self.P)
        # torch keeps `span` for the lifetime of the storage it wraps (every view / slice / detach() of the tensor shares that
        # storage); the finalizer holds the engine until the span dies, and close() defers sdrm_destroy while any span is alive
        self._n_views += 1
        weakref.finalize(span, Engine._view_released, self).atexit = False
        return torch.as_tensor(span, device=self.device)

    def philox_draws(self, seed, purpose, step, rows, quads, row0=0, with_bits=True):
        """Test hook (sdrm_debug_philox_draws): the device generator's normals [rows, 4 * quads] and the low three bits of its
        words [rows, 4 * quads] uint8 for (seed, purpose, step)."""
        normals = torch.empty(rows, 4 * quads, dtype=torch.float32, device=self.device)
        bits = torch.empty(rows, 4 * quads, dtype=torch.uint8, device=self.device) if with_bits else None
        self._check(self.lib.sdrm_debug_philox_draws(self._h, int(seed), int(purpose), int(step), int(row0), int(rows), int(quads),
                                                     _ptr(normals), _ptr(bits), _stream()), "sdrm_debug_philox_draws")
        return normals, bits

    def get_grads(self):
        return self._flat_out(self.lib.sdrm_get_grads, "sdrm_get_grads")

    def get_adam_state(self):
        m = torch.empty(self.P, dtype=torch.float32, device=self.device)
        v = torch.empty(self.P, dtype=torch.float32, device=self.device)
        step = C.c_int64()
        self._check(self.lib.sdrm_get_adam_state(self._h, _ptr(m), _ptr(v), C.byref(step), _stream()), "sdrm_get_adam_state")
        return m, v, int(step.value)

    def set_adam_state(self, m, v, step):
        m, v = self._dev(m, torch.float32), self._dev(v, torch.float32)
        self._check(self.lib.sdrm_set_adam_state(self._h, _ptr(m), _ptr(v), int(step), _stream()), "sdrm_set_adam_state")
        self._keepalive = (m, v)

    def adam_reset(self):
        self._check(self.lib.sdrm_adam_reset(self._h, _stream()), "sdrm_adam_reset")

    def set_schedule(self, beta1=1e-4, beta2=0.02):
        self._check(self.lib.sdrm_set_schedule(self._h, beta1, beta2), "sdrm_set_schedule")

    def get_schedule(self):
        n = self.T + 1
        b, a, ab = (np.empty(n, np.float32) for _ in range(3))
        self._check(self.lib.sdrm_get_schedule(self._h, b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                               ab.ctypes.data_as(C.c_void_p)), "sdrm_get_schedule")
        return b, a, ab

    # ------------------------------------------------------------------ training
    def _randoms(self, noise, t, keep, B):
        noise = self._dev(noise, torch.float32)
        t = self._dev(t, torch.int64)
        keep = self._dev(keep, torch.uint8)
        if tuple(noise.shape) != (B, self.L) or tuple(t.shape) != (B,) or tuple(keep.shape) != (3, B, self.L):
            raise SdrmError("explicit randoms: expected noise [B,L], t [B], keep [3,B,L]")
        self._keepalive = (noise, t, keep)
        return _lib.TrainRandoms(noise.data_ptr(), t.data_ptr(), keep.data_ptr())

    def train_forward(self, x0, noise=None, t=None, keep=None, seed=0, step=0, nd=1.0, row0=0, sums=None):
        """Phase 1 (:326-333 up to the loss sums).  Explicit randoms if `noise` is given, else Philox."""
        x0 = self._dev(x0, torch.float32)
        B = x0.shape[0]
        if x0.dim() != 2 or x0.shape[1] != self.L:
            raise SdrmError(f"train_forward: x0 must be [B,{self.L}]")
        self._x0 = x0
        sums = self._sums if sums is None else sums
        if noise is not None:
            rnd = self._randoms(noise, t, keep, B)
            rc = self.lib.sdrm_train_forward(self._h, _ptr(x0), B, int(row0), _lib.RNG_EXPLICIT, C.byref(rnd), 0, 0,
                                             float(nd), _ptr(sums), _stream())
        else:
            rc = self.lib.sdrm_train_forward(self._h, _ptr(x0), B, int(row0), _lib.RNG_PHILOX, None, int(seed),
                                             int(step), float(nd), _ptr(sums), _stream())
        self._check(rc, "sdrm_train_forward")
        return sums

    def train_backward(self, sums=None, grad=None):
        """Phase 2: seeds from the (global) sums, backward, flat gradient.  Returns the device loss scalar."""
        sums = self._sums if sums is None else sums
        self._check(self.lib.sdrm_train_backward(self._h, _ptr(sums), _ptr(grad), _ptr(self._loss), _stream()),
                    "sdrm_train_backward")
        return self._loss

    def train_backward_begin(self, sums=None, grad=None):
        """Phase 2, first call: on return the FIRST bucket of `grad` (see `grad_buckets`) is final in stream order;
        the upper layers' weight gradients are left to `train_backward_finish`."""
        sums = self._sums if sums is None else sums
        self._check(self.lib.sdrm_train_backward_begin(self._h, _ptr(sums), _ptr(grad), _ptr(self._loss), _stream()),
                    "sdrm_train_backward_begin")
        return self._loss

    def train_backward_finish(self, grad=None):
        """Phase 2, second call: upper-layer weight gradients, then the SECOND bucket is final."""
        self._check(self.lib.sdrm_train_backward_finish(self._h, _ptr(grad), _stream()), "sdrm_train_backward_finish")

    def grad_buckets(self):
        """((offset, length) of the first bucket, (offset, length) of the second) in the flat gradient."""
        v = [C.c_int64() for _ in range(4)]
        self._check(self.lib.sdrm_grad_buckets(self._h, *[C.byref(x) for x in v]), "sdrm_grad_buckets")
        return (int(v[0].value), int(v[1].value)), (int(v[2].value), int(v[3].value))

    def adam_step(self, lr, grad=None):
        """Phase 3: coupled-L2 Adam (:309,:337) at the caller's per-epoch lr (:316)."""
        self._check(self.lib.sdrm_adam_step(self._h, _ptr(grad), float(lr), _stream()), "sdrm_adam_step")

    def train_step(self, x0, lr, noise=None, t=None, keep=None, seed=0, step=0, nd=1.0):
        """One whole step (:326-337) on this GPU; returns the device loss scalar (no host sync, Q14)."""
        x0 = self._dev(x0, torch.float32)
        B = x0.shape[0]
        if x0.dim() != 2 or x0.shape[1] != self.L:
            raise SdrmError(f"train_step: x0 must be [B,{self.L}]")
        self._x0 = x0
        if noise is not None:
            rnd = self._randoms(noise, t, keep, B)
            rc = self.lib.sdrm_train_step(self._h, _ptr(x0), B, float(lr), _lib.RNG_EXPLICIT, C.byref(rnd), 0, 0,
                                          float(nd), _ptr(self._loss), _stream())
        else:
            rc = self.lib.sdrm_train_step(self._h, _ptr(x0), B, float(lr), _lib.RNG_PHILOX, None, int(seed), int(step),
                                          float(nd), _ptr(self._loss), _stream())
        self._check(rc, "sdrm_train_step")
        return self._loss

    # ------------------------------------------------------------------ multi-GPU exchange inside the library
    @staticmethod
    def comm_available() -> bool:
        """True when librccl resolves in this process (a local check: agree on it across ranks BEFORE comm_init_rank)."""
        return bool(_lib.load().sdrm_comm_available())

    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL unique id (rank 0 calls this and ships the bytes to the other ranks by any channel)."""
        buf = C.create_string_buffer(128)
        rc = _lib.load().sdrm_comm_unique_id(buf)
        if rc != 0:
            raise SdrmError(f"sdrm_comm_unique_id: {_lib.STATUS.get(rc, rc)} (librccl could not be loaded?)")
        return buf.raw

    def comm_init_rank(self, nranks: int, rank: int, unique_id: bytes):
        """Joins the RCCL communicator of the user-sharded step (the library owns it)."""
        if len(unique_id) != 128:
            raise SdrmError("comm_init_rank: the unique id is 128 bytes")
        self._check(self.lib.sdrm_comm_init_rank(self._h, int(nranks), int(rank), C.c_char_p(unique_id)), "sdrm_comm_init_rank")
        return self

    def comm_info(self):
        n, r = C.c_int(), C.c_int()
        self._check(self.lib.sdrm_comm_info(self._h, C.byref(n), C.byref(r)), "sdrm_comm_info")
        return int(n.value), int(r.value)

    def train_step_sharded(self, x0, lr, row0=0, noise=None, t=None, keep=None, seed=0, step=0, nd=1.0):
        """One step on this rank's rows with both exchanges (loss sums, gradient buckets) issued by the library over
        RCCL; returns the device scalar holding the GLOBAL loss."""
        x0 = self._dev(x0, torch.float32)
        B = x0.shape[0]
        if x0.dim() != 2 or x0.shape[1] != self.L:
            raise SdrmError(f"train_step_sharded: x0 must be [B,{self.L}]")
        self._x0 = x0
        if noise is not None:
            rnd = self._randoms(noise, t, keep, B)
            rc = self.lib.sdrm_train_step_sharded(self._h, _ptr(x0), B, int(row0), float(lr), _lib.RNG_EXPLICIT, C.byref(rnd),
                                                  0, 0, float(nd), _ptr(self._loss), _stream())
        else:
            rc = self.lib.sdrm_train_step_sharded(self._h, _ptr(x0), B, int(row0), float(lr), _lib.RNG_PHILOX, None, int(seed),
                                                  int(step), float(nd), _ptr(self._loss), _stream())
        self._check(rc, "sdrm_train_step_sharded")
        return self._loss

    def train_outputs(self, B):
        out = torch.empty(3, B, self.L, dtype=torch.float32, device=self.device)
        self._check(self.lib.sdrm_get_train_outputs(self._h, _ptr(out), _stream()), "sdrm_get_train_outputs")
        return out

    def preacts(self, layer, B):
        """Pre-activations [3,B,W] of layer `layer` from the last train forward (parity tests)."""
        out = torch.empty(3, B, self.W, dtype=torch.float32, device=self.device)
        self._check(self.lib.sdrm_get_preacts(self._h, int(layer), _ptr(out), _stream()), "sdrm_get_preacts")
        return out

    def launch_count(self):
        """Kernel launches issued through this handle so far (bench.py: launches per step)."""
        return int(self.lib.sdrm_launch_count(self._h))

    # ------------------------------------------------------------------ profiling (bench only)
    def profile_begin(self, capacity=4096, only=None):
        """`only`: name of the one kernel class to bracket (as `profile_end` returns them); None = every GEMM launch."""
        cls = -1
        if only is not None:
            names = [self.lib.sdrm_profile_name(c).decode() for c in range(self.lib.sdrm_profile_classes())]
            cls = names.index(only)
        self._check(self.lib.sdrm_profile_only(self._h, cls), "sdrm_profile_only")
        self._check(self.lib.sdrm_profile_begin(self._h, int(capacity)), "sdrm_profile_begin")

    def profile_end(self):
        """Returns {kernel class name: (total_ms, launches, algorithmic_flops)} for classes that ran."""
        self._check(self.lib.sdrm_profile_end(self._h, _stream()), "sdrm_profile_end")
        out = {}
        for c in range(self.lib.sdrm_profile_classes()):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            self._check(self.lib.sdrm_profile_get(self._h, c, C.byref(ms), C.byref(n), C.byref(fl)), "sdrm_profile_get")
            if n.value:
                out[self.lib.sdrm_profile_name(c).decode()] = (ms.value, int(n.value), fl.value)
        return out

    # ------------------------------------------------------------------ inference
    def forward(self, x, t, keep=None, seed=0, step=0, row0=0):
        """SDRM.forward(x, t) (:97-103); dropout is always on (Q2)."""
        x = self._dev(x, torch.float32)
        t = self._dev(t, torch.int64)
        n = x.shape[0]
        out = torch.empty(n, self.L, dtype=torch.float32, device=self.device)
        if keep is not None:
            keep = self._dev(keep, torch.uint8)
            rc = self.lib.sdrm_forward(self._h, _ptr(x), _ptr(t), n, _lib.RNG_EXPLICIT, _ptr(keep), 0, 0, 0, _ptr(out),
                                       _stream())
        else:
            rc = self.lib.sdrm_forward(self._h, _ptr(x), _ptr(t), n, _lib.RNG_PHILOX, None, int(seed), int(step),
                                       int(row0), _ptr(out), _stream())
        self._check(rc, "sdrm_forward")
        self._keepalive = (x, t, keep)
        return out

    def sample(self, n, nd=1.0, multires=False, xT=None, z=None, keep=None, Tj=None, seed=0, call_id=0, row0=0,
               return_Tj=False):
        """Latent part of sample_ddpm (:37-59): returns x_0 latents [n,L] (caller applies vae.decode)."""
        out = torch.empty(n, self.L, dtype=torch.float32, device=self.device)
        tj_out = torch.zeros(n, dtype=torch.int64, device=self.device) if (multires and return_Tj) else None
        if xT is not None:
            xT, z, keep = self._dev(xT, torch.float32), self._dev(z, torch.float32), self._dev(keep, torch.uint8)
            Tj = None if Tj is None else self._dev(Tj, torch.int64)
            if tuple(z.shape) != (self.T + 1, n, self.L) or tuple(keep.shape) != (self.T + 1, n, self.L):
                raise SdrmError("sample: z and keep must be [T+1,n,L]")
            rc = self.lib.sdrm_sample(self._h, n, float(nd), int(bool(multires)), _lib.RNG_EXPLICIT, _ptr(xT), _ptr(z),
                                      _ptr(keep), _ptr(Tj), 0, 0, 0, _ptr(out), _ptr(tj_out), _stream())
        else:
            rc = self.lib.sdrm_sample(self._h, n, float(nd), int(bool(multires)), _lib.RNG_PHILOX, None, None, None,
                                      None, int(seed), int(call_id), int(row0), _ptr(out), _ptr(tj_out), _stream())
        self._check(rc, "sdrm_sample")
        self._keepalive = (xT, z, keep, Tj)
        return (out, tj_out) if return_Tj else out

    def sample_begin(self, n, nd=1.0, multires=False, seed=0, call_id=0, row0=0):
        """Resumable PHILOX-mode sampler (bench.py interleaves its steps with train steps)."""
        self._check(self.lib.sdrm_sample_begin(self._h, int(n), float(nd), int(bool(multires)), _lib.RNG_PHILOX, None,
                                               None, None, None, int(seed), int(call_id), int(row0), None, _stream()),
                    "sdrm_sample_begin")
        self._sample_n = int(n)

    def sample_steps(self, count):
        self._check(self.lib.sdrm_sample_steps(self._h, int(count), _stream()), "sdrm_sample_steps")
        return int(self.lib.sdrm_sample_remaining(self._h))

    def sample_end(self):
        out = torch.empty(self._sample_n, self.L, dtype=torch.float32, device=self.device)
        self._check(self.lib.sdrm_sample_end(self._h, _ptr(out), _stream()), "sdrm_sample_end")
        return out

    def reverse_step(self, x, i, z, keep):
        x = self._dev(x, torch.float32).clone()
        z = None if z is None else self._dev(z, torch.float32)
        keep = self._dev(keep, torch.uint8)
        self._check(self.lib.sdrm_reverse_step(self._h, _ptr(x), x.shape[0], int(i), _ptr(z), _ptr(keep), _stream()),
                    "sdrm_reverse_step")
        self._keepalive = (z, keep)
        return x

    def equal_sparsity(self, raw, sparsity, return_threshold=False):
        """main.py:177-180 on the device: `(raw >= np.quantile(raw.flatten(), sparsity))` as a uint8 tensor of raw's shape
        (and the float32 threshold np.quantile returns, as a 0-d device tensor, when asked)."""
        raw = self._dev(raw, torch.float32)
        out = torch.empty(raw.shape, dtype=torch.uint8, device=self.device)
        thr = torch.empty((), dtype=torch.float32, device=self.device)
        self._check(self.lib.sdrm_equal_sparsity(self._h, _ptr(raw), raw.numel(), float(sparsity), _ptr(out), _ptr(thr),
                                                 _stream()), "sdrm_equal_sparsity")
        return (out, thr) if return_threshold else out

    # ------------------------------------------------------------------ VAE decode on the engine
    def _decoder(self, w1, b1, w2, b2):
        w1, b1, w2, b2 = (self._dev(t.detach() if isinstance(t, torch.Tensor) else t, torch.float32) for t in (w1, b1, w2, b2))
        hidden, latent = w1.shape
        n_items = w2.shape[0]
        if tuple(b1.shape) != (hidden,) or tuple(w2.shape) != (n_items, hidden) or tuple(b2.shape) != (n_items,):
            raise SdrmError("vae_decode: expected decoder[0].weight [hidden, latent], .bias [hidden], decoder[2].weight [items, hidden], .bias [items]")
        self._keepalive = (w1, b1, w2, b2)
        return _lib.VaeDecoder(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), latent, hidden, n_items), latent, n_items

    def vae_decode(self, z, w1, b1, w2, b2):
        """`VAE.decode(z)` (train_SDRM.py:252-254) for decoder = Linear -> Tanh -> Linear given as its four tensors."""
        dec, latent, n_items = self._decoder(w1, b1, w2, b2)
        z = self._dev(z, torch.float32)
        if z.dim() != 2 or z.shape[1] != latent:
            raise SdrmError(f"vae_decode: z must be [n,{latent}]")
        out = torch.empty(z.shape[0], n_items, dtype=torch.float32, device=self.device)
        self._check(self.lib.sdrm_vae_decode(self._h, C.byref(dec), _ptr(z), z.shape[0], _ptr(out), _stream()), "sdrm_vae_decode")
        return out

    def csr_to_device(self, m):
        """(indptr i64, indices i32, data f32 | None for an all-ones matrix, shape) of a scipy sparse matrix, on the device."""
        m = m.tocsr().copy()
        m.sum_duplicates()
        m.sort_indices()
        data = None if np.all(m.data == 1) else torch.from_numpy(m.data.astype(np.float32)).to(self.device)
        return (torch.from_numpy(m.indptr.astype(np.int64)).to(self.device),
                torch.from_numpy(m.indices.astype(np.int32)).to(self.device), data, m.shape)

    def csr_rows_to_dense(self, csr_dev, rows=None, row0=0, b=None, check=True):
        """dataloaders.py:46-79 + `.to_dense()` (train_SDRM.py:323) on the device: dense float32 [b, n_items] of the rows
        `rows` (int64 tensor, e.g. a slice of the epoch permutation) or row0 .. row0+b-1 of a `csr_to_device` matrix.
        Row ids and column indices are range-checked ON THE DEVICE (an offending row / entry stays zero, never a stray store);
        `check=True` reads the verdict back at once (one stream sync) and raises, `check=False` leaves it to a later
        `feed_status()` - what an epoch loop wants (`pipeline.DeviceFeed` asks once per epoch)."""
        indptr, indices, data, (n_rows, n_items) = csr_dev
        if rows is not None:
            rows = self._dev(rows, torch.int64)
            b = rows.numel()
        elif b is None:
            raise SdrmError("csr_rows_to_dense: give `rows` or `row0` and `b`")
        out = torch.empty(b, n_items, dtype=torch.float32, device=self.device)
        self._check(self.lib.sdrm_csr_rows_to_dense(self._h, _ptr(indptr), _ptr(indices), _ptr(data), int(n_rows), _ptr(rows), int(row0),
                                                    int(b), int(n_items), _ptr(out), _stream()), "sdrm_csr_rows_to_dense")
        if check:
            self.feed_status()
        return out

    def feed_status(self):
        """Raises if any `csr_rows_to_dense` launch since the last call met a row id / column index outside the matrix."""
        self._check(self.lib.sdrm_feed_status(self._h, _stream()), "sdrm_feed_status")

    def rank_metrics(self, scores, heldout, train=None, ks=(1, 3, 5, 10, 20, 50)):
        """utilities.py:116-171 on the device: (recall[nk,U], ndcg[nk,U]) float64 device tensors for a score matrix
        [U, I] (device or host) against the held-out CSR matrix, with the items of the `train` CSR matrix masked out
        (-inf).  `heldout` / `train` are scipy.sparse matrices (or anything with tocsr())."""
        scores = self._dev(scores, torch.float32)
        U, I = scores.shape
        ks = np.asarray(ks, dtype=np.int32)
        kmax = int(ks.max())
        tp = 1.0 / np.log2(np.arange(2, kmax + 2))                                   # utilities.py:145
        idcg = np.asarray([tp[:m].sum() for m in range(kmax + 1)], dtype=np.float64)   # utilities.py:149-150

        def csr(m):
            m = m.tocsr()
            if m.shape != (U, I):
                raise SdrmError(f"rank_metrics: sparse matrix shape {m.shape} != scores shape {(U, I)}")
            return (torch.from_numpy(m.indptr.astype(np.int64)).to(self.device),
                    torch.from_numpy(m.indices.astype(np.int32)).to(self.device))
        hp, hi = csr(heldout)
        tr = csr(train) if train is not None else (None, None)
        tp_d, idcg_d = torch.from_numpy(tp).to(self.device), torch.from_numpy(idcg).to(self.device)
        recall = torch.empty(len(ks), U, dtype=torch.float64, device=self.device)
        ndcg = torch.empty(len(ks), U, dtype=torch.float64, device=self.device)
        self._check(self.lib.sdrm_rank_metrics(self._h, _ptr(scores), U, I, _ptr(hp), _ptr(hi), _ptr(tr[0]), _ptr(tr[1]),
                                               ks.ctypes.data_as(C.c_void_p), len(ks), _ptr(tp_d), _ptr(idcg_d),
                                               _ptr(recall), _ptr(ndcg), _stream()), "sdrm_rank_metrics")
        self._keepalive = (scores, hp, hi, tr, tp_d, idcg_d)
        return recall, ndcg

    def perturb_input(self, x, t, noise):
        x, t, noise = self._dev(x, torch.float32), self._dev(t, torch.int64), self._dev(noise, torch.float32)
        out = torch.empty_like(x)
        self._check(self.lib.sdrm_perturb_input(self._h, _ptr(x), _ptr(t), _ptr(noise), x.shape[0], _ptr(out), _stream()),
                    "sdrm_perturb_input")
        return out


_UTILITY = {}


def utility_engine(device=None) -> Engine:
    """A small cached engine for the handle-independent device ops (equal_sparsity, rank_metrics, csr_rows_to_dense):
    the C ABI hangs error strings and the select workspace on a handle, nothing of the eps-net is used."""
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
    eng = _UTILITY.get(idx)
    if eng is None or not eng._h:
        eng = _UTILITY[idx] = Engine(8, 8, 4, 0, 16, device=idx)
    return eng
