"""ctypes binding of libsdrm_hip.so (C ABI in include/sdrm_hip.h).

There is NO fallback: if the library cannot be built or loaded, `load()` raises — the product path
never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

from . import _build

_LIB = None

c_void_p, c_int, c_int64, c_uint64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float


class TrainRandoms(C.Structure):
    _fields_ = [("noise", c_void_p), ("t", c_void_p), ("keep", c_void_p)]


class VaeDecoder(C.Structure):
    _fields_ = [("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("latent", c_int), ("hidden", c_int),
                ("n_items", c_int)]


# name -> (restype, argtypes); mirrors include/sdrm_hip.h one to one
SIGNATURES = {
    "sdrm_create": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_void_p)]),
    "sdrm_destroy": (c_int, [c_void_p]),
    "sdrm_last_error": (C.c_char_p, [c_void_p]),
    "sdrm_param_count": (c_int64, [c_void_p]),
    "sdrm_set_schedule": (c_int, [c_void_p, c_float, c_float]),
    "sdrm_get_schedule": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdrm_set_params": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_get_params": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_params_ptr": (c_void_p, [c_void_p]),
    "sdrm_get_grads": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_get_adam_state": (c_int, [c_void_p, c_void_p, c_void_p, C.POINTER(c_int64), c_void_p]),
    "sdrm_set_adam_state": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sdrm_adam_reset": (c_int, [c_void_p, c_void_p]),
    "sdrm_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, C.POINTER(TrainRandoms), c_uint64,
                                   c_uint64, c_float, c_void_p, c_void_p]),
    "sdrm_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdrm_train_backward_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdrm_train_backward_finish": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_grad_buckets": (c_int, [c_void_p, C.POINTER(c_int64), C.POINTER(c_int64), C.POINTER(c_int64), C.POINTER(c_int64)]),
    "sdrm_adam_step": (c_int, [c_void_p, c_void_p, c_float, c_void_p]),
    "sdrm_train_step": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, C.POINTER(TrainRandoms), c_uint64, c_uint64,
                                c_float, c_void_p, c_void_p]),
    "sdrm_comm_unique_id": (c_int, [c_void_p]),
    "sdrm_comm_available": (c_int, []),
    "sdrm_comm_init_rank": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "sdrm_allreduce_init": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_comm_info": (c_int, [c_void_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "sdrm_comm_destroy": (c_int, [c_void_p]),
    "sdrm_train_step_sharded": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_float, c_int, C.POINTER(TrainRandoms), c_uint64,
                                        c_uint64, c_float, c_void_p, c_void_p]),
    "sdrm_get_train_outputs": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_uint64, c_uint64, c_int64,
                             c_void_p, c_void_p]),
    "sdrm_sample": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                            c_uint64, c_int64, c_void_p, c_void_p, c_void_p]),
    "sdrm_sample_begin": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_uint64, c_uint64, c_int64, c_void_p, c_void_p]),
    "sdrm_sample_steps": (c_int, [c_void_p, c_int, c_void_p]),
    "sdrm_sample_remaining": (c_int, [c_void_p]),
    "sdrm_sample_end": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sdrm_reverse_step": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sdrm_perturb_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sdrm_get_preacts": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "sdrm_profile_begin": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_gradient_buckets": (c_int, [c_void_p, c_int]),
    "sdrm_profile_only": (c_int, [c_void_p, c_int]),
    "sdrm_profile_end": (c_int, [c_void_p, c_void_p]),
    "sdrm_profile_classes": (c_int, []),
    "sdrm_profile_name": (C.c_char_p, [c_int]),
    "sdrm_profile_get": (c_int, [c_void_p, c_int, C.POINTER(C.c_double), C.POINTER(c_int64), C.POINTER(C.c_double)]),
    "sdrm_launch_count": (c_int64, [c_void_p]),
    "sdrm_build_info": (C.c_char_p, []),
    "sdrm_csr_rows_to_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sdrm_feed_status": (c_int, [c_void_p, c_void_p]),
    "sdrm_vae_decode": (c_int, [c_void_p, C.POINTER(VaeDecoder), c_void_p, c_int, c_void_p, c_void_p]),
    "sdrm_equal_sparsity": (c_int, [c_void_p, c_void_p, c_int64, C.c_double, c_void_p, c_void_p, c_void_p]),
    "sdrm_rank_metrics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdrm_source_hash": (C.c_char_p, []),
    # include/sdrm_hip_debug.h (test / tuning hooks; every setter acts on one handle)
    "sdrm_debug_philox_draws": (c_int, [c_void_p, c_uint64, C.c_uint32, C.c_uint32, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sdrm_debug_device_check": (c_int, [C.c_char_p, c_int]),
    "sdrm_debug_set_tile": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_nt32_rows": (c_int, [c_void_p, c_int, c_int]),
    "sdrm_debug_set_chains": (c_int, [c_void_p, c_int]),
    "sdrm_debug_chains": (c_int, [c_void_p]),
    "sdrm_debug_set_fused_reverse": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_skinny": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_rowchain": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_rows48": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_rows48_split": (c_int, [c_void_p, c_int]),
    "sdrm_debug_rows48_split_available": (c_int, [c_void_p]),
    "sdrm_debug_split_skew": (c_int, [c_void_p, C.c_uint32]),
    "sdrm_debug_set_sample_persist": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_rows48_share": (c_int, [c_void_p, c_int]),
    "sdrm_debug_rowchain_available": (c_int, [c_void_p]),
    "sdrm_debug_set_wgrad_strips": (c_int, [c_void_p, c_int]),
    "sdrm_debug_set_dgrad_rows": (c_int, [c_void_p, c_int]),
    "sdrm_debug_comm_handle": (c_void_p, [c_void_p]),
    "sdrm_debug_plan_wgrad": (c_int, [c_int, c_int, c_int, C.POINTER(c_int), C.POINTER(c_int)]),
    "sdrm_debug_gemm_time": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_float), c_void_p]),
    "sdrm_debug_gemm": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
}

RNG_EXPLICIT, RNG_PHILOX = 0, 1
STATUS = {0: "SDRM_OK", -1: "SDRM_ERR_ARG", -2: "SDRM_ERR_SHAPE", -3: "SDRM_ERR_HIP", -4: "SDRM_ERR_STATE",
          -5: "SDRM_ERR_NOMEM", -6: "SDRM_ERR_RCCL", -7: "SDRM_ERR_DEVICE"}


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Loads the HIP library, rebuilding it first when it is missing or was compiled from other sources than the ones
    next to it (`_build.is_stale()`: the SHA-256 of csrc/ + include/ compiled into the binary against the files as they
    are now).  Raises RuntimeError if that is impossible - there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm ships its own libamdhip64: it must be the HIP runtime this process loads FIRST.  Loaded after a system copy that the
    # library's DT_NEEDED pulled in, it finds no device ("no ROCm-capable device is detected" from sdrm_create on a box with a GPU).
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - a C-ABI-only consumer without PyTorch: the system runtime is then the only one
        pass
    path = lib_path()
    override = os.environ.get("SDRM_LIB")   # diagnostics only (tools/ab_bench.sh): another build of the same ABI, loaded as it is
    if override:
        lib = C.CDLL(os.path.abspath(override))
        missing = [name for name in SIGNATURES if not hasattr(lib, name)]
        if missing:   # another ABI than include/sdrm_hip.h: refuse rather than call through mismatched signatures
            raise RuntimeError(f"SDRM_LIB={override}: not this ABI, missing {', '.join(missing[:6])}{' ...' if len(missing) > 6 else ''}")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        import sys
        print(f"sdrm_amd: SDRM_LIB override {override} (sources {lib.sdrm_source_hash().decode()[:16]}; in-tree sources "
              f"{_build.source_hash()[:16]})", file=sys.stderr)
        _LIB = lib
        return lib
    if _build.is_stale():
        if not build_if_missing:
            what = "is missing" if not os.path.exists(path) else "was built from different sources than csrc/ + include/"
            raise RuntimeError(f"{path} {what}: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _build.build_library()
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise RuntimeError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    got, want = lib.sdrm_source_hash().decode(), _build.source_hash()
    if got != want:  # pragma: no cover - is_stale() above reads the same bytes from the file
        raise RuntimeError(f"{path} carries source hash {got[:16]}, the sources hash to {want[:16]}")
    _LIB = lib
    return lib
