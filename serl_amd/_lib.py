"""ctypes loader for libserl_hip.so (the C ABI declared in include/serl_hip.h).

The product path has no CPU fallback: if the HIP library is missing or an entry point
fails, this module raises.  PyTorch is used only for device memory / streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SERL_AMD_LIB") or os.path.join(_HERE, "csrc", "libserl_hip.so")   # override: kernel experiments only

c_p = C.c_void_p
c_i = C.c_int
c_u = C.c_uint
c_ll = C.c_longlong
c_f = C.c_float
PP = C.POINTER(C.c_void_p)
PI = C.POINTER(C.c_int)

# name -> argtypes (restype is int status unless listed in _RESTYPE)
_SIGS = {
    "serl_version": [],
    "serl_device_arch": [c_i, C.c_char_p, c_i],
    "serl_set_matmul_precision": [c_i],
    "serl_get_matmul_precision": [],
    "serl_conv2d_fwd": [c_i, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_conv2d_dgrad": [c_i, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_conv2d_wgrad_workspace_bytes": [c_i] * 9,
    "serl_conv2d_wgrad": [c_i, PP, PP, PP, PP, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_gemm": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i,
                  c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll,
                  c_i, c_ll, c_i, c_i, c_i, c_i, c_p],
    "serl_ln_act_fwd": [c_p, c_i, c_ll, c_i, c_p, c_p, c_p, c_ll, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_ln_act_head_fwd": [c_p, c_i, c_ll, c_i, c_p, c_p, c_p, c_ll, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i,
                             c_p, c_ll, c_p, c_ll, c_p, c_p],
    "serl_ln_act_bwd": [c_p, c_i, c_p, c_p, c_ll, c_p, c_i, c_p, c_p, c_p, c_ll, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_trunk_ln_fwd": [c_p, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p,
                          c_p, c_p, c_i, c_i, c_i, c_p],
    "serl_ln_act_bwd_seg": [c_i, PP, PI, c_i, c_ll, PP, PI, PP, PP, PP, PP, PP, c_i, PI, c_i, c_p],
    "serl_colsum": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_p],
    "serl_rowdot": [c_p, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_p],
    "serl_copy2d": [c_p, c_i, c_p, c_i, c_i, c_i, c_p],
    "serl_fill": [c_p, c_f, c_ll, c_p],
    "serl_tanh_gaussian_fwd": [c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_u, c_u, c_i, c_i, c_i, c_p],
    "serl_tanh_gaussian_bwd": [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_p],
    "serl_tanh_gaussian_nll": [c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_i, c_p, c_p],
    "serl_bce_logits": [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p],
    "serl_ensemble_fwd_fused": [c_p, c_i, c_i, c_i, c_i] + [c_p] * 10 + [c_ll] * 4 + [c_i] + [c_p] * 8,
    "serl_tanh_gaussian_log_prob": [c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_p],
    "serl_ensemble_fwd_fused_pair": [c_p, c_p, c_p],
    "serl_policy_fwd_fused": [c_p, c_i, c_i, c_i] + [c_p] * 8 + [c_i, c_u, c_u, c_i, c_i, c_i, c_p, c_i] + [c_p] * 6,
    "serl_ensemble_bwd_fused": [c_p, c_i, c_i] + [c_p] * 4 + [c_i] + [c_p] * 11,
    "serl_ensemble_colsum": [c_p] * 8 + [c_i] * 3 + [c_p] * 8 + [c_p, c_i, c_i, c_p, c_p],
    "serl_td_loss": [c_p, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p],
    "serl_actor_loss": [c_p, c_i, c_i, c_p, c_p, c_p, c_p],
    "serl_adam_polyak": [c_p, c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p],
    "serl_temperature_update": [c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p],
    "serl_sle_fwd": [c_i, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_sle_bwd": [c_i, PP, c_i, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_convs_weight_bytes": [c_i] * 5,
    "serl_split_f32": [c_p, c_p, c_ll, c_i, c_p],
    "serl_merge_split": [c_p, c_p, c_ll, c_i, c_p],
    "serl_convs_prep_weights": [c_i, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_convs_prep_weights_multi": [c_i, PP, PP, PP, PI, PI, PI, PI, PI, c_p],
    "serl_convs_fwd": [c_i, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, PP, c_p],
    "serl_convs_dgrad": [c_i, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_convs_wgrad": [c_i, PP, PP, PP, PP, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_convs_wgrad_noalloc": [c_i, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_convs_wgrad_noalloc_max_bytes": [],
    "serl_convs_set_patch_kernels": [c_i],
    "serl_convs_set_cu_limit": [c_i],
    "serl_mbv3_set_fused": [c_i],
    "serl_convs_fwd12_supported": [c_i, c_i, c_i, c_i, c_i],
    "serl_convs_fwd12": [c_i, PP, PP, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, PP, c_p],
    "serl_convs_fwd12_keep_supported": [c_i, c_i, c_i, c_i, c_i],
    "serl_convs_fwd12_keep": [c_i, PP, PP, PP, PP, PP, PP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, PP, PP, PP, c_p],
    "serl_convs_dgrad_wgrad1_supported": [c_i, c_i, c_i, c_i, c_i],
    "serl_convs_dgrad_wgrad1_workspace_bytes": [c_i],
    "serl_convs_dgrad_wgrad1": [c_i, PP, PP, PP, PP, PP, PP, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_gather_crop_frames": [c_p, c_ll, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_gather_crop_packed": [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "serl_gather_window": [c_p, c_ll, c_p, c_p, c_i, c_i, c_i, c_p],
    "serl_crop_offsets": [c_u, c_u, c_i, c_i, c_i, c_i, c_p, c_p],
    "serl_prng_randint": [c_u, c_u, c_i, c_i, c_i, c_p, c_p, c_p],
    "serl_take_rows": [c_p, c_ll, c_ll, c_p, c_p, c_i, c_p],
    "serl_mbv3_param_floats": [],
    "serl_mbv3_feature_shape": [c_i, c_i, c_p, c_p, c_p],
    "serl_mbv3_workspace_bytes": [c_i, c_i, c_i],
    "serl_mbv3_forward": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_ll, c_p, c_p],
    "serl_learner_create": [c_p, c_p, c_i, c_p, c_i, c_p],
    "serl_learner_destroy": [c_p],
    "serl_learner_set_backbone": [c_p, c_p],
    "serl_learner_update": [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i],
    "serl_learner_act": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p],
    "serl_learner_bc_update": [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "serl_learner_bc_logit": [c_p, c_p, c_p, c_p, c_p, c_p],
    "serl_learner_read_buffer": [c_p, c_i, C.c_char_p, c_p, c_ll, c_p],
    "serl_learner_buffer_ptr": [c_p, c_i, C.c_char_p, c_p, c_p],
    "serl_learner_comm_stream": [c_p, c_p],
    "serl_learner_workspace_bytes": [c_p],
    "serl_learner_set_timing": [c_p, c_i, c_i],
    "serl_learner_timing_count": [c_p],
    "serl_learner_timing_get": [c_p, c_i, c_p, c_p],
    "serl_learner_phase_count": [c_p],
    "serl_learner_phase_get": [c_p, c_i, c_p, c_p],
    "serl_learner_timing_reset": [c_p],
    "serl_prng_split": [c_u, c_u, c_i, c_p],
    "serl_prng_bits": [c_u, c_u, c_i, c_p],
    "serl_prng_choice": [c_u, c_u, c_i, c_i, c_p],
    "serl_prng_fold_in": [c_u, c_u, c_u, c_p],
    "serl_row_combine": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_ll, c_p],
    "serl_bce_mixup": [c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p],
    "serl_gp_norm": [c_p, c_i, c_ll, c_f, c_p, c_p, c_p],
    "serl_ln_tangent_fwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "serl_ln_tangent_bwd": [c_p] * 12 + [c_i, c_i, c_i, c_i, c_p],
    "serl_axpy": [c_p, c_p, c_f, c_ll, c_p],
    "serl_vice_reward": [c_i, PP, PP, c_p, c_p, c_i, c_p],
    "serl_dropout": [c_p, c_i, c_i, c_i, c_u, c_u, c_f, c_i, c_i, c_p],
}
_RESTYPE = {"serl_conv2d_wgrad_workspace_bytes": c_ll, "serl_convs_weight_bytes": c_ll,
            "serl_convs_wgrad_noalloc_max_bytes": c_ll, "serl_learner_workspace_bytes": c_ll, "serl_mbv3_param_floats": c_ll,
            "serl_mbv3_workspace_bytes": c_ll, "serl_convs_dgrad_wgrad1_workspace_bytes": c_ll}

EXPORTS = tuple(_SIGS) + ("serl_last_error",)

_lib = None


class SerlHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SerlHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C serl_amd/csrc`). There is no CPU fallback.")
    # torch first: it ships its own libamdhip64; loading ours first would pull a second HIP runtime (/opt/rocm)
    # into the process, and kernels registered there see "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, c_i)
    lib.serl_last_error.argtypes = []
    lib.serl_last_error.restype = C.c_char_p
    _lib = lib
    if os.environ.get("SERL_AMD_CONV_PATCH") in ("0", "2", "3", "4", "5", "6", "7", "8"):   # A/B measurements (serl_convs_set_patch_kernels)
        lib.serl_convs_set_patch_kernels(int(os.environ["SERL_AMD_CONV_PATCH"]))
    env = os.environ.get("SERL_AMD_MATMUL_PRECISION")
    if env is not None:
        from . import set_matmul_precision
        set_matmul_precision(env)
    return lib


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().serl_last_error().decode()
        if status == 1:
            raise ValueError(f"{what}: {msg}")
        raise SerlHipError(f"{what}: {msg}")


def call(name: str, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def ptr_table(ptrs):
    """Host array of device pointers for the grouped conv entry points."""
    arr = (C.c_void_p * len(ptrs))(*[int(p) if p else None for p in ptrs])
    return C.cast(arr, PP), arr
