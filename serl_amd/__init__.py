"""serl_amd: MI355X-native learner-update hot path of SERL (drop-in for serl.agents / serl.data /
serl.networks / serl.distributions).  Hand-written gfx950 HIP kernels behind a C ABI."""
__version__ = "0.1.0"


_PRECISIONS = {"highest": 0, "float32": 0, "fp32": 0, "bf16_3x": 1, "bfloat16_3x": 1}


def set_matmul_precision(name: str) -> None:
    """Precision of all conv / dense kernels, process-wide (like jax.config "jax_default_matmul_precision"):
    "highest" = exact fp32 MFMA, "bf16_3x" (default) = bf16 hi/lo split, 3 bf16 MFMAs, fp32 accumulate.
    The environment variable SERL_AMD_MATMUL_PRECISION sets the initial value."""
    from . import _lib
    if name not in _PRECISIONS:
        raise ValueError(f"unknown precision {name!r}; one of {sorted(_PRECISIONS)}")
    _lib.call("serl_set_matmul_precision", _PRECISIONS[name])


def get_matmul_precision() -> str:
    from . import _lib
    return "highest" if _lib.load().serl_get_matmul_precision() == 0 else "bf16_3x"
