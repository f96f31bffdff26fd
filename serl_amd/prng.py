"""Host-side JAX-compatible PRNG key plumbing for the product path (native: the threefry code in
libserl_hip shared with the device kernels).  Mirrors jax.random.{PRNGKey, split, choice} as the
reference calls them (serl/agents/sac/sac_learner.py:63-64,121-122,168-198)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)                     # x64 off: a negative int32 seed has high word 0 (oracle/prng.py prng_key)
    hi = (seed >> 32) & 0xFFFFFFFF if seed >= 0 else 0
    return np.array([hi, seed & 0xFFFFFFFF], dtype=np.uint32)


def split(key, num: int = 2) -> np.ndarray:
    out = np.empty(2 * num, dtype=np.uint32)
    _lib.call("serl_prng_split", int(key[0]), int(key[1]), num, out.ctypes.data_as(C.c_void_p))
    return out.reshape(num, 2)


def bits(key, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    _lib.call("serl_prng_bits", int(key[0]), int(key[1]), n, out.ctypes.data_as(C.c_void_p))
    return out


def choice_no_replace(key, n: int, m: int) -> np.ndarray:
    out = np.empty(m, dtype=np.int32)
    _lib.call("serl_prng_choice", int(key[0]), int(key[1]), n, m, out.ctypes.data_as(C.c_void_p))
    return out


def fold_in(key, data: int) -> np.ndarray:
    out = np.empty(2, dtype=np.uint32)
    _lib.call("serl_prng_fold_in", int(key[0]), int(key[1]), int(data) & 0xFFFFFFFF, out.ctypes.data_as(C.c_void_p))
    return out
