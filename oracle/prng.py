"""ORACLE (test infrastructure only) -- CPU restatement of the JAX / TFP random streams
that the SERL learner update consumes.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
package.  The product path (`serl_amd/`) never does.

What is restated (JAX 0.4.x, `jax_threefry_partitionable=False`, x64 off; JAX itself is
absent from /root/reference -- it is an unpinned third-party dependency):

* threefry2x32 / array form / split / random_bits  -- call sites
  serl/agents/sac/sac_learner.py:121-122,168,172,177,198, serl/agents/drq/drq_learner.py:174-176,
  serl/utils/augmentations.py:46, serl/agents/agent.py:14
* randint                                          -- serl/utils/augmentations.py:6
* uniform / normal (erf_inv, XLA f32 polynomial)   -- TFP MultivariateNormalDiag.sample,
  serl/distributions/tanh_normal.py:43-45
* permutation / choice(replace=False)              -- serl/networks/ensemble.py:28

Pinning status: threefry2x32 is pinned by the three Random123 known-answer vectors and
`split` by the JAX-documented value of split(PRNGKey(0)) (tests/test_oracle_prng.py).
randint / uniform / normal / permutation follow the published JAX algorithm but no public
vector for them is available offline: PARITY UNPINNED for those streams (they only need to
agree between this oracle and the HIP path, which tests check bit-exactly for the integer
ones and to 1e-6 for normal()).
"""
from __future__ import annotations

import math

import numpy as np

U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def prng_key(seed: int) -> np.ndarray:
    """jax.random.PRNGKey(seed) -> uint32[2] = [hi, lo].  With jax_enable_x64 off (the reference never turns it on) the
    seed is an int32: the logical right shift by 32 gives hi = 0 also for negative seeds (recalled from jax/_src/prng.py
    threefry_seed; parity unpinned), lo = the two's-complement low word."""
    seed = int(seed)
    hi = (seed >> 32) & 0xFFFFFFFF if seed >= 0 else 0
    return np.array([hi, seed & 0xFFFFFFFF], dtype=U32)


def _rotl(x, r):
    return ((x << U32(r)) | (x >> U32(32 - r))).astype(U32)


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds (Random123). All arguments broadcastable uint32."""
    with np.errstate(over="ignore"):
        k0 = np.asarray(k0, dtype=U32)
        k1 = np.asarray(k1, dtype=U32)
        x0 = np.asarray(x0, dtype=U32).copy()
        x1 = np.asarray(x1, dtype=U32).copy()
        ks = (k0, k1, (k0 ^ k1 ^ U32(0x1BD11BDA)).astype(U32))
        x0 = (x0 + ks[0]).astype(U32)
        x1 = (x1 + ks[1]).astype(U32)
        for grp in range(5):
            for r in _ROT[grp % 2]:
                x0 = (x0 + x1).astype(U32)
                x1 = _rotl(x1, r)
                x1 = (x1 ^ x0).astype(U32)
            i = grp + 1
            x0 = (x0 + ks[i % 3]).astype(U32)
            x1 = (x1 + ks[(i + 1) % 3] + U32(i)).astype(U32)
    return x0, x1


def threefry_2x32(key, counts):
    """Array form used by jax._src.prng: odd lengths padded with one 0, the count vector is
    cut in halves which become the two counter words."""
    counts = np.asarray(counts, dtype=U32).ravel()
    n = counts.size
    if n % 2:
        counts = np.concatenate([counts, np.zeros(1, dtype=U32)])
    half = counts.size // 2
    y0, y1 = threefry2x32(key[0], key[1], counts[:half], counts[half:])
    return np.concatenate([y0, y1])[:n]


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split -> (num, 2) uint32; child i = out[2i:2i+2]."""
    return threefry_2x32(key, np.arange(2 * num, dtype=U32)).reshape(num, 2)


def random_bits(key, shape) -> np.ndarray:
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    return threefry_2x32(key, np.arange(n, dtype=U32)).reshape(shape)


def randint(key, shape, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint(key, shape, minval, maxval) for int32."""
    k1, k2 = split(key)
    hi = random_bits(k1, shape)
    lo = random_bits(k2, shape)
    span = U32(maxval - minval) if maxval > minval else U32(1)
    # jax/_src/random.py:_randint: multiplier = 2^16 % span, then (multiplier * multiplier) % span -- all in
    # uint32, so for span > 2^16 the product 2^32 wraps to 0 (the high word then drops out); restated as is.
    mult = (1 << 16) % int(span)
    mult = U32(((mult * mult) & 0xFFFFFFFF) % int(span))
    with np.errstate(over="ignore"):
        off = ((hi % span) * mult + (lo % span)).astype(U32) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


def uniform(key, shape, minval=0.0, maxval=1.0) -> np.ndarray:
    bits = random_bits(key, shape)
    fb = ((bits >> U32(9)) | U32(0x3F800000)).astype(U32)
    f = fb.view(np.float32) - np.float32(1.0)
    minval = np.float32(minval)
    maxval = np.float32(maxval)
    return np.maximum(minval, (f * (maxval - minval) + minval).astype(np.float32))


_ERFINV_LT = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06,
              0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941)
_ERFINV_GT = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844,
              0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682)


def erfinv_f32(x: np.ndarray) -> np.ndarray:
    """XLA's float32 erf_inv (Giles' polynomial), evaluated in float32."""
    x = np.asarray(x, dtype=np.float32)
    w = (-np.log1p((-x * x).astype(np.float32))).astype(np.float32)
    lt = w < np.float32(5.0)
    wl = (w - np.float32(2.5)).astype(np.float32)
    wg = (np.sqrt(np.maximum(w, np.float32(0))) - np.float32(3.0)).astype(np.float32)
    ww = np.where(lt, wl, wg).astype(np.float32)
    p = np.where(lt, np.float32(_ERFINV_LT[0]), np.float32(_ERFINV_GT[0])).astype(np.float32)
    for cl, cg in zip(_ERFINV_LT[1:], _ERFINV_GT[1:]):
        c = np.where(lt, np.float32(cl), np.float32(cg)).astype(np.float32)
        p = (c + p * ww).astype(np.float32)
    out = (p * x).astype(np.float32)
    return np.where(np.abs(x) == 1, np.float32(np.inf) * x, out).astype(np.float32)


def normal(key, shape) -> np.ndarray:
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0), dtype=np.float32)
    u = uniform(key, shape, lo, 1.0)
    return (np.float32(math.sqrt(2.0)) * erfinv_f32(u)).astype(np.float32)


def fold_in(key, data: int) -> np.ndarray:
    """jax.random.fold_in(key, data) = threefry_2x32(key, threefry_seed(data)) with the 32-bit seed [0, data]."""
    return threefry_2x32(key, np.array([0, int(data) & 0xFFFFFFFF], dtype=U32))


def bernoulli(key, p, shape) -> np.ndarray:
    """jax.random.bernoulli(key, p, shape) = uniform(key, shape) < p."""
    return uniform(key, shape) < np.float32(p)


def permutation(key, n: int) -> np.ndarray:
    """jax.random.permutation(key, arange(n)): rounds of stable sort by fresh random bits."""
    x = np.arange(n, dtype=np.int32)
    rounds = int(np.ceil(3 * np.log(max(1, n)) / np.log(np.iinfo(np.uint32).max)))
    for _ in range(rounds):
        key, sub = split(key)
        sort_keys = random_bits(sub, (n,))
        x = x[np.argsort(sort_keys, kind="stable")]
    return x


def choice_no_replace(key, n: int, m: int) -> np.ndarray:
    """jax.random.choice(key, arange(n), (m,), replace=False)."""
    return permutation(key, n)[:m]
