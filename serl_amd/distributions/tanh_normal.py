"""The tanh-Gaussian policy distribution as a device-side value object (mirrors what the reference's actor `apply_fn`
returns: serl/distributions/tanh_normal.py:17-53 `Normal(..., squash_tanh=True)` -> `TanhTransformedDistribution` over
`MultivariateNormalDiag`, tanh_transformed.py:17-30).

The flax MODULE `TanhNormal(base_cls, action_dim)` has no counterpart here (nothing is traced: the policy network runs
inside the native driver, csrc/driver.hip policy_fwd).  This class is the DISTRIBUTION that module returns, for callers
that hold a batch of head outputs `(mean, log_std)` on the GPU: `sample(seed=key)`, `log_prob(actions)`, `mode()` with the
reference's semantics and random stream -- `sample` draws `jax.random.normal(key, (B, A))` (threefry -> uniform -> XLA
erf_inv, bit-compatible with the oracle), `log_prob` of a sample uses the cached pre-tanh value as TFP does, `log_prob`
of arbitrary actions the Tanh bijector's float32 inverse.  All arithmetic in HIP kernels (serl_tanh_gaussian_fwd,
serl_tanh_gaussian_log_prob)."""
from __future__ import annotations

import torch

from .. import _lib, ops


class TanhNormal:
    """mean, log_std: (B, A) float32 CUDA tensors (log_std raw: clipped to [log_std_min, log_std_max] = [-20, 2] like the
    reference's Normal module -- other bounds are not supported by the kernels)."""

    def __init__(self, mean: torch.Tensor, log_std: torch.Tensor, log_std_min: float = -20.0, log_std_max: float = 2.0):
        if (log_std_min, log_std_max) != (-20.0, 2.0):
            raise NotImplementedError("the kernels clip log_std to [-20, 2] (the reference's defaults, tanh_normal.py:20-21)")
        if not (mean.is_cuda and log_std.is_cuda and mean.dtype == log_std.dtype == torch.float32 and mean.shape == log_std.shape
                and mean.ndim == 2):
            raise TypeError("TanhNormal needs two (B, A) float32 CUDA tensors")
        self.mean, self.log_std = mean.contiguous(), log_std.contiguous()
        self._last = None            # (actions, log_prob) of the latest sample: TFP caches the pre-tanh value the same way

    @property
    def batch_shape(self):
        return tuple(self.mean.shape[:1])

    @property
    def event_shape(self):
        return tuple(self.mean.shape[1:])

    def _fwd(self, key, deterministic, want_logp):
        B, A = self.mean.shape
        act = torch.empty_like(self.mean)
        logp = torch.empty(B, dtype=torch.float32, device=self.mean.device) if want_logp else None
        with torch.cuda.device(self.mean.device):
            ops.tanh_gaussian_fwd(self.mean.data_ptr(), self.log_std.data_ptr(), A, act.data_ptr(), A, None, None,
                                  None if logp is None else logp.data_ptr(), B, A, (0, 0) if key is None else key, 0, B,
                                  deterministic, ops.current_stream())
        return act, logp

    def sample(self, seed) -> torch.Tensor:
        """a = tanh(mean + exp(clip(log_std)) * eps), eps = jax.random.normal(seed, (B, A)); `seed`: a jax key (uint32[2])."""
        act, logp = self._fwd(seed, False, True)
        self._last = (act, logp)
        return act

    def sample_and_log_prob(self, seed):
        act, logp = self._fwd(seed, False, True)
        self._last = (act, logp)
        return act, logp

    def mode(self) -> torch.Tensor:
        """tanh(mean) (tanh_transformed.py:23-24)."""
        return self._fwd(None, True, False)[0]

    def log_prob(self, actions: torch.Tensor) -> torch.Tensor:
        """(B,) log pi(actions).  For the tensor the latest sample() returned: the value computed from its cached pre-tanh
        sample; else atanh(clip(actions)) as the Tanh bijector inverts in float32."""
        if self._last is not None and actions is self._last[0]:
            return self._last[1]
        B, A = self.mean.shape
        if not (actions.is_cuda and actions.dtype == torch.float32 and tuple(actions.shape) == (B, A)):
            raise TypeError(f"log_prob needs a ({B}, {A}) float32 CUDA tensor")
        actions = actions.contiguous()
        logp = torch.empty(B, dtype=torch.float32, device=self.mean.device)
        with torch.cuda.device(self.mean.device):
            _lib.call("serl_tanh_gaussian_log_prob", self.mean.data_ptr(), self.log_std.data_ptr(), A, actions.data_ptr(), A,
                      logp.data_ptr(), B, A, ops.current_stream())
        return logp


TanhTransformedDistribution = TanhNormal      # the reference's class of the returned object (tanh_transformed.py:17)
