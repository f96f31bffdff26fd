"""serl/distributions: the tanh-Gaussian policy DISTRIBUTION as a device-side value object (`TanhNormal`: sample / log_prob /
mode with the reference's semantics and jax random stream).  The flax modules that BUILD it (`Normal`, `TanhNormal(base_cls,
action_dim)`, `TanhDeterministic`, `Sigmoid`) have no counterpart: the policy network runs inside the native driver
(driver.hip policy_fwd / policy_bwd: serl_policy_fwd_fused, serl_tanh_gaussian_fwd / _bwd)."""
from .tanh_normal import TanhNormal, TanhTransformedDistribution

__all__ = ["TanhNormal", "TanhTransformedDistribution"]
