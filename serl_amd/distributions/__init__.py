"""The tanh-Gaussian policy head (serl/distributions/tanh_normal.py, tanh_transformed.py) is evaluated
by the HIP kernels serl_tanh_gaussian_fwd / _bwd; see serl_amd.networks.modules.policy_fwd."""
from ..networks.modules import policy_bwd, policy_fwd

__all__ = ["policy_fwd", "policy_bwd"]
