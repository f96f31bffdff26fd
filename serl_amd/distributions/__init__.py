"""The tanh-Gaussian policy head (serl/distributions/tanh_normal.py, tanh_transformed.py) is evaluated by the HIP
kernels serl_tanh_gaussian_fwd / _bwd, sequenced by the native driver (driver.hip policy_fwd / policy_bwd)."""
__all__ = []
