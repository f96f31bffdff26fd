"""serl_amd: MI355X-native learner-update hot path of SERL (drop-in for serl.agents / serl.data /
serl.networks / serl.distributions).  Hand-written gfx950 HIP kernels behind a C ABI."""
__version__ = "0.1.0"
