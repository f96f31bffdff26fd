"""DDPGLearner (serl/agents/ddpg/ddpg_learner.py:18-225) on the SAC learner's kernels.

The reference's "DDPG" is the SAC update with a single critic (`Ensemble(num=1)`, :66), no temperature and no
entropy term: the critic step is `SACLearner.update_critic` with `backup_entropy=False` and a 1-of-1 "subsample"
(:146-200, same four rng splits), the actor loss is `-mean(q)` (:127-139; the policy is still the stochastic
TanhNormal and is sampled), there is no temperature step, and `info` holds `actor_loss, critic_loss, q` (:225).
Here that is SACLearner with a temperature buffer pinned at log(0) = -inf, i.e. alpha = 0 in the shared actor-loss
and tanh-Gaussian backward kernels, and the temperature step skipped.

Exploration (`compute_actions`, :91-104) adds `0.05 * normal(key, shape)` to the sampled action; the reference
derives that key from `self.rng` a second time (:102), so it is the SAME key and shape the policy sample was drawn
with: the noise equals 0.05 x the standard-normal draw behind the sample.  Reproduced as is.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from ... import prng
from ...networks.spec import NetConfig
from ..sac.sac_learner import SACLearner
from ..train_state import TempState


class DDPGLearner(SACLearner):
    _HAS_TEMP = False
    _INFO_KEYS = ("actor_loss", "critic_loss", "q")

    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4, critic_lr: float = 3e-4,
               hidden_dims: Sequence[int] = (256, 256), discount: float = 0.99, tau: float = 0.005,
               critic_dropout_rate: Optional[float] = None, critic_layer_norm: bool = False, device=None, dist=None):
        action_dim = int(action_space.shape[-1])
        cfg = NetConfig(action_dim=action_dim, hidden_dims=tuple(int(h) for h in hidden_dims), num_qs=1,
                        critic_layer_norm=bool(critic_layer_norm), obs_dim=int(observation_space.shape[-1]))
        self = cls._build(cfg, seed, actor_lr, critic_lr, 0.0, discount, tau, None, critic_dropout_rate, 0.0, 1.0,
                          False, device, dist, nkeys=3)                       # rng, actor_key, critic_key (:47)
        dev = self.actor.store.flat.device
        # alpha = exp(-inf) = 0: the entropy terms of the shared kernels vanish exactly (0 * finite log-prob)
        self.temp = TempState(torch.tensor([float("-inf"), 0.0, 0.0], dtype=torch.float32, device=dev), 0, 0.0)
        return self

    def sample_actions(self, observations):
        """ddpg_learner.py:91-115: a ~ pi(.|obs) plus 0.05 x the same standard-normal draw (see module doc)."""
        key, rng = prng.split(self.rng)
        a, eps = self._act(observations, key, False, want_eps=True)
        return a + 0.05 * eps, self.replace(rng=rng)
