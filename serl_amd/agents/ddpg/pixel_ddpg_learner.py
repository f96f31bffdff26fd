"""PixelDDPGLearner (serl/agents/ddpg/pixel_ddpg_learner.py:25-181) on the DrQ learner's kernels.

`update` = `_unpack`, critic encoder shared into the actor once, random-shift augmentation of obs and next_obs,
then `DDPGLearner.update` (one critic inside the PixelMultiplexer, -mean(q) actor loss, no temperature).
Two differences from DrQLearner.update, both reproduced: the augmentation keys are `key, rng = split(rng)`
(:169,171 -- key is child 0; DrQ takes child 1), and the split rng is dropped -- `DDPGLearner.update` runs on
`new_agent`, whose rng was never replaced (:167,180), so the critic step's first key equals the obs
augmentation key and the agent's rng advances only by the DDPG update's own splits.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from ..drq.drq_learner import DrQLearner
from ..train_state import TempState
from .ddpg_learner import DDPGLearner


class PixelDDPGLearner(DrQLearner):
    _HAS_TEMP = False
    _INFO_KEYS = ("actor_loss", "critic_loss", "q")

    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4, critic_lr: float = 3e-4,
               cnn_features: Sequence[int] = (32, 32, 32, 32), cnn_filters: Sequence[int] = (3, 3, 3, 3),
               cnn_strides: Sequence[int] = (2, 1, 1, 1), cnn_padding: str = "VALID", latent_dim: int = 50,
               encoder: str = "d4pg", hidden_dims: Sequence[int] = (256, 256), discount: float = 0.99,
               tau: float = 0.005, critic_dropout_rate: Optional[float] = None, critic_layer_norm: bool = False,
               pixel_keys: Tuple[str, ...] = ("pixels",), depth_keys: Tuple[str, ...] = (), device=None, dist=None,
               backbone=None):
        cfg = cls._pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                             latent_dim, encoder, hidden_dims, 1, critic_layer_norm, pixel_keys, depth_keys)
        self = cls._build(cfg, seed, actor_lr, critic_lr, 0.0, discount, tau, None, critic_dropout_rate, 0.0, 1.0,
                          False, device, dist, nkeys=3)                       # rng, actor_key, critic_key (:59)
        dev = self.actor.store.flat.device
        self.temp = TempState(torch.tensor([float("-inf"), 0.0, 0.0], dtype=torch.float32, device=dev), 0, 0.0)
        return self._attach_backbone(backbone)      # encoder="mobilenet" (pixel_ddpg_learner.py:78-83)

    # key, rng = split(rng) for the two augmentation keys (:169,171), and the DDPG update runs on new_agent, whose rng
    # was never replaced (:180): evaluated by the native driver (aug_mode 2, driver.hip serl_learner_update)
    _AUG_MODE = 2

    sample_actions = DDPGLearner.sample_actions
