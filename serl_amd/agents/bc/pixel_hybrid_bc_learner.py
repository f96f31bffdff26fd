"""PixelHybridBCLearner on MI355X (mirrors serl/agents/bc/pixel_hybrid_bc_learner.py:27-185).

Two networks on the same augmented observations: the continuous TanhNormal actor over the first action_dim - 1 action
dimensions (its encoder is trained, stop_gradient=False) and a discrete "gripper" actor PixelMultiplexer(stop_gradient=True)
-> MLP -> Dense(1) trained with a sigmoid BCE against (last action dimension > 0).  Its encoder leaves get a zero
gradient, so Adam leaves them at their initial values (:93-106) -- reproduced: the gradient buffer's encoder prefix is
never written.

update(batch, utd_ratio, pixel_keys): `utd_ratio` is accepted and unused, as in the reference (:136).  rng, key1, key2 =
split(self.rng, 3) (:154); the augmentation key is child 0 of split(self.rng) (:142) and that split's rng is dropped.

compute_actions / eval_actions (:123-133): the reference reads `self.continuous_actor`, a field the class never defines
(AttributeError in the snapshot).  BUILD-DEFINED as the evident intent: the mode of `self.actor`, concatenated with the
discrete actor's logit; eval_actions then sets the last entry to 0.5 if it exceeds 0.5, else -0.5 (:132).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch

from ... import ops
from .pixel_bc_learner import PixelBCLearner


class PixelHybridBCLearner(PixelBCLearner):
    _HYBRID = True
    _INFO_KEYS = ("nll", "bce")

    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4,
               cnn_features: Sequence[int] = (32, 32, 32, 32), cnn_filters: Sequence[int] = (3, 3, 3, 3),
               cnn_strides: Sequence[int] = (2, 1, 1, 1), cnn_padding: str = "VALID", latent_dim: int = 50,
               encoder: str = "d4pg", hidden_dims: Sequence[int] = (256, 256), pixel_keys: Tuple[str, ...] = ("pixels",),
               depth_keys: Tuple[str, ...] = (), device=None, backbone=None):
        return cls._create(seed, observation_space, action_space, actor_lr, cnn_features, cnn_filters, cnn_strides,
                           cnn_padding, latent_dim, encoder, hidden_dims, pixel_keys, depth_keys, device, backbone)

    def update(self, batch, utd_ratio: int, pixel_keys=("pixels",)):
        return self._bc_update(batch, pixel_keys)

    def compute_actions(self, observations) -> np.ndarray:
        cfg = self._cfg
        dev = self.actor.store.flat.device
        mode = self._act(observations, None, True)                       # tanh(mean) of the continuous actor
        with torch.cuda.device(dev):
            keep, imgs = [], []
            for cam in cfg.cams:
                x = np.ascontiguousarray(np.asarray(observations[cam], dtype=np.uint8)).reshape(cfg.H, cfg.W, 3 * cfg.stack)
                t = torch.from_numpy(x).to(dev)
                keep.append(t)
                imgs.append(t.data_ptr())
            s = torch.from_numpy(np.asarray(observations["state"], dtype=np.float32).reshape(-1)).to(dev)
            out = torch.empty(1, dtype=torch.float32, device=dev)
            self._shared["native"].bc_logit(self.discrete_actor.store.base, imgs, s.data_ptr(), out.data_ptr(),
                                            ops.current_stream())
            logit = out.cpu().numpy()
        return np.concatenate([mode, logit]).astype(np.float32)

    def eval_actions(self, observations) -> np.ndarray:
        action = np.array(self.compute_actions(observations), copy=True)
        action[-1] = 0.5 if action[-1] > 0.5 else -0.5
        return action
