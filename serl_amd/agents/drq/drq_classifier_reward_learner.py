"""DrQClassifierRewardLearner (serl/agents/drq/drq_classifier_reward_learner.py:27-219): the DrQ learner plus a frozen
binary reward classifier `BinaryClassifier(TwoMobileNetEncoder(dropout 0.5) -> Dense(128) -> Dropout -> LayerNorm -> ReLU
-> Dense(1))` that is only inferred (`classify_reward`, :191-195; its weights come from a checkpoint:
cable_route_franka_rlpd_classifier_reward_multithread.py:141-146 -- `agent.replace(classifier=restore_checkpoint_(...))`).

Differences from DrQLearner, reproduced: target_entropy defaults to -action_dim (:61-62); `create` splits 5 keys and then
1 + len(pixel_keys) more (:66-68); `update` takes the augmentation keys as `key, rng = split(rng)` (key = child 0) and never
writes the split rng back, so the SAC update runs on the agent's original rng (:207-218; driver aug_mode 2).
The reference class passes `vice_label_smoothing=` to a dataclass without that field (:179) and cannot be constructed as
written; here the keyword is accepted and stored.  The classifier always sits on the frozen MobileNetV3
backbone (:157), whatever `encoder` the RL networks use."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ... import ops, prng
from ...networks import heads
from ...networks.params import ParamStore
from ...networks.spec import init_tree
from ..train_state import TrainState
from .drq_learner import DrQLearner


class DrQClassifierRewardLearner(DrQLearner):
    _AUG_MODE = 2

    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4, critic_lr: float = 3e-4,
               vice_lr: float = 3e-4, temp_lr: float = 3e-4, cnn_features: Sequence[int] = (32, 32, 32, 32),
               cnn_filters: Sequence[int] = (3, 3, 3, 3), cnn_strides: Sequence[int] = (2, 1, 1, 1),
               cnn_padding: str = "VALID", latent_dim: int = 50, encoder: str = "d4pg",
               hidden_dims: Sequence[int] = (256, 256), discount: float = 0.99, tau: float = 0.005, num_qs: int = 2,
               num_min_qs: Optional[int] = None, critic_dropout_rate: Optional[float] = None,
               vice_dropout_rate: Optional[float] = None, vice_label_smoothing: float = 0.1,
               critic_layer_norm: bool = False, target_entropy: Optional[float] = None, init_temperature: float = 1.0,
               backup_entropy: bool = True, pixel_keys: Tuple[str, ...] = ("pixels",), depth_keys: Tuple[str, ...] = (),
               device=None, dist=None, backbone=None):
        cfg = cls._pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                             latent_dim, encoder, hidden_dims, num_qs, critic_layer_norm, pixel_keys, depth_keys)
        if cfg.stack != 1:
            raise ValueError("the reward classifier's MobileNet backbone needs pixel observations of shape (H, W, 3, 1)")
        if target_entropy is None:
            target_entropy = -cfg.action_dim                                   # :61-62
        self = cls._build(cfg, seed, actor_lr, critic_lr, temp_lr, discount, tau, num_min_qs, critic_dropout_rate,
                          target_entropy, init_temperature, backup_entropy, device, dist, nkeys=5)   # :66
        self.rng = prng.split(self.rng, 1 + len(cfg.pixel_keys))[0]             # :67-68
        self.vice_label_smoothing = float(vice_label_smoothing)
        self._attach_reward_backbone(backbone)
        fh, fw, fc = self._shared["feat_shape"]
        lay = heads.binary_classifier_layout(len(cfg.cams), fh, fw, fc)
        dev = self.actor.store.flat.device
        self.classifier = TrainState.create(ParamStore.from_tree(lay, init_tree(lay, [int(seed), 0xC1A5]), dev), vice_lr)
        return self

    def _attach_reward_backbone(self, backbone):
        """The frozen MobileNetV3 features feed the reward network in every configuration, and the RL encoders too when
        encoder='mobilenet' (one set of pretrained variables in the reference, :97-98,157)."""
        from ...networks.mobilenet import Backbone
        dev = self.actor.store.flat.device
        if backbone is None:
            backbone = Backbone.synthetic(0, device=dev)
        self._shared["backbone"] = backbone
        self._shared["feat_shape"] = Backbone.feature_shape(self._cfg.H, self._cfg.W)
        self._shared["reward_ws"] = heads.Scratch(dev)
        if self._cfg.encoder == "mobilenet":
            self._native.set_backbone(backbone)
        return self

    def _encode_images(self, x):
        """(n,H,W,3) or (H,W,3) [optionally with a trailing stack axis of 1]: uint8 0..255 or float already in [0,1]
        -> (device features (n, fh, fw, fc), n)."""
        cfg = self._cfg
        dev = self.actor.store.flat.device
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        if t.shape[-1] == 1 and t.ndim >= 4 and t.shape[-2] == 3:
            t = t.reshape(t.shape[:-1])
        if t.ndim == 3:
            t = t[None]
        if t.dtype != torch.uint8:
            t = t.to(torch.float32)
        t = t.to(dev).contiguous()
        assert tuple(t.shape[1:]) == (cfg.H, cfg.W, 3), f"image shape {tuple(t.shape)}"
        return self._shared["backbone"].forward(t), t.shape[0]

    def classify_reward(self, obs):
        """sigmoid(classifier(obs, training=False)) (:191-195).  obs: mapping camera -> image(s) scaled to [0, 1] as the
        scripts pass them (cable_route_...py:228-231), or uint8.  Returns a numpy array (1,) / (n, 1)."""
        cfg = self._cfg
        with torch.cuda.device(self.actor.store.flat.device):
            feats, n, single = [], None, False
            keep = []
            for cam in cfg.cams:
                single = single or np.asarray(obs[cam]).ndim == 3
                f, n = self._encode_images(obs[cam])
                keep.append(f)
                feats.append(f.data_ptr())
            fh, fw, fc = self._shared["feat_shape"]
            out = heads.binary_classifier_fwd(self.classifier.store, feats, n, fh, fw, fc, self._shared["reward_ws"],
                                              ops.current_stream())
            host = out.cpu().numpy()
        return host.reshape(-1) if single else host
