"""VICELearner (serl/agents/vice/vice_learner.py:29-474): DrQ-RLPD whose reward is a learned VICE classifier per camera
on frozen MobileNetV3 features.

  encode(x)                      :231-244  frozen MobileNetEncoder(reshape, /255, ImageNet normalise, backbone)
  vice_reward(obs, pixel_keys)   :246-275  per camera `key, rng = split(rng)`; sigmoid(classifier(encode(obs[cam])),
                                           training=False); 'mean' over the cameras; returns (agent with the advanced rng, dict)
  update(batch, utd, pixel_keys) :277-314  _unpack, share encoder, augmentation keys `key, rng = split(rng)` (not written
                                           back), rewards := log(mean / (1 - mean)) of the AUGMENTED observations' VICE
                                           rewards, SAC update from the rng vice_reward left; info gains "vice_rews"
  update_classifier(batch, keys) :316-474  serl_amd/agents/vice/classifier_update.py
The RL networks use TwoMobileNetEncoder with the same frozen backbone (the reference's create() only works with
encoder="mobilenet": the VICE encoder refers to the MobileNet of that branch, :95-97,164).  In update() the driver computes
the backbone features of all augmented rows once; a `features ready` hook evaluates the classifiers on them and writes the
rewards before the first critic step reads them."""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ... import ops, prng
from ...native import UpdateBatch
from ...networks import heads
from ...networks.params import ParamStore
from ...networks.spec import init_tree
from ..drq.drq_classifier_reward_learner import DrQClassifierRewardLearner
from ..sac.sac_learner import _to_dev
from ..train_state import TrainState


class VICELearner(DrQClassifierRewardLearner):
    _AUG_MODE = 2
    _INFO_KEYS = ("actor_loss", "entropy", "critic_loss", "q", "temperature", "temperature_loss", "vice_rews")

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
               vice_goal_pool=None, device=None, dist=None, backbone=None):
        if encoder != "mobilenet":
            raise NotImplementedError("VICELearner needs encoder='mobilenet': its VICE encoder is the MobileNet of that "
                                      "branch (vice_learner.py:95-97,164); 'd4pg' raises NameError in the reference")
        cfg = cls._pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                             latent_dim, encoder, hidden_dims, num_qs, critic_layer_norm, pixel_keys, depth_keys)
        if target_entropy is None:
            target_entropy = -cfg.action_dim                                   # :76-77
        self = cls._build(cfg, seed, actor_lr, critic_lr, temp_lr, discount, tau, num_min_qs, critic_dropout_rate,
                          target_entropy, init_temperature, backup_entropy, device, dist, nkeys=5)   # :80
        self.rng = prng.split(self.rng, 1 + len(cfg.pixel_keys))[0]             # :81-82
        self.vice_label_smoothing = float(vice_label_smoothing)
        self.vice_dropout_rate = vice_dropout_rate
        self.vice_goal_pool = vice_goal_pool
        self._attach_reward_backbone(backbone)
        fh, fw, fc = self._shared["feat_shape"]
        dev = self.actor.store.flat.device
        lay = heads.vice_classifier_layout(fh, fw, fc, cfg.critic_layer_norm)   # use_layer_norm=critic_layer_norm (:159)
        self.vice_classifiers = OrderedDict()
        for i, p_key in enumerate(cfg.pixel_keys):                              # one classifier per camera (:181-204)
            store = ParamStore.from_tree(lay, init_tree(lay, [int(seed), 0x71CE, i]), dev)
            self.vice_classifiers[p_key] = TrainState.create(store, vice_lr)
        return self

    @property
    def vice_encoder(self):
        """The frozen feature extractor (reference: a TrainState with a no-op optimiser, :171-175)."""
        return self._shared["backbone"]

    # ------------------------------------------------------------------ encode / reward
    def encode(self, x):
        """:231-244: images (n,H,W,3,1) / (H,W,3,1) uint8 -> frozen MobileNetV3 features (n, fh, fw, 960), device tensor."""
        with torch.cuda.device(self.actor.store.flat.device):
            return self._encode_images(x)[0]

    def _vice_logits(self, feat_ptrs, n, st, tag=""):
        """feat_ptrs: {camera: device pointer of (n, fh, fw, fc) features} -> [logits tensor (n,) per pixel key]."""
        fh, fw, fc = self._shared["feat_shape"]
        ws = self._shared["reward_ws"]
        return [heads.vice_classifier_fwd(self.vice_classifiers[k].store, feat_ptrs[k], n, fh, fw, fc,
                                          self._cfg.critic_layer_norm, ws, st, tag=f"{tag}_{k}")
                for k in self._cfg.pixel_keys]

    def vice_reward(self, observations, pixel_keys=("pixels",)):
        """:246-275 -> (agent with rng advanced by one split per camera, OrderedDict camera -> (n,) rewards, 'mean')."""
        pixel_keys = self._check_keys(pixel_keys)
        rng = self.rng
        with torch.cuda.device(self.actor.store.flat.device):
            st = ops.current_stream()
            feats, keep, n = {}, [], None
            for k in pixel_keys:
                _, rng = prng.split(rng)                    # the dropout key: unused with training=False, still consumed
                f, n = self._encode_images(observations[k])
                keep.append(f)
                feats[k] = f.data_ptr()
            logits = self._vice_logits(feats, n, st, tag="r")
            dev = self.actor.store.flat.device
            sig = [torch.empty(n, dtype=torch.float32, device=dev) for _ in pixel_keys]
            mean = torch.empty(n, dtype=torch.float32, device=dev)
            ops.vice_reward([t.data_ptr() for t in logits], [t.data_ptr() for t in sig], mean.data_ptr(), None, n, st)
        out = OrderedDict((k, s) for k, s in zip(pixel_keys, sig))
        out["mean"] = mean
        return self.replace(rng=rng), out

    def _check_keys(self, pixel_keys):
        pixel_keys = tuple(pixel_keys)
        if pixel_keys != self._cfg.pixel_keys and pixel_keys != ("pixels",):
            raise ValueError(f"pixel_keys {pixel_keys} differ from the keys the agent was created with {self._cfg.pixel_keys}")
        return self._cfg.pixel_keys

    # ------------------------------------------------------------------ update
    def update(self, batch, utd_ratio: int, pixel_keys=("pixels",)):
        cfg = self._cfg
        pixel_keys = self._check_keys(pixel_keys)
        n = len(batch["rewards"])
        assert n % utd_ratio == 0
        dev = self.actor.store.flat.device
        f32 = torch.float32
        B = n // utd_ratio
        actor = self.actor.clone()                                              # share the encoder once (:290-291)
        nenc = self.critic.store.layout.prefix_size("image_encoder")
        actor.store.flat[:nenc].copy_(self.critic.store.flat[:nenc])
        ub = UpdateBatch()
        keep = [_to_dev(batch["observations"]["state"], dev, f32).reshape(n, cfg.state_dim),
                _to_dev(batch["next_observations"]["state"], dev, f32).reshape(n, cfg.state_dim)]
        ub.state, ub.next_state = keep[0].data_ptr(), keep[1].data_ptr()
        from ...data.device_batch import hold_windows
        with hold_windows(batch["observations"].values()):      # released once the gather is enqueued (drq_learner.py)
            lazy = self._pixel_sources(batch, n, ub, keep)
            # new_agent.vice_reward(observations) runs on the agent's ORIGINAL rng (the augmentation splits are not
            # written back) and advances it by one split per camera; SACLearner.update continues from there (:298-310)
            rng_sac = self.rng
            for _ in pixel_keys:
                _, rng_sac = prng.split(rng_sac)
            ub.use_sac_rng, ub.sac_rng[0], ub.sac_rng[1] = 1, int(rng_sac[0]), int(rng_sac[1])
            rewards = torch.empty(n, dtype=f32, device=dev)                    # the batch's rewards are replaced (:304-308)
            rew_sum = torch.zeros(1, dtype=f32, device=dev)
            native = self._native

            def vice_rewards_ready():
                st = ops.current_stream()
                feats = {k: native.buffer_ptr(B, f"bb_0_{cfg.cams.index(k)}")[0] for k in pixel_keys}   # augmented obs
                logits = self._vice_logits(feats, n, st, tag="u")
                ops.vice_reward([t.data_ptr() for t in logits], None, None, rewards.data_ptr(), n, st)
                ops.colsum(rewards.data_ptr(), rew_sum.data_ptr(), 1, n, 1, 1, 0, st)
            out = self._run_update(dict(batch, rewards=rewards), ub, n, utd_ratio, actor, self.rng, keep,
                                   features_hook=vice_rewards_ready, extra_info={"vice_rews": (rew_sum, 1.0 / n)})
            for fw in lazy:
                fw.consumed = True
        return out

    # ------------------------------------------------------------------ classifier training
    def update_classifier(self, batch, pixel_keys=("pixels",)):
        """:316-474 (mixup + label smoothing + gradient penalty)."""
        from .classifier_update import update_classifier
        return update_classifier(self, batch, self._check_keys(pixel_keys))
