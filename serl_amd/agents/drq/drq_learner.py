"""DrQ pixel learner on MI355X (mirrors serl/agents/drq/drq_learner.py).

update(batch, utd_ratio, pixel_keys):
  1. `_unpack` of packed S+1 frame stacks          (drq_learner.py:168-169, utils/commons.py:7-29)
  2. critic encoder copied into the actor, once      (:171-172, commons.py:32-47)
  3. random-shift augmentation of obs and next_obs, whole B*U batch, per-camera keys  (:174-177, :131-137)
  4. SACLearner.update on the augmented batch         (:186)
Steps 1 and 3 (and, for device-resident replay batches, the frame-stack gather of
MemoryEfficientReplayBuffer.sample) are ONE HIP kernel per camera and side (gather_crop.hip), launched -- like
everything else of the call -- by the native step driver (driver.hip); this module resolves the batch's pixel leaves
(lazy frame windows / packed / unpacked, host or device) into device pointers.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ...data.device_batch import FrameWindows, hold_windows
from ...native import KIND_FRAMES, KIND_PACKED, KIND_UNPACKED, UpdateBatch
from ...networks.spec import NetConfig
from ...spaces import is_dict
from ..sac.sac_learner import SACLearner, _to_dev


class DrQLearner(SACLearner):
    _AUG_MODE = 1      # rng, key = split(rng): the augmentation key is child 1 and the rng advances (drq_learner.py:174-177)

    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4, critic_lr: float = 3e-4,
               temp_lr: float = 3e-4, cnn_features: Sequence[int] = (32, 32, 32, 32),
               cnn_filters: Sequence[int] = (3, 3, 3, 3), cnn_strides: Sequence[int] = (2, 1, 1, 1),
               cnn_padding: str = "VALID", latent_dim: int = 50, encoder: str = "d4pg",
               hidden_dims: Sequence[int] = (256, 256), discount: float = 0.99, tau: float = 0.005, num_qs: int = 2,
               num_min_qs: Optional[int] = None, critic_dropout_rate: Optional[float] = None,
               critic_layer_norm: bool = False, target_entropy: Optional[float] = None,
               init_temperature: float = 1.0, backup_entropy: bool = True, pixel_keys: Tuple[str, ...] = ("pixels",),
               depth_keys: Tuple[str, ...] = (), device=None, dist=None, backbone=None):
        """`backbone` (extra keyword, encoder="mobilenet" only): a serl_amd.networks.mobilenet.Backbone holding the frozen
        MobileNetV3-Large weights.  The reference downloads pretrained ones through jeffnet (drq_classifier_reward_
        learner.py:94-98), which is impossible offline: without the keyword a synthetic backbone is used."""
        cfg = cls._pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                             latent_dim, encoder, hidden_dims, num_qs, critic_layer_norm, pixel_keys, depth_keys)
        self = cls._build(cfg, seed, actor_lr, critic_lr, temp_lr, discount, tau, num_min_qs, critic_dropout_rate,
                          target_entropy, init_temperature, backup_entropy, device, dist)
        return self._attach_backbone(backbone)

    def _attach_backbone(self, backbone):
        if self._cfg.encoder == "mobilenet":
            from ...networks.mobilenet import Backbone
            dev = self.actor.store.flat.device
            if backbone is None:
                backbone = Backbone.synthetic(0, device=dev)
            self._shared["backbone"] = backbone
            self._native.set_backbone(backbone)
        elif backbone is not None:
            raise ValueError("backbone= is only used with encoder='mobilenet'")
        return self

    @staticmethod
    def _pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding, latent_dim,
                   encoder, hidden_dims, num_qs, critic_layer_norm, pixel_keys, depth_keys):
        if encoder not in ("d4pg", "mobilenet"):    # "resnet": the reference's branch references undefined classes
            raise NotImplementedError(f"encoder={encoder!r}: 'd4pg' (TwoD4PGEncoder) or 'mobilenet' (TwoMobileNetEncoder)")
        if encoder == "d4pg" and cnn_padding != "VALID":
            raise NotImplementedError("only VALID padding (every SERL config)")
        if depth_keys:
            raise NotImplementedError("depth_keys are not used by any SERL config")
        if not is_dict(observation_space):
            raise TypeError("DrQLearner needs a Dict observation space with 'state' and pixel keys")
        pixel_keys = tuple(pixel_keys)
        shapes = {k: tuple(observation_space.spaces[k].shape) for k in pixel_keys}
        H, W, C, S = shapes[pixel_keys[0]]
        assert C == 3 and all(s == (H, W, C, S) for s in shapes.values()), "pixel spaces must be (H, W, 3, stack)"
        extra = set(observation_space.spaces) - set(pixel_keys) - {"state"}
        if extra:
            raise NotImplementedError(f"unsupported extra observation keys {sorted(extra)}")
        kw = dict(action_dim=int(action_space.shape[-1]), hidden_dims=tuple(int(h) for h in hidden_dims),
                  num_qs=int(num_qs), critic_layer_norm=bool(critic_layer_norm), pixel_keys=pixel_keys,
                  state_dim=int(observation_space.spaces["state"].shape[-1]), H=int(H), W=int(W), stack=int(S),
                  latent_dim=int(latent_dim))
        if encoder == "mobilenet":
            # frozen MobileNetV3 features -> SpatialLearnedEmbeddings -> Dropout(0.1) (two_mobilenet_encoder.py:10-45);
            # the backbone takes 3-channel images, i.e. frame stack 1 (every shipped script: FrameStack(env, 1))
            if S != 1:
                raise ValueError("encoder='mobilenet' needs pixel observations of shape (H, W, 3, 1)")
            from ...networks.mobilenet import Backbone
            fh, fw, fc = Backbone.feature_shape(H, W)
            return NetConfig(encoder="mobilenet", feat_h=fh, feat_w=fw, feat_c=fc, **kw)
        return NetConfig(cnn_features=tuple(cnn_features), cnn_filters=tuple(cnn_filters),
                         cnn_strides=tuple(cnn_strides), **kw)

    # ------------------------------------------------------------------ pixel leaves -> device pointers
    def _pixel_sources(self, batch, n, ub: UpdateBatch, keep):
        """Fill ub.pixels (cameras in sorted order).  A leaf is a lazy FrameWindows of a device replay buffer (or the
        packed tensor it was pinned to before an insert overwrote its rows), a packed (n,H,W,3,S+1) array when the key
        is absent from next_observations (`_unpack`), or unpacked obs / next_obs arrays; host arrays are uploaded."""
        cfg = self._cfg
        dev = self.actor.store.flat.device
        H, W, S = cfg.H, cfg.W, cfg.stack
        obs, nxt = batch["observations"], batch["next_observations"]
        lazy = []

        def dev_u8(x, T):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
            t = t.to(dev).contiguous()
            assert t.dtype == torch.uint8 and tuple(t.shape) == (n, H, W, 3, T), f"{tuple(t.shape)} != {(n, H, W, 3, T)}"
            keep.append(t)
            return t.data_ptr()
        for ci, cam in enumerate(cfg.cams):
            src = ub.pixels[ci]
            leaf = obs[cam]
            packed = cam not in nxt
            if isinstance(leaf, FrameWindows) and leaf.packed is not None:
                leaf = leaf.packed            # an insert() hit this batch's rows before the gather: snapshot taken then
            if isinstance(leaf, FrameWindows):
                assert leaf.num_stack == S and len(leaf) == n
                src.kind, src.frames, src.capacity, src.indx = KIND_FRAMES, leaf.frames.data_ptr(), leaf.capacity, \
                    leaf.indx.data_ptr()
                lazy.append(leaf)
            elif packed:
                src.kind, src.obs = KIND_PACKED, dev_u8(leaf, S + 1)
            else:
                src.kind, src.obs, src.next_obs = KIND_UNPACKED, dev_u8(leaf, S), dev_u8(nxt[cam], S)
        return lazy

    # ------------------------------------------------------------------ update
    def update(self, batch, utd_ratio: int, pixel_keys=("pixels",)):
        cfg = self._cfg
        pixel_keys = tuple(pixel_keys)
        if pixel_keys != cfg.pixel_keys and pixel_keys != ("pixels",):
            raise ValueError(f"pixel_keys {pixel_keys} differ from the keys the agent was created with {cfg.pixel_keys}")
        n = len(batch["rewards"])
        assert n % utd_ratio == 0
        dev = self.actor.store.flat.device
        f32 = torch.float32
        # share the critic's encoder with the actor, once per call (drq_learner.py:171-172)
        actor = self.actor.clone()
        nenc = self.critic.store.layout.prefix_size("image_encoder")
        actor.store.flat[:nenc].copy_(self.critic.store.flat[:nenc])
        ub = UpdateBatch()
        keep = [_to_dev(batch["observations"]["state"], dev, f32).reshape(n, cfg.state_dim),
                _to_dev(batch["next_observations"]["state"], dev, f32).reshape(n, cfg.state_dim)]
        ub.state, ub.next_state = keep[0].data_ptr(), keep[1].data_ptr()
        with hold_windows(batch["observations"].values()):      # vs. insert() on a replay thread (device_batch.py)
            lazy = self._pixel_sources(batch, n, ub, keep)
            out = self._run_update(batch, ub, n, utd_ratio, actor, self.rng, keep)
            for fw in lazy:     # every row's gather is enqueued: later inserts (same stream) are ordered behind it
                fw.consumed = True
        return out
