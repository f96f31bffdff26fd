"""DrQ pixel learner on MI355X (mirrors serl/agents/drq/drq_learner.py).

update(batch, utd_ratio, pixel_keys):
  1. `_unpack` of packed S+1 frame stacks          (drq_learner.py:168-169, utils/commons.py:7-29)
  2. critic encoder copied into the actor, once      (:171-172, commons.py:32-47)
  3. random-shift augmentation of obs and next_obs, whole B*U batch, per-camera keys  (:174-177, :131-137)
  4. SACLearner.update on the augmented batch         (:186)
Steps 1 and 3 (and, for device-resident replay batches, the frame-stack gather of
MemoryEfficientReplayBuffer.sample) are ONE HIP kernel per camera and side (gather_crop.hip).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ... import ops, prng
from ...data.device_batch import FrameWindows
from ...networks import modules as nn
from ...networks.spec import NetConfig
from ...spaces import is_dict
from ..sac.sac_learner import SACLearner, _to_dev

PAD = 4   # batched_random_crop(..., padding=4), serl/utils/augmentations.py:44


class DrQLearner(SACLearner):
    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4, critic_lr: float = 3e-4,
               temp_lr: float = 3e-4, cnn_features: Sequence[int] = (32, 32, 32, 32),
               cnn_filters: Sequence[int] = (3, 3, 3, 3), cnn_strides: Sequence[int] = (2, 1, 1, 1),
               cnn_padding: str = "VALID", latent_dim: int = 50, encoder: str = "d4pg",
               hidden_dims: Sequence[int] = (256, 256), discount: float = 0.99, tau: float = 0.005, num_qs: int = 2,
               num_min_qs: Optional[int] = None, critic_dropout_rate: Optional[float] = None,
               critic_layer_norm: bool = False, target_entropy: Optional[float] = None,
               init_temperature: float = 1.0, backup_entropy: bool = True, pixel_keys: Tuple[str, ...] = ("pixels",),
               depth_keys: Tuple[str, ...] = (), device=None, dist=None):
        cfg = cls._pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                             latent_dim, encoder, hidden_dims, num_qs, critic_layer_norm, pixel_keys, depth_keys)
        return cls._build(cfg, seed, actor_lr, critic_lr, temp_lr, discount, tau, num_min_qs, critic_dropout_rate,
                          target_entropy, init_temperature, backup_entropy, device, dist)

    @staticmethod
    def _pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding, latent_dim,
                   encoder, hidden_dims, num_qs, critic_layer_norm, pixel_keys, depth_keys):
        if encoder != "d4pg":
            raise NotImplementedError(f"encoder={encoder!r}: only the d4pg conv encoder is on the hot path")
        if cnn_padding != "VALID":
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
        return NetConfig(action_dim=int(action_space.shape[-1]), hidden_dims=tuple(int(h) for h in hidden_dims),
                        num_qs=int(num_qs), critic_layer_norm=bool(critic_layer_norm), pixel_keys=pixel_keys,
                        state_dim=int(observation_space.spaces["state"].shape[-1]), H=int(H), W=int(W), stack=int(S),
                        cnn_features=tuple(cnn_features), cnn_filters=tuple(cnn_filters),
                        cnn_strides=tuple(cnn_strides), latent_dim=int(latent_dim))

    # ------------------------------------------------------------------ augmentation + unpack + gather
    def _augment(self, batch, n, utd_ratio, key_obs, key_next, ws):
        """Returns ({'obs'|'next': [u8 (n,H,W,3S) device tensor per camera in cfg.cams order]}, crop) where
        crop(lo, rows, stream) launches the fused gather + _unpack + random-shift kernels for rows [lo, lo+rows)
        of every (side, camera); the per-row offsets are computed here, for all rows.
        data_augmentation_fn: per pixel key (caller order) key, rng = split(rng); crop with `key`."""
        cfg = self._cfg
        dev = self.actor.store.flat.device
        st = ops.current_stream()
        H, W, S = cfg.H, cfg.W, cfg.stack
        dist = self._dist
        world = dist.world_size if dist is not None else 1
        rank = dist.rank if dist is not None else 0
        out = {"obs": {}, "next": {}}
        jobs, lazy = [], []
        obs, nxt = batch["observations"], batch["next_observations"]
        img = H * W * 3 * S
        for side, rng, t0 in (("obs", key_obs, 0), ("next", key_next, 1)):
            for cam in cfg.pixel_keys:                                   # augmentation key order = pixel_keys
                key, rng = prng.split(rng)
                off = ws.tensor(f"off_{side}_{cam}", (n, 2), torch.int32)
                if world == 1:
                    ops.crop_offsets(key, n, 0, n, PAD, off.data_ptr(), st)
                else:                                                    # global-index keyed
                    U = utd_ratio
                    B = n // U
                    for i in range(U):
                        ops.crop_offsets(key, B, i * B * world + rank * B, n * world, PAD,
                                         off.data_ptr() + 8 * i * B, st)
                dst = ws.tensor(f"aug_{side}_{cam}", (n, H, W, 3 * S), torch.uint8)
                packed = cam not in nxt
                src = obs[cam] if (side == "obs" or packed) else nxt[cam]
                if isinstance(src, FrameWindows) and src.packed is not None:
                    src = src.packed              # an insert() hit this batch's rows before the gather: snapshot taken then
                if isinstance(src, FrameWindows):
                    lazy.append(src)

                    def job(lo, rows, stream, src=src, off=off, dst=dst, t0=t0):
                        ops.gather_crop_frames(src.frames.data_ptr(), src.capacity, src.indx.data_ptr() + 8 * lo, S + 1,
                                               t0, off.data_ptr() + 8 * lo, dst.data_ptr() + lo * img, rows, H, W, S,
                                               PAD, stream)
                else:
                    T = S + 1 if packed else S
                    if isinstance(src, torch.Tensor):
                        s_dev = src.to(dev).contiguous()
                    else:
                        s_dev = ws.tensor(f"h2d_{side if not packed else 'pk'}_{cam}", (n, H, W, 3, T), torch.uint8)
                        if side == "obs" or not packed:
                            s_dev.copy_(torch.from_numpy(np.ascontiguousarray(src)))
                    assert tuple(s_dev.shape) == (n, H, W, 3, T), f"{cam}: {tuple(s_dev.shape)} != {(n, H, W, 3, T)}"

                    def job(lo, rows, stream, s_dev=s_dev, off=off, dst=dst, T=T, t0=(t0 if packed else 0)):
                        ops.gather_crop_packed(s_dev.data_ptr() + lo * H * W * 3 * T, T, t0, off.data_ptr() + 8 * lo,
                                               dst.data_ptr() + lo * img, rows, H, W, S, PAD, stream)
                jobs.append(job)
                out[side][cam] = dst

        def crop(lo, rows, stream):
            for job in jobs:
                job(lo, rows, stream)
            if lo + rows >= n:       # every row's gather is enqueued: later inserts are ordered behind it by the stream
                for fw in lazy:
                    fw.consumed = True
        return out, crop

    def _prepare(self, batch, n):
        raise RuntimeError("DrQLearner prepares features inside update()")

    def _features(self, passes, feats, lo, B, ws, st, save_tag=None, trunk_streams=None, fork=None, prep=True,
                  z_ptrs=None):
        cfg = self._cfg
        ldz = cfg.latent + cfg.action_dim
        img_bytes = cfg.H * cfg.W * 3 * cfg.stack
        enc_passes = []
        for tag, store, which in passes:
            imgs = [feats["aug"][which][cam].data_ptr() + lo * img_bytes for cam in cfg.cams]
            enc_passes.append((tag, store, imgs))
        nn.encoder_fwd(cfg, enc_passes, B, ws, st, mask_tag=save_tag, prep=prep)
        if fork is not None:          # the trunks of the passes are independent: side streams wait for the encoders
            fork.record(torch.cuda.current_stream())
            for s_ in trunk_streams.values():
                s_.wait_event(fork)
        for tag, store, which in passes:
            state = feats["next_state" if which == "next" else "state"]
            st_ = trunk_streams[tag].cuda_stream if trunk_streams and tag in trunk_streams else st
            z = z_ptrs[tag] if z_ptrs and tag in z_ptrs else ws.ptr(f"z{tag}", (B, ldz))
            nn.trunk_fwd(cfg, tag, store, state.data_ptr() + 4 * lo * cfg.state_dim, z, ldz, B, ws, st_,
                         save=(tag == save_tag))

    def _features_bwd(self, tag, store, grads, feats, lo, B, ws, st, dz_ptr, lddz, need_encoder, gside=None,
                      dz_slabs=(1, 0)):
        cfg = self._cfg
        ldz = cfg.latent + cfg.action_dim
        which = "obs"
        state = feats["state"]
        nn.trunk_bwd(cfg, tag, store, grads, state.data_ptr() + 4 * lo * cfg.state_dim, ws.ptr(f"z{tag}", (B, ldz)),
                     ldz, dz_ptr, lddz, B, ws, st, need_denc=need_encoder, gside=gside, dz_slabs=dz_slabs)
        if need_encoder:
            img_bytes = cfg.H * cfg.W * 3 * cfg.stack
            imgs = [feats["aug"][which][cam].data_ptr() + lo * img_bytes for cam in cfg.cams]
            nn.encoder_bwd(cfg, tag, store, grads, imgs, B, ws, st, gside=gside, use_masks=True)

    # ------------------------------------------------------------------ update
    def _aug_keys(self):
        """(key for obs, key for next_obs, rng the SAC update starts from).
        rng, key = split(rng) -- note: rng is child 0 here (drq_learner.py:174,176)."""
        rng, key_obs = prng.split(self.rng)
        rng, key_next = prng.split(rng)
        return key_obs, key_next, rng

    def update(self, batch, utd_ratio: int, pixel_keys=("pixels",)):
        cfg = self._cfg
        pixel_keys = tuple(pixel_keys)
        if pixel_keys != cfg.pixel_keys and pixel_keys != ("pixels",):
            raise ValueError(f"pixel_keys {pixel_keys} differ from the keys the agent was created with {cfg.pixel_keys}")
        n = len(batch["rewards"])
        assert n % utd_ratio == 0
        dev = self.actor.store.flat.device
        f32 = torch.float32
        ws = self._ws(n // utd_ratio)
        # share the critic's encoder with the actor, once per call (drq_learner.py:171-172)
        actor = self.actor.clone()
        nenc = self.critic.store.layout.prefix_size("image_encoder")
        actor.store.flat[:nenc].copy_(self.critic.store.flat[:nenc])
        key_obs, key_next, rng = self._aug_keys()
        aug, crop = self._augment(batch, n, utd_ratio, key_obs, key_next, ws)
        feats = dict(aug=aug, crop=crop,
                     state=_to_dev(batch["observations"]["state"], dev, f32).reshape(n, cfg.state_dim),
                     next_state=_to_dev(batch["next_observations"]["state"], dev, f32).reshape(n, cfg.state_dim))
        return self.replace(rng=rng)._update_prepared(batch, feats, n, utd_ratio, actor)
