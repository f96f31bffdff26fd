"""PixelBCLearner on MI355X (mirrors serl/agents/bc/pixel_bc_learner.py:26-139).

actor = PixelMultiplexer(encoder, TanhNormal(MLP), stop_gradient=False): behaviour cloning trains the image encoder
through the policy loss, so the update is the one place where the *actor's* conv stack is differentiated -- the native
driver reuses the critic's encoder / trunk backward for it (serl_learner_bc_update, csrc/driver.hip).

update(batch, pixel_keys):
  1. `_unpack` when the pixel keys are missing from next_observations   (:108-109; only the observations are used)
  2. key, rng = split(self.rng); random-shift augmentation of the observations with `key`   (:111-112, :90-96)
  3. actions clipped to +-(1 - 1e-5)   (:114-121)
  4. `rng, key1 = jax.random.split(self.rng, 3)`   (:123)  -- in the reference this line unpacks three keys into two
     names and raises ValueError, i.e. PixelBCLearner.update cannot run in the snapshot.  BUILD-DEFINED here as the
     evident intent, the line PixelHybridBCLearner.update has at :154: rng = child 0, key1 = child 1 of the 3-way split
     of the rng the call STARTED with (the augmentation's split is not carried over, as in the hybrid learner).
  5. nll = -mean log_prob(actions) under the policy applied with training=True, rngs={"dropout": key1}; Adam   (:125-137)
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch

from ... import ops, prng
from ...native import BcBuffers, NativeLearner, UpdateBatch
from ...networks.params import ParamStore
from ...networks.spec import actor_layout, critic_layout, init_tree
from ..agent import Agent
from ..drq.drq_learner import DrQLearner
from ..sac.sac_learner import UpdateInfo, _device, _to_dev
from ..train_state import TrainState


class BcInfo(UpdateInfo):
    """{"nll": ...} (and "bce" for the hybrid learner): device sums divided by the batch size on first access."""

    def __init__(self, dev_scalars, n, keys):
        self._dev, self._n, self._keys, self._host = dev_scalars, n, tuple(keys), None

    def _materialize(self):
        if self._host is None:
            h = self._dev.cpu().numpy()
            full = {"nll": float(h[0]) / self._n, "bce": float(h[1]) / self._n}
            self._host = {k: full[k] for k in self._keys}
            self._dev = None
        return self._host


class PixelBCLearner(Agent):
    _HYBRID = False
    _INFO_KEYS = ("nll",)

    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4,
               cnn_features: Sequence[int] = (32, 32, 32, 32), cnn_filters: Sequence[int] = (3, 3, 3, 3),
               cnn_strides: Sequence[int] = (2, 1, 1, 1), cnn_padding: str = "VALID", latent_dim: int = 50,
               encoder: str = "d4pg", hidden_dims: Sequence[int] = (256, 256), pixel_keys: Tuple[str, ...] = ("pixels",),
               depth_keys: Tuple[str, ...] = (), device=None, backbone=None):
        return cls._create(seed, observation_space, action_space, actor_lr, cnn_features, cnn_filters, cnn_strides,
                           cnn_padding, latent_dim, encoder, hidden_dims, pixel_keys, depth_keys, device, backbone)

    @classmethod
    def _create(cls, seed, observation_space, action_space, actor_lr, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                latent_dim, encoder, hidden_dims, pixel_keys, depth_keys, device, backbone):
        from ...networks.spec import discrete_actor_layout
        cfg = DrQLearner._pixel_cfg(observation_space, action_space, cnn_features, cnn_filters, cnn_strides, cnn_padding,
                                    latent_dim, encoder, hidden_dims, 1, False, pixel_keys, depth_keys)
        full_dim = cfg.action_dim
        if cls._HYBRID:      # the last action dimension is the gripper's discrete action (pixel_hybrid_bc_learner.py:52)
            import dataclasses
            cfg = dataclasses.replace(cfg, action_dim=full_dim - 1)
            if cfg.action_dim < 1:
                raise ValueError("PixelHybridBCLearner needs at least one continuous action dimension")
        dev = _device(device)
        rng = prng.PRNGKey(seed)
        keys = prng.split(rng, 3 if cls._HYBRID else 2)       # pixel_bc_learner.py:56 / pixel_hybrid_bc_learner.py:57
        rng, actor_key = keys[0], keys[1]
        al = actor_layout(cfg)
        self = cls.__new__(cls)
        self.rng = rng
        self.actor = TrainState.create(ParamStore.from_tree(al, init_tree(al, [int(actor_key[0]), int(actor_key[1])]), dev),
                                       actor_lr)
        self._cfg, self._full_action_dim = cfg, full_dim
        with torch.cuda.device(dev):
            native = NativeLearner(cfg, num_min_qs=None, backup_entropy=False, has_temperature=False, discount=0.0,
                                   tau=0.0, target_entropy=0.0, actor_lr=actor_lr, critic_lr=actor_lr, temp_lr=0.0,
                                   critic_layout_fn=discrete_actor_layout if cls._HYBRID else critic_layout,
                                   critic_no_action=cls._HYBRID)
        self._shared = {"native": native, "actor_grads": self.actor.store.zeros_like()}
        if cls._HYBRID:
            dk = keys[2]
            dl = discrete_actor_layout(cfg)
            self.discrete_actor = TrainState.create(
                ParamStore.from_tree(dl, init_tree(dl, [int(dk[0]), int(dk[1])]), dev), actor_lr)
            self._shared["disc_grads"] = self.discrete_actor.store.zeros_like()   # encoder leaves stay zero (stop_gradient)
        if cfg.encoder == "mobilenet":
            from ...networks.mobilenet import Backbone
            if backbone is None:
                backbone = Backbone.synthetic(0, device=dev)
            self._shared["backbone"] = backbone
            native.set_backbone(backbone)
        elif backbone is not None:
            raise ValueError("backbone= is only used with encoder='mobilenet'")
        return self

    # ------------------------------------------------------------------ update
    def update(self, batch, pixel_keys=("pixels",)):
        return self._bc_update(batch, pixel_keys)

    def _bc_update(self, batch, pixel_keys):
        cfg = self._cfg
        pixel_keys = tuple(pixel_keys)
        if pixel_keys != cfg.pixel_keys and pixel_keys != ("pixels",):
            raise ValueError(f"pixel_keys {pixel_keys} differ from the keys the agent was created with {cfg.pixel_keys}")
        dev = self.actor.store.flat.device
        f32 = torch.float32
        ld = self._full_action_dim
        actions = _to_dev(batch["actions"], dev, f32).reshape(-1, ld)
        n = actions.shape[0]
        key_aug, _ = prng.split(self.rng)                        # key, rng = split(rng): the split rng is dropped
        k3 = prng.split(self.rng, 3)
        new_rng, key1, key2 = k3[0], k3[1], k3[2]
        ub = UpdateBatch()
        ub.n, ub.aug_mode = n, 1
        state = _to_dev(batch["observations"]["state"], dev, f32).reshape(n, cfg.state_dim)
        keep = [state, actions]
        ub.state, ub.next_state, ub.actions = state.data_ptr(), state.data_ptr(), actions.data_ptr()
        # only the observations are read: a packed (S + 1)-frame leaf is addressed as its first S frames (`_unpack`)
        from ...data.device_batch import hold_windows
        with hold_windows(batch["observations"].values()):      # vs. insert() on a replay thread (device_batch.py)
            lazy = DrQLearner._pixel_sources(self, batch, n, ub, keep)
            actor = self.actor.clone()
            bb = BcBuffers()
            bb.actor, bb.actor_mu, bb.actor_nu = actor.store.base, actor.mu.data_ptr(), actor.nu.data_ptr()
            bb.actor_grads, bb.actor_step = self._shared["actor_grads"].base, actor.step
            fields = dict(actor=actor, rng=new_rng)
            if self._HYBRID:
                disc = self.discrete_actor.clone()
                bb.disc, bb.disc_mu, bb.disc_nu = disc.store.base, disc.mu.data_ptr(), disc.nu.data_ptr()
                bb.disc_grads, bb.disc_step = self._shared["disc_grads"].base, disc.step
                fields["discrete_actor"] = disc
            info = torch.zeros(2, dtype=f32, device=dev)
            with torch.cuda.device(dev):
                self._shared["native"].bc_update(bb, ub, ld, key_aug, key1, key2, info.data_ptr(), ops.current_stream())
            actor.step += 1
            if self._HYBRID:
                fields["discrete_actor"].step += 1
            for fw in lazy:
                fw.consumed = True
        del keep
        return self.replace(**fields), BcInfo(info, n, self._INFO_KEYS)

    def _ws(self, B):
        from ..sac.sac_learner import _WorkspaceView
        return _WorkspaceView(self._shared["native"], self._cfg, B)
