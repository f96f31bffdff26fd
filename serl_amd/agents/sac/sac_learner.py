"""SAC / REDQ / RLPD learner on MI355X (mirrors serl/agents/sac/sac_learner.py).

`create` / `update(batch, utd_ratio)` keep the reference signatures and semantics:
  * UTD loop over contiguous minibatch slices, critic step each, then ONE actor step and ONE
    temperature step on the last minibatch (sac_learner.py:221-238)
  * critic step: a' ~ pi(s'), REDQ subsample of the target ensemble, min-Q TD target (optional entropy
    backup), MSE over ensemble x batch, Adam, Polyak on all critic params (sac_learner.py:160-219)
  * actor step: mean-over-ensemble Q, alpha*logp - Q (sac_learner.py:120-144); temperature (146-158)
  * RNG: the jax split chain of SURVEY.md appendix B
All of it -- every kernel launch, stream, event, rng key, REDQ member stride and Adam bias correction of a call -- is
produced by the native step driver (serl_amd/csrc/driver.hip) behind ONE C-ABI call, `serl_learner_update`; this module
holds the agent's device buffers, clones them (functional semantics) and passes pointers.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Optional, Sequence

import numpy as np
import torch

from ... import ops, prng
from ...native import AgentBuffers, NativeLearner, UpdateBatch
from ...networks.params import ParamStore
from ...networks.spec import NetConfig, actor_layout, critic_layout, init_tree
from ..agent import Agent
from ..train_state import TempState, TrainState


def _device(device):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("serl_amd needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(x, dev, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=None)).to(dev).to(dtype).contiguous()


class UpdateInfo(Mapping):
    """The logged scalars of one update() call.  Like the jax Arrays the reference returns (sac_learner.py:238), the
    values are device results of asynchronously enqueued work: update() returns without waiting for the GPU and the
    first access (indexing, iteration over items, comparison, dict(info)) copies the 8 floats to the host."""
    KEYS = ("actor_loss", "entropy", "critic_loss", "q", "temperature", "temperature_loss")

    def __init__(self, dev_scalars, global_batch, num_qs, keys=None, extra=None):
        self._dev, self._bg, self._e, self._host = dev_scalars, global_batch, num_qs, None
        self._keys = tuple(keys) if keys is not None else self.KEYS
        self._extra = extra or {}        # key -> (1-element device tensor, scale): e.g. VICELearner's "vice_rews"

    def _materialize(self):
        if self._host is None:
            h = self._dev.cpu().numpy()                      # one small D2H per update call, on first access
            bg, e = self._bg, self._e
            full = {
                "actor_loss": float(h[2]) / bg, "entropy": float(h[6]),
                "critic_loss": float(h[0]) / (e * bg), "q": float(h[1]) / (e * bg),
                "temperature": float(h[4]), "temperature_loss": float(h[5]),
            }
            for k, (t, scale) in self._extra.items():
                full[k] = float(t.cpu().numpy().reshape(-1)[0]) * scale
            self._host = {k: full[k] for k in self._keys}
            self._dev, self._extra = None, {}
        return self._host

    def __getitem__(self, k):
        return self._materialize()[k]

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __repr__(self):
        return repr(self._materialize())


class _WorkspaceView:
    """Read access to the driver's workspace of the last update at minibatch size B (tests, tools):
    `tensor("aug_obs_<camera>" | "aug_next_<camera>", shape, dtype)` returns a copy of the augmented images."""

    def __init__(self, native: NativeLearner, cfg: NetConfig, B: int):
        self._n, self._cfg, self._B = native, cfg, B

    def tensor(self, name, shape, dtype=torch.float32):
        if name.startswith("aug_"):
            _, side, cam = name.split("_", 2)
            name = f"aug_{0 if side == 'obs' else 1}_{self._cfg.cams.index(cam)}"
        return self._n.read_buffer(self._B, name, shape, dtype)


class SACLearner(Agent):
    """State-based SAC learner. Fields as in the reference dataclass (sac_learner.py:19-30)."""
    _HAS_TEMP = True            # DDPGLearner: no entropy term, no temperature step
    _INFO_KEYS = UpdateInfo.KEYS
    _AUG_MODE = 0               # pixel learners: 1 DrQ key order, 2 PixelDDPG key order (driver.hip)

    # ------------------------------------------------------------------ construction
    @classmethod
    def create(cls, seed: int, observation_space, action_space, actor_lr: float = 3e-4, critic_lr: float = 3e-4,
               temp_lr: float = 3e-4, hidden_dims: Sequence[int] = (256, 256), discount: float = 0.99,
               tau: float = 0.005, num_qs: int = 2, num_min_qs: Optional[int] = None,
               critic_dropout_rate: Optional[float] = None, critic_layer_norm: bool = False,
               target_entropy: Optional[float] = None, init_temperature: float = 1.0, backup_entropy: bool = True,
               device=None, dist=None):
        action_dim = int(action_space.shape[-1])
        cfg = NetConfig(action_dim=action_dim, hidden_dims=tuple(int(h) for h in hidden_dims), num_qs=int(num_qs),
                        critic_layer_norm=bool(critic_layer_norm), obs_dim=int(observation_space.shape[-1]))
        return cls._build(cfg, seed, actor_lr, critic_lr, temp_lr, discount, tau, num_min_qs, critic_dropout_rate,
                          target_entropy, init_temperature, backup_entropy, device, dist)

    @classmethod
    def _build(cls, cfg, seed, actor_lr, critic_lr, temp_lr, discount, tau, num_min_qs, critic_dropout_rate,
               target_entropy, init_temperature, backup_entropy, device, dist, nkeys=4):
        if critic_dropout_rate is not None and not 0.0 <= float(critic_dropout_rate) < 1.0:
            raise ValueError("critic_dropout_rate must be in [0, 1)")
        dev = _device(device)
        if target_entropy is None:
            target_entropy = -cfg.action_dim / 2                        # sac_learner.py:60-61
        rng = prng.PRNGKey(seed)
        rng, actor_key, critic_key = prng.split(rng, nkeys)[:3]         # :63-64 (4 keys; DDPG: 3, ddpg_learner.py:47)
        al, cl = actor_layout(cfg), critic_layout(cfg)
        actor_store = ParamStore.from_tree(al, init_tree(al, [int(actor_key[0]), int(actor_key[1])]), dev)
        critic_store = ParamStore.from_tree(cl, init_tree(cl, [int(critic_key[0]), int(critic_key[1])]), dev)
        self = cls.__new__(cls)
        self.rng = rng
        self.actor = TrainState.create(actor_store, actor_lr)
        self.critic = TrainState.create(critic_store, critic_lr)
        self.target_critic = TrainState.create(critic_store.clone(), None)     # same initial params (:92-96)
        self.temp = TempState.create(init_temperature, temp_lr, dev)
        self.tau, self.discount, self.target_entropy = float(tau), float(discount), float(target_entropy)
        self.num_qs, self.num_min_qs, self.backup_entropy = cfg.num_qs, num_min_qs, bool(backup_entropy)
        self.critic_dropout_rate = critic_dropout_rate
        self._cfg = cfg
        with torch.cuda.device(dev):
            native = NativeLearner(cfg, num_min_qs=num_min_qs, backup_entropy=backup_entropy,
                                   has_temperature=cls._HAS_TEMP, discount=discount, tau=tau,
                                   target_entropy=target_entropy, actor_lr=actor_lr, critic_lr=critic_lr, temp_lr=temp_lr,
                                   critic_dropout_rate=critic_dropout_rate)
        # shared by every agent copy derived from this one: the driver handle and the (reused) gradient buffers
        self._shared = {"dist": dist, "native": native, "critic_grads": critic_store.zeros_like(),
                        "actor_grads": actor_store.zeros_like()}
        return self

    # ------------------------------------------------------------------ helpers
    @property
    def _dist(self):
        return self._shared.get("dist")

    @property
    def _native(self) -> NativeLearner:
        return self._shared["native"]

    def _ws(self, B) -> _WorkspaceView:
        return _WorkspaceView(self._native, self._cfg, B)

    # ------------------------------------------------------------------ update
    def update(self, batch, utd_ratio: int):
        n = len(batch["rewards"])
        assert n % utd_ratio == 0                                         # sac_learner.py:228
        dev = self.actor.store.flat.device
        f32 = torch.float32
        ub = UpdateBatch()
        keep = [_to_dev(batch["observations"], dev, f32).reshape(n, -1),
                _to_dev(batch["next_observations"], dev, f32).reshape(n, -1)]
        ub.obs, ub.next_obs = keep[0].data_ptr(), keep[1].data_ptr()
        return self._run_update(batch, ub, n, utd_ratio, self.actor.clone(), self.rng, keep)

    def _run_update(self, batch, ub: UpdateBatch, n, utd_ratio, actor, rng, keep, features_hook=None, extra_info=None):
        """Clone the state (functional semantics: `self` stays valid and unchanged), hand the pointers to the driver."""
        cfg = self._cfg
        dev = self.actor.store.flat.device
        f32 = torch.float32
        dist = self._dist
        world = dist.world_size if dist is not None else 1
        A = cfg.action_dim
        actions = _to_dev(batch["actions"], dev, f32).reshape(n, A)
        rewards = _to_dev(batch["rewards"], dev, f32).reshape(n)
        masks = _to_dev(batch["masks"], dev, f32).reshape(n)
        ub.n, ub.aug_mode = n, self._AUG_MODE
        ub.actions, ub.rewards, ub.masks = actions.data_ptr(), rewards.data_ptr(), masks.data_ptr()
        critic, target, temp = self.critic.clone(), self.target_critic.clone(), self.temp.clone()
        cgr, agr = self._shared["critic_grads"], self._shared["actor_grads"]
        info = torch.zeros(8, dtype=f32, device=dev)
        ab = AgentBuffers()
        ab.actor, ab.actor_mu, ab.actor_nu = actor.store.base, actor.mu.data_ptr(), actor.nu.data_ptr()
        ab.critic, ab.critic_mu, ab.critic_nu = critic.store.base, critic.mu.data_ptr(), critic.nu.data_ptr()
        ab.target, ab.temp = target.store.base, temp.buf.data_ptr()
        ab.critic_grads, ab.actor_grads = cgr.base, agr.base
        ab.actor_step, ab.critic_step, ab.temp_step = actor.step, critic.step, temp.step
        nenc = self.critic.store.layout.prefix_size("image_encoder") if cfg.pixel else 0    # data-parallel buckets (native.py)
        with torch.cuda.device(dev):
            new_rng = self._native.update(ab, ub, utd_ratio, rng, info.data_ptr(), ops.current_stream(), dist=dist,
                                          reduce_tensors=(cgr.flat, agr.flat, info[:4], None, cgr.flat[nenc:], cgr.flat[:nenc]),
                                          features_hook=features_hook)
        critic.step += utd_ratio
        actor.step += 1
        if self._HAS_TEMP:
            temp.step += 1
        del keep, actions, rewards, masks      # (stream-ordered frees: the caching allocator reuses them on this stream)
        new_agent = self.replace(actor=actor, critic=critic, target_critic=target, temp=temp, rng=new_rng)
        return new_agent, UpdateInfo(info, n // utd_ratio * world, cfg.num_qs, self._INFO_KEYS, extra_info)
