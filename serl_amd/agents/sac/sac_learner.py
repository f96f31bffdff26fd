"""SAC / REDQ / RLPD learner on MI355X (mirrors serl/agents/sac/sac_learner.py).

`create` / `update(batch, utd_ratio)` keep the reference signatures and semantics:
  * UTD loop over contiguous minibatch slices, critic step each, then ONE actor step and ONE
    temperature step on the last minibatch (sac_learner.py:221-238)
  * critic step: a' ~ pi(s'), REDQ subsample of the target ensemble, min-Q TD target (optional entropy
    backup), MSE over ensemble x batch, Adam, Polyak on all critic params (sac_learner.py:160-219)
  * actor step: mean-over-ensemble Q, alpha*logp - Q (sac_learner.py:120-144); temperature (146-158)
  * RNG: the jax split chain of SURVEY.md appendix B, evaluated on the host (native threefry)
All tensor math runs in hand-written HIP kernels (serl_amd.ops); Python only sequences launches.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Optional, Sequence

import numpy as np
import torch

from ... import ops, prng
from ...frozen_dict import FrozenDict
from ...networks import modules as nn
from ...networks.modules import Workspace
from ...networks.params import ParamStore
from ...networks.spec import NetConfig, actor_layout, critic_head_prefix, critic_layout, init_tree
from ..agent import Agent
from ..train_state import ADAM_B1, ADAM_B2, ADAM_EPS, TempState, TrainState, bias_corrections


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

    def __init__(self, dev_scalars, global_batch, num_qs, keys=None):
        self._dev, self._bg, self._e, self._host = dev_scalars, global_batch, num_qs, None
        self._keys = tuple(keys) if keys is not None else self.KEYS

    def _materialize(self):
        if self._host is None:
            h = self._dev.cpu().numpy()                      # one small D2H per update call, on first access
            bg, e = self._bg, self._e
            full = {
                "actor_loss": float(h[2]) / bg, "entropy": float(h[6]),
                "critic_loss": float(h[0]) / (e * bg), "q": float(h[1]) / (e * bg),
                "temperature": float(h[4]), "temperature_loss": float(h[5]),
            }
            self._host = {k: full[k] for k in self._keys}
            self._dev = None
        return self._host

    def __getitem__(self, k):
        return self._materialize()[k]

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __repr__(self):
        return repr(self._materialize())


class SACLearner(Agent):
    _HAS_TEMP = True            # DDPGLearner: no entropy term, no temperature step
    _INFO_KEYS = UpdateInfo.KEYS

    """State-based SAC learner. Fields as in the reference dataclass (sac_learner.py:19-30)."""

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
        if critic_dropout_rate:
            raise NotImplementedError("critic dropout is not used by any SERL config and is not implemented")
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
        self._cfg = cfg
        self._shared = {"dist": dist}
        return self

    # ------------------------------------------------------------------ helpers
    @property
    def _dist(self):
        return self._shared.get("dist")

    def _ws(self, B) -> Workspace:
        key = ("ws_update", B)
        ws = self._shared.get(key)
        if ws is None:
            ws = self._shared[key] = Workspace(self.actor.store.flat.device)
        return ws

    def _grads(self, name, like: ParamStore) -> ParamStore:
        g = self._shared.get(name)
        if g is None:
            g = self._shared[name] = like.zeros_like()
        return g

    def _side_streams(self):
        """Two side HIP streams + reusable events: the target and critic forward chains of a critic step are
        independent of the actor chain until the TD target, and each is a string of small latency-bound launches."""
        ss = self._shared.get("side_streams")
        if ss is None:
            ss = self._shared["side_streams"] = dict(
                s=[torch.cuda.Stream(), torch.cuda.Stream()],
                fork=torch.cuda.Event(), join=[torch.cuda.Event(), torch.cuda.Event()],
                actor=torch.cuda.Stream(), astart=torch.cuda.Event(), agate=torch.cuda.Event(), aobs=torch.cuda.Event(), adone=[], cdone=[])
        return ss

    def _target_members(self, key):
        """REDQ subsample (serl/networks/ensemble.py:25-37) -> (first member, member stride, count)."""
        E = self.num_qs
        if self.num_min_qs is None:
            return 0, 1, E, None
        idx = prng.choice_no_replace(key, E, int(self.num_min_qs))
        if len(idx) == 1:
            return int(idx[0]), 1, 1, idx
        d = int(idx[1]) - int(idx[0])
        if all(int(idx[i + 1]) - int(idx[i]) == d for i in range(len(idx) - 1)):
            return int(idx[0]), d, len(idx), idx
        return None, None, len(idx), idx

    def _gathered_target(self, idx, src):
        """General num_min_qs > 2: copy the chosen members of `src` into a scratch store (rare path)."""
        cfg = self._cfg
        scratch = self._shared.get("tgt_scratch")
        if scratch is None:
            scratch = self._shared["tgt_scratch"] = src.zeros_like()
        pre = critic_head_prefix(cfg)
        st = ops.current_stream()
        for path, (off, shape) in src.layout.entries.items():
            if path[:len(pre)] != pre:
                continue
            per = int(np.prod(shape[1:]))
            for j, m in enumerate(idx):
                ops.copy2d(scratch.base + 4 * (off + j * per), per, src.base + 4 * (off + int(m) * per), per, 1, per, st)
        return scratch

    # ------------------------------------------------------------------ feature stage (overridden for pixels)
    def _prepare(self, batch, n):
        dev = self.actor.store.flat.device
        f32 = torch.float32
        return dict(obs=_to_dev(batch["observations"], dev, f32).reshape(n, -1),
                    next_obs=_to_dev(batch["next_observations"], dev, f32).reshape(n, -1))

    def _features(self, passes, feats, lo, B, ws, st, save_tag=None, trunk_streams=None, fork=None, prep=True,
                  z_ptrs=None):
        """passes: list of (tag, store, which) -> fills z_<tag>[:, :latent] (or z_ptrs[tag]). State-based: a strided copy.
        trunk_streams: {tag: torch stream} for passes whose per-pass work may run on a side stream; `fork` is
        the event the side streams wait on (recorded here, after everything they depend on)."""
        if fork is not None:
            fork.record(torch.cuda.current_stream())
            for s_ in trunk_streams.values():
                s_.wait_event(fork)
        cfg = self._cfg
        ldz = cfg.latent + cfg.action_dim
        for tag, _store, which in passes:
            src = feats["next_obs" if which == "next" else "obs"]
            st_ = trunk_streams[tag].cuda_stream if trunk_streams and tag in trunk_streams else st
            z = z_ptrs[tag] if z_ptrs and tag in z_ptrs else ws.ptr(f"z{tag}", (B, ldz))
            ops.copy2d(z, ldz, src.data_ptr() + 4 * lo * cfg.obs_dim, cfg.obs_dim, B, cfg.obs_dim, st_)

    def _features_bwd(self, tag, store, grads, feats, lo, B, ws, st, dz_ptr, lddz, need_encoder, gside=None,
                      dz_slabs=(1, 0)):
        pass                               # no trainable feature stage for state inputs

    # ------------------------------------------------------------------ update
    def update(self, batch, utd_ratio: int):
        n = len(batch["rewards"])
        assert n % utd_ratio == 0                                         # sac_learner.py:228
        feats = self._prepare(batch, n)
        return self._update_prepared(batch, feats, n, utd_ratio, self.actor.clone())

    def _update_prepared(self, batch, feats, n, utd_ratio, actor):
        cfg = self._cfg
        dev = self.actor.store.flat.device
        f32 = torch.float32
        B = n // utd_ratio
        A = cfg.action_dim
        ldz = cfg.latent + A
        E = cfg.num_qs
        dist = self._dist
        world = dist.world_size if dist is not None else 1
        rank = dist.rank if dist is not None else 0
        Bg = B * world
        st = ops.current_stream()
        ws = self._ws(B)

        actions = _to_dev(batch["actions"], dev, f32).reshape(n, A)
        rewards = _to_dev(batch["rewards"], dev, f32).reshape(n)
        masks = _to_dev(batch["masks"], dev, f32).reshape(n)

        # functional semantics: step fresh copies (`actor` is already one), leave `self` untouched
        critic, target, temp = self.critic.clone(), self.target_critic.clone(), self.temp.clone()
        cgr = self._grads("critic_grads", critic.store)
        agr = self._grads("actor_grads", actor.store)
        info = ws.tensor("info", (8,))
        zA, zT, zC = (ws.ptr(f"z{t}", (B, ldz)) for t in "ATC")
        log_temp = temp.buf.data_ptr()
        rng = self.rng

        # rng chain of the UTD loop (sac_learner.py:166-209): four splits per critic step, known up front
        step_keys = []
        for i in range(utd_ratio):
            key_act, rng = prng.split(rng)
            key_sub, rng = prng.split(rng)
            _, rng = prng.split(rng)                       # target dropout key (consumed, unused)
            _, rng = prng.split(rng)                       # critic dropout key
            step_keys.append((key_act, key_sub))
        key_pi, rng = prng.split(rng)                      # actor step (sac_learner.py:121-122): policy sample key
        _, rng = prng.split(rng)                           # ... and its critic dropout key
        ss = self._side_streams()
        main = torch.cuda.current_stream()
        s0, s1 = ss["s"]
        zT_all = nlogp_all = None
        zP = ws.ptr("zC_pi", (B, ldz))                               # critic input of the actor step
        # pixel learners: fused gather + crop of all B*U rows up front (feats["crop"])
        crop = feats.get("crop")
        if crop is not None:
            crop(0, n, st)
        if True:
            # The actor's parameters do not change inside the UTD loop (its step comes after it, :235), so its
            # pass over next_obs -- encoder, trunk, policy sample: a third of the forward conv work -- does not
            # depend on the critic steps.  It runs for all U minibatches on its own stream, ahead of the loop, and
            # fills the HBM bandwidth the latency-bound head / TD / dense-backward phases of the main stream leave idle.
            astream, adone, cdone = ss["actor"], ss["adone"], ss["cdone"]
            while len(adone) < utd_ratio:
                adone.append(torch.cuda.Event())
                cdone.append(torch.cuda.Event())
            ss["astart"].record(main)
            astream.wait_event(ss["astart"])
            # per-step input buffers of the target ensemble: the run-ahead pass writes the next action straight into
            # the action columns of step j's buffer, the target trunk of step j fills the latent columns later
            zT_all = ws.ptr("zT_all", (utd_ratio, B, ldz))
            nlogp_all = ws.ptr("nlogp_all", (utd_ratio, B))
            sta = astream.cuda_stream

            def enqueue_actor(j):
                # the actor's parameters are constant during the loop: its S-format weight copies are made once
                self._features([("A", actor.store, "next")], feats, j * B, B, ws, sta, prep=(j == 0))
                nn.policy_fwd(cfg, "A", actor.store, zA, ldz, B, ws, sta, step_keys[j][0], rank * B, Bg,
                              zT_all + 4 * (j * B * ldz + cfg.latent), ldz, want_logp=self.backup_entropy,
                              logp_out=nlogp_all + 4 * j * B)
                adone[j].record(astream)
            def enqueue_actor_obs():
                self._features([("A", actor.store, "obs")], feats, (utd_ratio - 1) * B, B, ws, sta, save_tag="A",
                               prep=False)
                nn.policy_fwd(cfg, "A", actor.store, zA, ldz, B, ws, sta, key_pi, rank * B, Bg, zP + 4 * cfg.latent,
                              ldz, save=True)
                ss["aobs"].record(astream)
            # the host interleaves the two streams' launches (pass j+1 is enqueued inside critic step j): enqueued all
            # up front, the main stream would sit empty while the host is busy with the actor stream's 500 launches
            enqueue_actor(0)

        for i in range(utd_ratio):
            lo = i * B
            ops.phase("step")
            # ---- critic step (sac_learner.py:160-219)
            key_act, key_sub = step_keys[i]
            m0, mstride, emin, idx = self._target_members(key_sub)
            tstore = target.store
            if m0 is None:
                tstore, m0, mstride = self._gathered_target(idx, target.store), 0, 1
            if True:
                # target chain on the main stream, critic chain on a side stream, next action from the actor stream
                zT = zT_all + 4 * i * B * ldz
                self._features([("T", target.store, "next"), ("C", critic.store, "obs")],
                               feats, lo, B, ws, st, save_tag="C", trunk_streams={"C": s1}, fork=ss["fork"],
                               z_ptrs={"T": zT})
                ops.phase("features_fwd")
                # start pass i+1 when the main stream leaves its (HBM-bound) encoder forward
                ss["agate"].record(main)
                astream.wait_event(ss["agate"])
                if i + 1 < utd_ratio:
                    enqueue_actor(i + 1)
                else:                    # behind the last pass: the actor's pass over obs for the actor step
                    enqueue_actor_obs()
                main.wait_event(ss["adone"][i])
                nlogp = nlogp_all + 4 * i * B if self.backup_entropy else None
            next_q = nn.ensemble_fwd(cfg, "T", tstore, m0, mstride, emin, zT, ldz, B, ws, st, save=False)
            st1 = s1.cuda_stream
            ops.copy2d(zC + 4 * cfg.latent, ldz, actions.data_ptr() + 4 * lo * A, A, B, A, st1)
            q = nn.ensemble_fwd(cfg, "C", critic.store, 0, 1, E, zC, ldz, B, ws, st1, save=True)
            ss["join"][1].record(s1)
            main.wait_event(ss["join"][1])
            dq = ws.ptr("dq", (E, B))
            ops.td_loss(next_q, emin, rewards.data_ptr() + 4 * lo, masks.data_ptr() + 4 * lo,
                        nlogp if self.backup_entropy else None, log_temp, self.discount, q, E, B, 1.0 / (E * Bg),
                        ws.ptr("target_q", (B,)), dq, info.data_ptr(), st)
            ops.phase("heads_fwd")
            gside = self._shared.get("grad_side")
            if gside is None:
                gside = self._shared["grad_side"] = nn.GradSide()
            gside.begin()
            # pixel learners: the trunk backward sums the split-K slabs of the input gradient on read; state-based:
            # observations are not trainable, the critic step needs no input gradient at all
            dslab = nn.ensemble_bwd(cfg, "C", critic.store, cgr, dq, zC, ldz, B, ws, st, gside=gside,
                                    dinp="slabs" if cfg.pixel else None)
            dinp, dz_slabs = (dslab[0], dslab[1:]) if dslab is not None else (None, (1, 0))
            ops.phase("ensemble_bwd")
            self._features_bwd("C", critic.store, cgr, feats, lo, B, ws, st, dinp, ldz, need_encoder=True, gside=gside,
                               dz_slabs=dz_slabs)
            gside.join()
            ops.phase("features_bwd")
            if dist is not None:      # data parallel: one blocking all-reduce(sum) of the flat gradient per critic step
                dist.all_reduce_sum(cgr.flat)
            critic.step += 1
            bc1, bc2 = bias_corrections(critic.step)
            ops.adam_polyak(critic.store.base, critic.mu.data_ptr(), critic.nu.data_ptr(), cgr.base,
                            target.store.base, critic.store.flat.numel(), critic.lr, ADAM_B1, ADAM_B2, ADAM_EPS, bc1, bc2,
                            self.tau, 1.0, st)

        # ---- actor step on the last minibatch (sac_learner.py:120-144, 235)
        ops.phase("actor")
        lo = (utd_ratio - 1) * B
        # the actor's own pass over obs (encoder, trunk, policy sample) ran ahead behind the last next_obs pass;
        # only the critic pass with the post-UTD critic parameters is left on the main stream
        main.wait_event(ss["aobs"])
        self._features([("C", critic.store, "obs")], feats, lo, B, ws, st, z_ptrs={"C": zP})
        logp = ws.ptr("logp_A", (B,))
        zC = zP                                            # critic input of the actor step
        q = nn.ensemble_fwd(cfg, "C", critic.store, 0, 1, E, zC, ldz, B, ws, st, save=True)
        ops.actor_loss(q, E, B, logp, log_temp, info.data_ptr() + 8, st)
        dq = ws.ptr("dq", (E, B))
        ops.fill(dq, -1.0 / (E * Bg), E * B, st)
        dinp = nn.ensemble_bwd(cfg, "C", critic.store, None, dq, zC, ldz, B, ws, st)
        A_ = cfg.action_dim
        dms = ws.ptr("dms", (B, 2 * A_))
        ms = ws.ptr("ms_A", (B, 2 * A_))
        ops.tanh_gaussian_bwd(zC + 4 * cfg.latent, ldz, ws.ptr("eps_A", (B, A_)), ms + 4 * A_, 2 * A_,
                              dinp + 4 * cfg.latent, ldz, log_temp, 1.0 / Bg, dms, dms + 4 * A_, 2 * A_, B, A_, st)
        dz = nn.policy_bwd(cfg, "A", actor.store, agr, zA, ldz, B, ws, st, need_dz=cfg.pixel)
        self._features_bwd("A", actor.store, agr, feats, lo, B, ws, st, dz, cfg.latent, need_encoder=False)
        if dist is not None:
            dist.all_reduce_sum(agr.flat)
            dist.all_reduce_sum(info[:4])                   # loss / q / logp sums -> global before the temperature step
        actor.step += 1
        bc1, bc2 = bias_corrections(actor.step)
        ops.adam_polyak(actor.store.base, actor.mu.data_ptr(), actor.nu.data_ptr(), agr.base, None,
                        actor.store.flat.numel(), actor.lr, ADAM_B1, ADAM_B2, ADAM_EPS, bc1, bc2, 0.0, 1.0, st)
        # ---- temperature (sac_learner.py:146-158, 236)
        if self._HAS_TEMP:
            temp.step += 1
            bc1, bc2 = bias_corrections(temp.step)
            ops.temperature_update(temp.buf.data_ptr(), info.data_ptr() + 12, 1.0 / Bg, self.target_entropy, temp.lr,
                                   ADAM_B1, ADAM_B2, ADAM_EPS, bc1, bc2, info.data_ptr() + 16, st)

        ops.phase("end")
        new_agent = self.replace(actor=actor, critic=critic, target_critic=target, temp=temp, rng=rng)
        snap = info.clone()         # `info` is workspace memory: the next update() call overwrites it
        return new_agent, UpdateInfo(snap, Bg, E, self._INFO_KEYS)
