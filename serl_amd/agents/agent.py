"""Base agent: action sampling / mode (mirrors serl/agents/agent.py:12-37).

Agents are immutable: every method that changes state returns a new agent; device buffers of the
old one stay valid (the env thread keeps acting with an older copy while the learner thread
updates, serl_examples/*: `agent_queue`)."""
from __future__ import annotations

import copy

import numpy as np
import torch

from .. import ops, prng
from ..networks import modules as nn
from ..networks.modules import Workspace


class Agent:
    # set by subclasses' create(): actor (TrainState), rng (uint32[2]), _cfg (NetConfig), _shared (dict)

    def replace(self, **fields):
        new = copy.copy(self)
        for k, v in fields.items():
            if not hasattr(self, k):
                raise TypeError(f"{type(self).__name__}.replace: unknown field {k!r}")
            setattr(new, k, v)
        return new

    # ------------------------------------------------------------------ acting
    def _act(self, observations, key, deterministic, want_eps=False):
        cfg = self._cfg
        dev = self.actor.store.flat.device
        ws: Workspace = self._shared.setdefault("ws_act", Workspace(dev))
        st = ops.current_stream()
        A = cfg.action_dim
        ldz = cfg.latent + A
        z = ws.ptr("zA", (1, ldz))
        if cfg.pixel:
            imgs = []
            for cam in cfg.cams:
                x = np.ascontiguousarray(np.asarray(observations[cam], dtype=np.uint8)).reshape(
                    1, cfg.H, cfg.W, 3 * cfg.stack)                      # (H,W,3,S) -> channel c*S+s
                t = ws.tensor(f"img_{cam}", x.shape, torch.uint8)
                t.copy_(torch.from_numpy(x), non_blocking=False)
                imgs.append(t.data_ptr())
            s = ws.tensor("state", (1, cfg.state_dim))
            s.copy_(torch.from_numpy(np.asarray(observations["state"], dtype=np.float32).reshape(1, -1)))
            nn.encoder_fwd(cfg, [("A", self.actor.store, imgs)], 1, ws, st)
            nn.trunk_fwd(cfg, "A", self.actor.store, s.data_ptr(), z, ldz, 1, ws, st, save=False)
        else:
            zt = ws.tensor("zA", (1, ldz))
            zt[:, :cfg.obs_dim].copy_(torch.from_numpy(np.asarray(observations, dtype=np.float32).reshape(1, -1)))
        act = ws.tensor("act", (1, A))
        nn.policy_fwd(cfg, "A", self.actor.store, z, ldz, 1, ws, st, key, 0, 1, act.data_ptr(), A,
                      deterministic=deterministic, save=want_eps, want_logp=False)
        if want_eps:                 # the standard-normal draw behind the sample (DDPGLearner's exploration noise)
            return act.cpu().numpy()[0], ws.tensor("eps_A", (1, A)).cpu().numpy()[0]
        return act.cpu().numpy()[0]

    def eval_actions(self, observations) -> np.ndarray:
        """mode of the policy = tanh(mean) (agent.py:19-22,29-31)."""
        return self._act(observations, None, True)

    def sample_actions(self, observations):
        """key, rng = split(rng); a ~ pi(.|obs) (agent.py:12-16,33-37). Returns (action, new_agent)."""
        key, rng = prng.split(self.rng)
        a = self._act(observations, key, False)
        return a, self.replace(rng=rng)
