"""Base agent: action sampling / mode (mirrors serl/agents/agent.py:12-37).

Agents are immutable: every method that changes state returns a new agent; device buffers of the
old one stay valid (the env thread keeps acting with an older copy while the learner thread
updates, serl_examples/*: `agent_queue`)."""
from __future__ import annotations

import copy

import numpy as np
import torch

from .. import ops, prng


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
        """One policy evaluation on a single, un-batched observation: one call into the native driver
        (serl_learner_act: encoder, trunk, TanhNormal head), one small D2H copy of the action."""
        cfg = self._cfg
        dev = self.actor.store.flat.device
        A = cfg.action_dim
        imgs, keep = None, []
        with torch.cuda.device(dev):
            if cfg.pixel:
                imgs = []
                for cam in cfg.cams:
                    x = np.ascontiguousarray(np.asarray(observations[cam], dtype=np.uint8)).reshape(
                        cfg.H, cfg.W, 3 * cfg.stack)                         # (H,W,3,S) -> channel c*S+s
                    t = torch.from_numpy(x).to(dev)
                    keep.append(t)
                    imgs.append(t.data_ptr())
                s = torch.from_numpy(np.asarray(observations["state"], dtype=np.float32).reshape(-1)).to(dev)
            else:
                s = torch.from_numpy(np.asarray(observations, dtype=np.float32).reshape(-1)).to(dev)
            out = torch.empty(2 * A, dtype=torch.float32, device=dev)          # [action | eps]
            self._shared["native"].act(self.actor.store.base, imgs, s.data_ptr(), key, deterministic, out.data_ptr(),
                                       out.data_ptr() + 4 * A if want_eps else None, ops.current_stream())
            host = out.cpu().numpy()
        if want_eps:                 # the standard-normal draw behind the sample (DDPGLearner's exploration noise)
            return host[:A], host[A:]
        return host[:A]

    def eval_actions(self, observations) -> np.ndarray:
        """mode of the policy = tanh(mean) (agent.py:19-22,29-31)."""
        return self._act(observations, None, True)

    def sample_actions(self, observations):
        """key, rng = split(rng); a ~ pi(.|obs) (agent.py:12-16,33-37). Returns (action, new_agent)."""
        key, rng = prng.split(self.rng)
        a = self._act(observations, key, False)
        return a, self.replace(rng=rng)
