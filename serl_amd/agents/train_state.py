"""TrainState-like holders over flat device buffers (flax.training.train_state.TrainState is what
the reference stores in `agent.actor / critic / target_critic / temp`, serl/agents/sac/sac_learner.py:69-104).
Immutable by convention: `replace` returns a new object; the learner clones buffers before stepping."""
from __future__ import annotations

import numpy as np

from ..frozen_dict import FrozenDict
from ..networks.params import ParamStore

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # optax.adam defaults


def bias_corrections(step: int):
    """1 - b^t evaluated in float32 like optax (`1 - decay**count`)."""
    f = np.float32
    return float(f(1) - f(ADAM_B1) ** f(step)), float(f(1) - f(ADAM_B2) ** f(step))


class TrainState:
    def __init__(self, store: ParamStore, mu=None, nu=None, step: int = 0, lr=None, apply_fn=None):
        self.store, self.mu, self.nu = store, mu, nu
        self.step, self.lr, self.apply_fn = int(step), lr, apply_fn

    @classmethod
    def create(cls, store: ParamStore, lr=None, apply_fn=None):
        import torch
        if lr is None:      # target network: no optimiser (sac_learner.py:92-96)
            return cls(store, None, None, 0, None, apply_fn)
        return cls(store, torch.zeros_like(store.flat), torch.zeros_like(store.flat), 0, lr, apply_fn)

    # ---- flax-like views
    @property
    def params(self) -> FrozenDict:
        return self.store.tree()

    @property
    def opt_state(self):
        if self.mu is None:
            return None
        lay = self.store.layout
        return FrozenDict(count=self.step, mu=lay.unflatten(self.mu), nu=lay.unflatten(self.nu))

    def clone(self) -> "TrainState":
        return TrainState(self.store.clone(), None if self.mu is None else self.mu.clone(),
                          None if self.nu is None else self.nu.clone(), self.step, self.lr, self.apply_fn)

    def replace(self, **kw) -> "TrainState":
        new = TrainState(self.store, self.mu, self.nu, self.step, self.lr, self.apply_fn)
        for k, v in kw.items():
            if k == "params":
                new.store = v if isinstance(v, ParamStore) else ParamStore.from_tree(
                    self.store.layout, v, self.store.flat.device)
            elif k == "step":
                new.step = int(v)
            elif k == "opt_state":
                import torch
                lay, dev = self.store.layout, self.store.flat.device
                new.mu = torch.from_numpy(lay.flatten(v["mu"])).to(dev)
                new.nu = torch.from_numpy(lay.flatten(v["nu"])).to(dev)
                new.step = int(v["count"])
            elif k in ("apply_fn", "lr"):
                setattr(new, k, v)
            else:
                raise TypeError(f"TrainState.replace: unknown field {k!r}")
        return new


class TempState:
    """Temperature 'network': one scalar log_temp + its Adam moments in a 3-float device buffer
    (serl/agents/sac/temperature.py:5-14)."""

    def __init__(self, buf, step: int, lr: float):
        self.buf, self.step, self.lr = buf, int(step), lr

    @classmethod
    def create(cls, init_temperature: float, lr: float, device):
        import torch
        buf = torch.tensor([np.log(np.float32(init_temperature)), 0.0, 0.0], dtype=torch.float32, device=device)
        return cls(buf, 0, lr)

    @property
    def params(self) -> FrozenDict:
        return FrozenDict(log_temp=self.buf[0])

    @property
    def opt_state(self):
        return FrozenDict(count=self.step, mu=FrozenDict(log_temp=self.buf[1]), nu=FrozenDict(log_temp=self.buf[2]))

    def apply_fn(self, variables=None):
        return float(np.exp(self.buf[0].item()))

    def clone(self):
        return TempState(self.buf.clone(), self.step, self.lr)

    def replace(self, **kw):
        new = TempState(self.buf, self.step, self.lr)
        for k, v in kw.items():
            if k == "params":
                buf = self.buf.clone()
                lt = v["log_temp"]
                buf[0] = float(lt.item() if hasattr(lt, "item") else lt)
                new.buf = buf
            elif k == "step":
                new.step = int(v)
            else:
                raise TypeError(f"TempState.replace: unknown field {k!r}")
        return new
