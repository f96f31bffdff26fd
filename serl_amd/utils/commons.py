"""Mirrors serl/utils/commons.py (module path `serl.utils.commons`, imported by every reference training script:
binpick_franka_fwbw_vice.py:15, cable_route_..._multithread.py:13, pcb_insertion_ddpg.py:16, pixel_bc.py:16):
_unpack, _share_encoder, get_data, restore_checkpoint_, _reset_weights, ema -- on host/device array leaves and the
TrainState-like holders of serl_amd.agents.train_state."""
from ..frozen_dict import freeze
from .checkpoints import restore_checkpoint, restore_checkpoint_, save_checkpoint

__all__ = ["_unpack", "_share_encoder", "get_data", "restore_checkpoint_", "_reset_weights", "ema", "save_checkpoint",
           "restore_checkpoint"]


def _unpack(batch):
    """serl/utils/commons.py:7-29 for host/device array leaves (DrQLearner.update fuses this on device)."""
    obs = dict(batch["observations"])
    nxt = dict(batch["next_observations"])
    for k in list(obs.keys()):
        if k not in nxt:
            packed = obs[k]
            if hasattr(packed, "materialize"):
                packed = packed.materialize()
            obs[k] = packed[..., :-1]
            nxt[k] = packed[..., 1:]
    out = dict(batch)
    out["observations"], out["next_observations"] = obs, nxt
    return freeze(out)


def _share_encoder(source, target):
    """serl/utils/commons.py:32-47: copy every top-level '...encoder...' subtree from source into target."""
    new = target.clone()
    n = source.store.layout.prefix_size("image_encoder")
    new.store.flat[:n].copy_(source.store.flat[:n])
    return new



def get_data(data, start, end):
    """serl/utils/commons.py:49-61: slice every leaf of a (nested) dataset dict."""
    if type(data) == dict:                                   # noqa: E721  (the reference tests the exact type)
        return {k: get_data(v, start, end) for k, v in data.items()}
    return data[start:end]


def _reset_weights(source, target):
    """serl/utils/commons.py:76-90: copy every top-level subtree WITHOUT 'encoder' in its name from source to target."""
    replacers = {k: v for k, v in source.params.items() if "encoder" not in k}
    return target.replace(params=target.params.copy(add_or_replace=replacers))


def ema(series, alpha=0.5):
    """serl/utils/commons.py:92-104."""
    import numpy as np
    smoothed = np.zeros_like(series, dtype=float)
    smoothed[0] = series[0]
    for i in range(1, len(series)):
        smoothed[i] = alpha * series[i] + (1 - alpha) * smoothed[i - 1]
    return smoothed
