"""Host mirrors of serl/utils/commons.py helpers that operate on batches / TrainStates."""
from ..frozen_dict import freeze


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


__all__ = ["_unpack", "_share_encoder"]
