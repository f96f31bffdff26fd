"""Network descriptions (flat layouts with flax leaf names) and their HIP forward/backward passes."""
from .params import Layout, ParamStore
from .spec import NetConfig, actor_layout, critic_layout, init_tree


def subsample_ensemble(key, params, num_sample, num_qs):
    """Host mirror of serl/networks/ensemble.py:25-37 for parameter trees of array-likes."""
    from .. import prng
    if num_sample is None:
        return params
    indx = prng.choice_no_replace(key, num_qs, num_sample)

    def pick(t):
        if hasattr(t, "items"):
            return {k: pick(v) for k, v in t.items()}
        return t[indx]
    if "Ensemble_0" in params:
        out = dict(params)
        out["Ensemble_0"] = pick(params["Ensemble_0"])
        return out
    return pick(params)


__all__ = ["Layout", "ParamStore", "NetConfig", "actor_layout", "critic_layout", "init_tree", "subsample_ensemble"]
