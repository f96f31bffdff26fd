"""Run the reference's training scripts on serl_amd without editing them: `serl_amd.compat.install()` registers the
serl_amd packages under the reference's module paths (`serl.agents`, `serl.data`, `serl.networks`,
`serl.distributions`, `serl.utils`, `serl.utils.commons`, `serl.utils.augmentations`) in `sys.modules`.

The scripts' own import lines this serves (serl_examples/*): `from serl.agents import SACLearner, PixelDDPGLearner,
...`, `from serl.data import ReplayBuffer, MemoryEfficientReplayBuffer`, `from serl.utils.commons import get_data,
restore_checkpoint_, ema`.  `serl.evaluation` and `serl.wrappers` (env side, out of scope) are left to the real `serl`
package when it is installed; when it is not, `serl` becomes an empty parent package so that the learner-side imports
above still resolve."""
from __future__ import annotations

import importlib
import sys
import types

_SUBMODULES = ("agents", "data", "networks", "distributions", "utils", "utils.commons", "utils.augmentations")


def install(name: str = "serl", force_stub: bool = False):
    """Alias serl_amd's packages under `name`.  Returns the list of module paths registered."""
    parent = None
    if not force_stub:
        try:
            parent = importlib.import_module(name)          # the real serl package (its __init__ imports nothing)
        except Exception:                                    # noqa: BLE001  absent, or broken by a missing dependency
            parent = None
    if parent is None:
        parent = types.ModuleType(name)
        parent.__path__ = []                                 # a package: `import serl.x` consults sys.modules first
        parent.__doc__ = "alias parent created by serl_amd.compat.install()"
    sys.modules[name] = parent
    done = []
    for sub in _SUBMODULES:
        mod = importlib.import_module(f"serl_amd.{sub}")
        sys.modules[f"{name}.{sub}"] = mod
        holder = parent if "." not in sub else sys.modules[f"{name}.{sub.rsplit('.', 1)[0]}"]
        if holder is parent:
            setattr(parent, sub, mod)
        done.append(f"{name}.{sub}")
    return done
