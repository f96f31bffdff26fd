"""Checkpoint wire format of the reference's scripts.

The reference saves every TrainState of an agent with `flax.training.checkpoints.save_checkpoint(dir, agent.actor,
prefix='actor_', step=i, keep=1000)` and restores with `restore_checkpoint(path, agent.actor, step)` followed by
`agent.replace(actor=...)` (serl_examples/drq_rlpd_with_classifier_reward/cable_route_...py:141-146,164-170,184-207;
serl/utils/commons.py:63-74).  This module restates that format for the TrainState-like holders of this package:

* file `<ckpt_dir>/<prefix><step>` holding `flax.serialization.to_bytes(target)`: msgpack of the state dict
  `{'step', 'params', 'opt_state'}` (apply_fn / tx are static fields and are not serialised); nested dicts keep flax's
  parameter names, tuples become dicts keyed '0', '1', ...; optax.adam's state is `{'0': {'count','mu','nu'}, '1': {}}`
  and the target network's no-op optimiser state is nil;
* an array is msgpack ExtType 1 whose payload is `packb((shape, dtype.name, raw bytes))`, a numpy scalar ExtType 3
  (same payload), a complex number ExtType 2 (flax/serialization.py `_msgpack_ext_pack` / `_ndarray_to_bytes`).

flax is not installed here and the reference ships no checkpoint file, so this follows the published format of
flax.serialization (legacy single-file checkpoints; orbax directories are not handled): **parity unpinned**.
"""
from __future__ import annotations

import os
import re

import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3


def _msgpack():
    import msgpack
    return msgpack


def _ndarray_to_bytes(a) -> bytes:
    a = np.asarray(a)
    if a.dtype.hasobject or a.dtype.isalignedstruct:
        raise ValueError("object and structured dtypes are not serialisable")
    return _msgpack().packb((a.shape, a.dtype.name, a.tobytes()), use_bin_type=True)


def _ndarray_from_bytes(data: bytes) -> np.ndarray:
    shape, dtype_name, buf = _msgpack().unpackb(data, raw=True)
    name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    return np.frombuffer(buf, dtype=np.dtype(name)).reshape(tuple(shape)).copy()


def _ext_pack(x):
    msgpack = _msgpack()
    if hasattr(x, "detach"):                       # device / torch leaf
        x = x.detach().cpu().numpy()
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_to_bytes(x))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
    return x


def _ext_unpack(code, data):
    if code == _EXT_NDARRAY:
        return _ndarray_from_bytes(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_bytes(data)[()]
    if code == _EXT_COMPLEX:
        re_, im = _msgpack().unpackb(data)
        return complex(re_, im)
    return _msgpack().ExtType(code, data)


def msgpack_serialize(state_dict) -> bytes:
    """flax.serialization.msgpack_serialize (arrays below the 2**30-byte chunking threshold)."""
    return _msgpack().packb(_plain(state_dict), default=_ext_pack, strict_types=True)


def msgpack_restore(data: bytes):
    """flax.serialization.msgpack_restore."""
    return _msgpack().unpackb(data, ext_hook=_ext_unpack, raw=False, strict_map_key=False)


def _plain(tree):
    """FrozenDict / mappings -> dict, sequences -> {'0': ..}, leaves untouched (flax to_state_dict rules)."""
    if hasattr(tree, "items"):
        return {str(k): _plain(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return {str(i): _plain(v) for i, v in enumerate(tree)}
    return tree


# ------------------------------------------------------------------------------------------------ state dicts
def to_state_dict(target) -> dict:
    """State dict of a TrainState / TempState as flax's struct.dataclass serialisation produces it."""
    from ..agents.train_state import TempState, TrainState
    if isinstance(target, TrainState):
        params = target.store.numpy_tree()
        if target.mu is None:
            opt = None
        else:
            lay = target.store.layout
            opt = {"0": {"count": np.asarray(target.step, dtype=np.int32),
                         "mu": lay.unflatten(target.mu.detach().cpu().numpy(), wrap=lambda a: a.copy()),
                         "nu": lay.unflatten(target.nu.detach().cpu().numpy(), wrap=lambda a: a.copy())},
                   "1": {}}
        return {"step": int(target.step), "params": params, "opt_state": opt}
    if isinstance(target, TempState):
        h = target.buf.detach().cpu().numpy()
        return {"step": int(target.step), "params": {"log_temp": h[0].copy()},
                "opt_state": {"0": {"count": np.asarray(target.step, dtype=np.int32),
                                    "mu": {"log_temp": h[1].copy()}, "nu": {"log_temp": h[2].copy()}}, "1": {}}}
    return _plain(target)


def from_state_dict(target, state):
    """Inverse of to_state_dict: a new TrainState / TempState shaped like `target` holding `state`'s values.
    Names and shapes must match the target's (ValueError otherwise, like flax)."""
    from ..agents.train_state import TempState, TrainState
    if isinstance(target, TrainState):
        new = target.replace(params=state["params"])
        opt = state.get("opt_state")
        if target.mu is not None:
            if opt is None:
                raise ValueError("checkpoint has no optimizer state but the target TrainState has one")
            adam = opt["0"]
            new = new.replace(opt_state={"count": int(np.asarray(adam["count"])), "mu": adam["mu"], "nu": adam["nu"]})
        new.step = int(np.asarray(state["step"]))
        return new
    if isinstance(target, TempState):
        import torch
        adam = state["opt_state"]["0"]
        vals = [state["params"]["log_temp"], adam["mu"]["log_temp"], adam["nu"]["log_temp"]]
        buf = torch.tensor([float(np.asarray(v)) for v in vals], dtype=torch.float32, device=target.buf.device)
        return TempState(buf, int(np.asarray(state["step"])), target.lr)
    return state


def to_bytes(target) -> bytes:
    return msgpack_serialize(to_state_dict(target))


def from_bytes(target, data: bytes):
    state = msgpack_restore(data)
    return state if target is None else from_state_dict(target, state)


# ------------------------------------------------------------------------------------------------ files
def _natural_key(name: str):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def _checkpoint_files(ckpt_dir: str, prefix: str):
    if not os.path.isdir(ckpt_dir):
        return []
    pat = re.compile(re.escape(prefix) + r"(\d+(?:\.\d+)?)$")
    names = [n for n in os.listdir(ckpt_dir) if pat.match(n) and not n.endswith("tmp")]
    return [os.path.join(ckpt_dir, n) for n in sorted(names, key=_natural_key)]


def latest_checkpoint(ckpt_dir: str, prefix: str = "checkpoint_"):
    files = _checkpoint_files(ckpt_dir, prefix)
    return files[-1] if files else None


def save_checkpoint(ckpt_dir, target, step, prefix: str = "checkpoint_", keep: int = 1, overwrite: bool = False) -> str:
    """flax.training.checkpoints.save_checkpoint: write `<ckpt_dir>/<prefix><step>` atomically, refuse to go back in
    step unless `overwrite`, keep the newest `keep` files of this prefix."""
    ckpt_dir = os.fspath(ckpt_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    existing = _checkpoint_files(ckpt_dir, prefix)
    if existing and not overwrite:
        newest = _natural_key(os.path.basename(existing[-1]))
        if _natural_key(os.path.basename(path)) <= newest:
            raise ValueError(f"trying to save an outdated checkpoint at step {step!r}: {existing[-1]} exists "
                             "(pass overwrite=True)")
    tmp = os.path.join(ckpt_dir, f"{prefix}tmp")
    with open(tmp, "wb") as f:
        f.write(to_bytes(target))
    os.replace(tmp, path)
    files = _checkpoint_files(ckpt_dir, prefix)
    if overwrite:                                  # drop checkpoints newer than the one just written
        for p in files[files.index(path) + 1:]:
            os.remove(p)
        files = files[:files.index(path) + 1]
    for p in files[:-keep] if keep > 0 else []:
        os.remove(p)
    return path


def restore_checkpoint(ckpt_dir, target, step=None, prefix: str = "checkpoint_"):
    """flax.training.checkpoints.restore_checkpoint: `ckpt_dir` may be the checkpoint file itself; with a directory,
    `step` selects `<prefix><step>` and None the newest.  No checkpoint found -> `target` is returned unchanged."""
    ckpt_dir = os.fspath(ckpt_dir)
    if step is not None:
        path = os.path.join(ckpt_dir, f"{prefix}{step}")
        if not os.path.exists(path):
            raise ValueError(f"matching checkpoint not found: {path}")
    elif os.path.isfile(ckpt_dir):
        path = ckpt_dir
    else:
        path = latest_checkpoint(ckpt_dir, prefix)
        if path is None:
            return target
    with open(path, "rb") as f:
        return from_bytes(target, f.read())


def restore_checkpoint_(path, item, step):
    """serl/utils/commons.py:63-74."""
    assert os.path.exists(path)
    return restore_checkpoint(path, item, step)
