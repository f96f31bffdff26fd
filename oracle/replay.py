"""ORACLE (test infrastructure only) -- numpy restatement of SERL's replay path.

Follows serl/data/dataset.py:81-104 (index draw + fancy-index gather),
serl/data/replay_buffer.py:17-79 (ring storage) and
serl/data/memory_efficient_replay_buffer.py:13-169 (de-duplicated frame storage, validity
mask, rejection resampling, demo tail overwrite, sliding-window frame-stack gather).
Written as plain functions over a dict so it shares no structure with the product classes
in serl_amd/data.  PARITY: the index stream is numpy's own Generator (the reference calls
the same numpy API, gym.utils.seeding.np_random(seed) == Generator(PCG64(SeedSequence(seed)))),
pinned by the numpy KAT in tests/test_oracle_replay.py.
"""
from __future__ import annotations

import numpy as np


def make_rng(seed):
    """gym 0.26 seeding.np_random(seed)."""
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


def _alloc(spec, capacity):
    if isinstance(spec, dict):
        return {k: _alloc(v, capacity) for k, v in spec.items()}
    shape, dtype = spec
    return np.zeros((capacity, *shape), dtype=dtype)


def _write(store, item, at):
    if isinstance(store, dict):
        assert store.keys() == item.keys()
        for k in store:
            _write(store[k], item[k], at)
    else:
        store[at] = item


def _take(store, indx):
    if isinstance(store, dict):
        return {k: _take(v, indx) for k, v in store.items()}
    return store[indx]


def new_ring(obs_spec, act_spec, capacity, next_obs_spec=None):
    """replay_buffer.py:45-70."""
    next_obs_spec = obs_spec if next_obs_spec is None else next_obs_spec
    data = dict(
        observations=_alloc(obs_spec, capacity),
        next_observations=_alloc(next_obs_spec, capacity),
        actions=_alloc(act_spec, capacity),
        rewards=np.zeros((capacity,), np.float32),
        masks=np.zeros((capacity,), np.float32),
        dones=np.zeros((capacity,), bool),
    )
    return dict(data=data, size=0, capacity=capacity, head=0)


def ring_insert(buf, transition):
    """replay_buffer.py:75-79."""
    _write(buf["data"], transition, buf["head"])
    buf["head"] = (buf["head"] + 1) % buf["capacity"]
    buf["size"] = min(buf["size"] + 1, buf["capacity"])


def ring_sample(buf, rng, batch_size, indx=None):
    """dataset.py:81-104 (sampling over len(self) == current size)."""
    if indx is None:
        indx = rng.integers(buf["size"], size=batch_size)
    return _take(buf["data"], indx), indx


def ring_upload(buf, block):
    """replay_buffer.py:141-147 (GPUReplayBuffer.upload): a block of rows at the insert index."""
    n = len(block["rewards"])
    lo = buf["head"]
    assert lo + n <= buf["capacity"]

    def put(store, item):
        if isinstance(store, dict):
            for k in item:
                put(store[k], item[k])
        else:
            store[lo:lo + n] = item
    put(buf["data"], block)
    buf["head"] = (lo + n) % buf["capacity"]
    buf["size"] = min(buf["size"] + n, buf["capacity"])


def ring_sample_jax(buf, key, batch_size):
    """dataset.py:116-129 (sample_jax): `key, rng = split(rng)`; indx = randint(key, (B,), 0, len);
    returns (new rng, indx.max(), gathered batch, indx).  Index stream = oracle.prng's restatement of
    jax.random.randint (parity unpinned beyond the threefry KATs)."""
    from . import prng
    sub, rng = prng.split(key)
    indx = prng.randint(sub, (batch_size,), 0, buf["size"])
    return rng, int(indx.max()), _take(buf["data"], indx), indx


# ------------------------------------------------------------------ memory-efficient buffer
def new_frame_buffer(obs_spec, act_spec, capacity, pixel_keys):
    """memory_efficient_replay_buffer.py:14-52: pixel leaves lose their trailing stack axis,
    next_observations has no pixel leaves."""
    num_stack = None
    unstacked = dict(obs_spec)
    for k in pixel_keys:
        shape, dtype = obs_spec[k]
        num_stack = shape[-1] if num_stack is None else num_stack
        assert num_stack == shape[-1]
        unstacked[k] = (shape[:-1], dtype)
    next_spec = {k: v for k, v in unstacked.items() if k not in pixel_keys}
    buf = new_ring(unstacked, act_spec, capacity, next_spec)
    buf.update(pixel_keys=tuple(pixel_keys), num_stack=num_stack, first=True,
               valid=np.zeros(capacity, bool))
    return buf


def frame_insert(buf, tr):
    """memory_efficient_replay_buffer.py:54-90."""
    S = buf["num_stack"]
    if buf["head"] == 0 and buf["capacity"] == buf["size"] and not buf["first"]:
        for idx in range(buf["size"] - S, buf["size"]):           # :55-60 wrap-around re-copy
            element = _take(buf["data"], idx)
            buf["valid"][buf["head"]] = False
            ring_insert(buf, element)
    row = {k: (dict(v) if isinstance(v, dict) else v) for k, v in tr.items()}
    obs_px = {k: row["observations"].pop(k) for k in buf["pixel_keys"]}
    nxt_px = {k: row["next_observations"].pop(k) for k in buf["pixel_keys"]}
    if buf["first"]:
        for i in range(S):                                        # :72-78
            for k in buf["pixel_keys"]:
                row["observations"][k] = obs_px[k][..., i]
            buf["valid"][buf["head"]] = False
            ring_insert(buf, row)
    for k in buf["pixel_keys"]:
        row["observations"][k] = nxt_px[k][..., -1]               # :80-81
    buf["first"] = bool(row["dones"])
    buf["valid"][buf["head"]] = True
    ring_insert(buf, row)
    for i in range(S):                                            # :88-90
        buf["valid"][(buf["head"] + i) % buf["size"]] = False


def frame_sample_indices(buf, rng, batch_size, demo_batch_size=0, demo_size=0):
    """memory_efficient_replay_buffer.py:114-131: vector draw, in-order scalar redraw of
    invalid slots, then the demo tail overwrite (no validity check)."""
    n = buf["size"]
    indx = rng.integers(n, size=batch_size)
    for i in range(batch_size):
        while not buf["valid"][indx[i]]:
            indx[i] = rng.integers(n)
    if demo_size and demo_batch_size:
        demo = rng.integers(demo_size, size=demo_batch_size)
        indx[-len(demo):] = demo
    return indx


def frame_gather(buf, indx, pack_obs_and_next_obs=False):
    """memory_efficient_replay_buffer.py:133-169."""
    S = buf["num_stack"]
    data = buf["data"]
    batch = {k: _take(data[k], indx) for k in data if k != "observations"}
    batch["observations"] = {k: data["observations"][k][indx]
                             for k in data["observations"] if k not in buf["pixel_keys"]}
    for k in buf["pixel_keys"]:
        frames = data["observations"][k]
        win = np.lib.stride_tricks.sliding_window_view(frames, S + 1, axis=0)
        win = win[indx - S]                                       # negative -> wraps (quirk C.7)
        if pack_obs_and_next_obs:
            batch["observations"][k] = win
        else:
            batch["observations"][k] = win[..., :-1]
            batch["next_observations"][k] = win[..., 1:]
    return batch
