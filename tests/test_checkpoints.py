"""Checkpoint wire format (serl_amd/utils/checkpoints.py) on CPU tensors: the byte layout of
flax.serialization as published (hand-assembled known-answer bytes; flax itself is absent -> parity unpinned),
TrainState / TempState round trips and flax's file-management rules."""
import os

import numpy as np
import pytest
import torch

from serl_amd.agents.train_state import TempState, TrainState
from serl_amd.networks.params import Layout, ParamStore
from serl_amd.utils import _reset_weights, checkpoints as ck, ema, get_data


def _state(seed, with_opt=True):
    lay = Layout().add(("image_encoder", "Conv_0", "kernel"), (3, 3, 2, 4)).add(("image_encoder", "Conv_0", "bias"), (4,))
    lay.add(("Dense_0", "kernel"), (5, 3)).add(("Dense_0", "bias"), (3,))
    g = torch.Generator().manual_seed(seed)
    ts = TrainState.create(ParamStore(lay, torch.randn(lay.size, generator=g)), lr=3e-4 if with_opt else None)
    if with_opt:
        ts.mu.copy_(torch.randn(lay.size, generator=g)); ts.nu.copy_(torch.rand(lay.size, generator=g)); ts.step = 17
    return ts


def _same_leaves(a: TrainState, b: TrainState):
    ta, tb = a.store.numpy_tree(), b.store.numpy_tree()
    for path in a.store.layout.entries:
        x, y = ta, tb
        for k in path:
            x, y = x[k], y[k]
        if not np.array_equal(x, y):
            return False
    return True


def _same_flat(lay, a, b):
    """Equality of two flat buffers on the leaves of the layout (alignment gaps between leaves are not state)."""
    return all(torch.equal(a[off:off + int(np.prod(shape))], b[off:off + int(np.prod(shape))])
               for off, shape in lay.entries.values())


def test_known_answer_bytes():
    a = np.array([1.0, 2.0], dtype=np.float32)
    want = (b"\x81\xa1a" + b"\xc7\x15\x01" + b"\x93\x91\x02" + b"\xa7float32" + b"\xc4\x08" + a.tobytes())
    assert ck.msgpack_serialize({"a": a}) == want
    back = ck.msgpack_restore(want)
    assert back["a"].dtype == np.float32 and np.array_equal(back["a"], a)
    # scalars: python ints stay msgpack ints, numpy scalars are ExtType 3 with shape ()
    blob = ck.msgpack_serialize({"step": 5, "count": np.int32(7), "none": None, "t": (np.zeros((1, 2), np.uint8),)})
    back = ck.msgpack_restore(blob)
    assert back["step"] == 5 and back["count"] == 7 and back["count"].dtype == np.int32 and back["none"] is None
    assert list(back["t"].keys()) == ["0"] and back["t"]["0"].shape == (1, 2)
    assert b"\xc7\x0e\x03\x93\x90\xa5int32\xc4\x04\x07\x00\x00\x00" in blob


def test_state_dict_layout_and_roundtrip(tmp_path):
    ts = _state(0)
    sd = ck.to_state_dict(ts)
    assert set(sd) == {"step", "params", "opt_state"} and sd["step"] == 17
    assert set(sd["opt_state"]) == {"0", "1"} and sd["opt_state"]["1"] == {}
    assert set(sd["opt_state"]["0"]) == {"count", "mu", "nu"} and int(sd["opt_state"]["0"]["count"]) == 17
    assert sd["params"]["image_encoder"]["Conv_0"]["kernel"].shape == (3, 3, 2, 4)
    p = ck.save_checkpoint(tmp_path, ts, step=20000, prefix="actor_", keep=1000)
    assert os.path.basename(p) == "actor_20000"
    other = _state(1)
    for restored in (ck.restore_checkpoint(tmp_path, other, prefix="actor_"),          # newest in directory
                     ck.restore_checkpoint(tmp_path, other, step=20000, prefix="actor_"),
                     ck.restore_checkpoint_(p, other, None)):                            # the file itself (commons.py:63-74)
        assert _same_leaves(restored, ts) and restored.step == 17
        lay = ts.store.layout
        assert _same_flat(lay, restored.mu, ts.mu) and _same_flat(lay, restored.nu, ts.nu)
        assert not _same_leaves(other, ts)                                               # target untouched
    # target network: no optimizer -> nil opt_state
    tgt = _state(2, with_opt=False)
    assert ck.to_state_dict(tgt)["opt_state"] is None
    ck.save_checkpoint(tmp_path, tgt, step=1, prefix="target_critic_")
    back = ck.restore_checkpoint(tmp_path, _state(3, with_opt=False), prefix="target_critic_")
    assert _same_leaves(back, tgt) and back.mu is None
    # temperature
    temp = TempState(torch.tensor([-2.3, 0.01, 0.002]), 9, 3e-4)
    ck.save_checkpoint(tmp_path, temp, step=1, prefix="temp_")
    t2 = ck.restore_checkpoint(tmp_path, TempState.create(1.0, 3e-4, "cpu"), prefix="temp_")
    assert torch.equal(t2.buf, temp.buf) and t2.step == 9
    # raw state dict when no target is given
    raw = ck.restore_checkpoint(tmp_path, None, prefix="actor_")
    assert np.array_equal(raw["params"]["Dense_0"]["bias"], ts.store.numpy_tree()["Dense_0"]["bias"])


def test_file_management_and_errors(tmp_path):
    ts = _state(0)
    for s in (1, 2, 3, 10):
        ck.save_checkpoint(tmp_path, ts, step=s, keep=2)
    assert sorted(os.listdir(tmp_path)) == ["checkpoint_10", "checkpoint_3"]            # natural order, newest 2 kept
    assert ck.latest_checkpoint(tmp_path).endswith("checkpoint_10")
    with pytest.raises(ValueError):
        ck.save_checkpoint(tmp_path, ts, step=5, keep=2)                                 # older than the newest
    ck.save_checkpoint(tmp_path, ts, step=5, keep=2, overwrite=True)
    assert sorted(os.listdir(tmp_path)) == ["checkpoint_3", "checkpoint_5"]
    empty = tmp_path / "none"
    empty.mkdir()
    assert ck.restore_checkpoint(empty, ts) is ts                                        # nothing found: target returned
    with pytest.raises(ValueError):
        ck.restore_checkpoint(tmp_path, ts, step=4)
    with pytest.raises(AssertionError):
        ck.restore_checkpoint_(tmp_path / "missing", ts, None)
    # a checkpoint of another architecture does not restore silently
    lay = Layout().add(("Dense_0", "kernel"), (5, 3))
    wrong = TrainState.create(ParamStore(lay, torch.zeros(lay.size)), lr=1e-3)
    with pytest.raises((ValueError, KeyError)):
        ck.restore_checkpoint(tmp_path, wrong)


def test_commons_helpers():
    src, dst = _state(0), _state(1)
    out = _reset_weights(src, dst)                     # commons.py:76-90: everything but '...encoder...' subtrees
    a, b, o = src.store.numpy_tree(), dst.store.numpy_tree(), out.store.numpy_tree()
    assert np.array_equal(o["Dense_0"]["kernel"], a["Dense_0"]["kernel"])
    assert np.array_equal(o["image_encoder"]["Conv_0"]["kernel"], b["image_encoder"]["Conv_0"]["kernel"])
    d = {"x": np.arange(10), "y": {"z": np.arange(20).reshape(10, 2)}}
    s = get_data(d, 2, 5)
    assert np.array_equal(s["x"], [2, 3, 4]) and s["y"]["z"].shape == (3, 2)
    assert np.allclose(ema([1.0, 3.0, 5.0], 0.5), [1.0, 2.0, 3.5])
