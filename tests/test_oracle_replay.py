"""CPU tests of the replay path: the oracle (oracle/replay.py) against the numpy KAT and the storage
invariants, and the product's host buffers (serl_amd.data, device=None) against the oracle: identical
validity masks, identical sampled indices (bit-exact) and identical gathered batches."""
import json
import os

import numpy as np
import pytest

from helpers import pixel_spaces, rollout_transitions, state_spaces
from oracle import replay as orp

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prng_kat.json")))
KEYS = ("wrist_1", "wrist_2")


def test_numpy_generator_kat():
    """gym.utils.seeding.np_random(42) == Generator(PCG64(SeedSequence(42))): pins the index stream."""
    g = orp.make_rng(42)
    kat = GOLD["numpy_pcg64_seed42_integers1000"]
    assert g.integers(1000, size=8).tolist() == kat["vector"]
    assert int(g.integers(1000)) == kat["next_scalar"]
    from serl_amd.data import Dataset
    d = Dataset(dict(x=np.arange(1000)), seed=42)
    assert d.np_random.integers(1000, size=8).tolist() == kat["vector"]


def _fill(H, W, S, cap, n, episode_len, seed=0):
    from serl_amd.data import MemoryEfficientReplayBuffer
    obs_space, act_space = pixel_spaces(KEYS, H, W, S, 5, 3)
    buf = MemoryEfficientReplayBuffer(obs_space, act_space, cap, pixel_keys=KEYS)
    spec = {k: ((H, W, 3, S), np.uint8) for k in KEYS}
    spec["state"] = ((5,), np.float32)
    obuf = orp.new_frame_buffer(spec, ((3,), np.float32), cap, KEYS)
    trs = list(rollout_transitions(np.random.default_rng(seed), n, KEYS, H, W, S, 5, 3, episode_len=episode_len))
    for tr in trs:
        buf.insert(tr)
        orp.frame_insert(obuf, tr)
    return buf, obuf, trs


@pytest.mark.parametrize("S,cap,n,ep", [(3, 64, 40, 7), (3, 64, 200, 11), (1, 50, 130, 9), (2, 37, 37, 100)])
def test_frame_buffer_matches_oracle(S, cap, n, ep):
    buf, obuf, trs = _fill(8, 16, S, cap, n, ep)
    assert len(buf) == obuf["size"] and buf._insert_index == obuf["head"]
    assert np.array_equal(buf._is_correct_index, obuf["valid"])
    for k in KEYS:
        assert np.array_equal(buf.dataset_dict["observations"][k][:len(buf)], obuf["data"]["observations"][k][:len(buf)])
    # invariant: every valid row has S in-episode predecessors = the transition's stacked obs + next frame
    valid = np.flatnonzero(obuf["valid"][:obuf["size"]])
    assert len(valid) > 0
    fr = obuf["data"]["observations"][KEYS[0]]
    for i in valid:
        assert i >= S
        win = np.stack([fr[i - S + t] for t in range(S + 1)], -1)
        # find the transition whose newest next-frame sits in row i
        match = [tr for tr in trs if np.array_equal(tr["next_observations"][KEYS[0]][..., -1], fr[i])]
        assert any(np.array_equal(tr["observations"][KEYS[0]], win[..., :-1]) and
                   np.array_equal(tr["next_observations"][KEYS[0]], win[..., 1:]) for tr in match)
    # sampling: same generator stream -> same indices, same batches (packed and unpacked, with demo tail)
    for args in (dict(batch_size=32), dict(batch_size=20, pack_obs_and_next_obs=True),
                 dict(batch_size=24, pack_obs_and_next_obs=True, demo_batch_size=12, demo_size=min(10, len(buf)))):
        buf.seed(5)
        org = orp.make_rng(5)
        for _ in range(3):
            b = buf.sample(**args)
            idx = orp.frame_sample_indices(obuf, org, args["batch_size"], args.get("demo_batch_size", 0),
                                           args.get("demo_size", 0))
            ob = orp.frame_gather(obuf, idx, args.get("pack_obs_and_next_obs", False))
            assert np.array_equal(b["actions"], ob["actions"]) and np.array_equal(b["rewards"], ob["rewards"])
            assert np.array_equal(b["observations"]["state"], ob["observations"]["state"])
            assert np.array_equal(b["next_observations"]["state"], ob["next_observations"]["state"])
            for k in KEYS:
                assert np.array_equal(b["observations"][k], ob["observations"][k])
                if not args.get("pack_obs_and_next_obs", False):
                    assert np.array_equal(b["next_observations"][k], ob["next_observations"][k])
                else:
                    assert k not in b["next_observations"]
            if not args.get("demo_size"):
                assert obuf["valid"][idx].all()


def test_ring_buffer_and_errors():
    from serl_amd.data import MemoryEfficientReplayBuffer, ReplayBuffer
    from serl_amd.spaces import Box
    obs_space, act_space = state_spaces(4, 2)
    rb = ReplayBuffer(obs_space, act_space, 10)
    ob = orp.new_ring(((4,), np.float32), ((2,), np.float32), 10)
    rng = np.random.default_rng(0)
    for i in range(23):
        tr = dict(observations=rng.standard_normal(4).astype(np.float32),
                  next_observations=rng.standard_normal(4).astype(np.float32),
                  actions=rng.standard_normal(2).astype(np.float32), rewards=np.float32(i), masks=np.float32(1),
                  dones=False)
        rb.insert(tr)
        orp.ring_insert(ob, tr)
    assert len(rb) == 10 and rb._insert_index == ob["head"] == 3
    rb.seed(1)
    g = orp.make_rng(1)
    b = rb.sample(6)
    ref, idx = orp.ring_sample(ob, g, 6)
    assert np.array_equal(b["rewards"], ref["rewards"]) and np.array_equal(b["observations"], ref["observations"])
    assert np.array_equal(rb.sample(3, indx=np.array([0, 5, 9]))["rewards"], ob["data"]["rewards"][[0, 5, 9]])
    assert set(rb.sample(2, keys=["rewards", "masks"]).keys()) == {"rewards", "masks"}
    to, dl = rb.download(2, 5)
    assert to == 5 and len(dl["rewards"]) == 3
    it = rb.get_download_iterator()
    assert len(next(it)["rewards"]) == 10
    with pytest.raises(RuntimeError):
        next(it)
    with pytest.raises(TypeError):
        ReplayBuffer(object(), act_space, 4)
    with pytest.raises(AssertionError):
        rb.insert(dict(observations=np.zeros(4), next_observations=np.zeros(4), actions=np.zeros(2)))
    pspace, aspace = pixel_spaces(KEYS, 8, 16, 2, 5, 3)
    mb = MemoryEfficientReplayBuffer(pspace, aspace, 16, pixel_keys=KEYS)
    with pytest.raises(NotImplementedError):
        mb.sample(2, indx=np.array([1, 2]))
    bad = dict(pspace.spaces)
    bad["wrist_2"] = Box(0, 255, (8, 16, 3, 3), np.uint8)
    from serl_amd.spaces import Dict
    with pytest.raises(AssertionError):
        MemoryEfficientReplayBuffer(Dict(bad), aspace, 16, pixel_keys=KEYS)


def test_iterator_yields_batches_cpu():
    buf, obuf, _ = _fill(8, 16, 2, 64, 50, 10)
    buf.seed(3)
    it = buf.get_iterator(queue_size=2, sample_args=dict(batch_size=8, pack_obs_and_next_obs=True), device="cpu")
    b1, b2 = next(it), next(it)
    assert tuple(b1["observations"][KEYS[0]].shape) == (8, 8, 16, 3, 3)
    assert b1["actions"].shape[0] == 8 and not np.array_equal(np.asarray(b1["rewards"]), np.asarray(b2["rewards"]))


def test_dataset_split_filter_normalize():
    """Host-side Dataset helpers (serl/data/dataset.py:131-195) against a direct restatement."""
    from serl_amd.data import Dataset
    rng = np.random.default_rng(5)
    n = 57
    dones = np.zeros(n, bool)
    dones[[9, 20, 21, 40]] = True                      # last episode (41..56) unfinished
    data = dict(observations=dict(state=rng.standard_normal((n, 3)).astype(np.float32)),
                rewards=rng.standard_normal(n).astype(np.float32), dones=dones)
    ds = Dataset({k: (dict(v) if isinstance(v, dict) else v.copy()) for k, v in data.items()}, seed=4)
    starts, ends, rets = ds._trajectory_boundaries_and_returns()
    assert starts == [0, 10, 21, 22, 41] and ends == [10, 21, 22, 41]
    exp = [data["rewards"][s:e].astype(np.float64).sum() for s, e in zip(starts, ends)]
    assert np.allclose(rets, exp, rtol=1e-5, atol=1e-6)
    # split: the Generator's shuffle of arange(n), first int(n*ratio) rows train
    index = np.arange(n, dtype=np.int32)
    np.random.Generator(np.random.PCG64(np.random.SeedSequence(4))).shuffle(index)
    tr, te = ds.split(0.7)
    cut = int(n * 0.7)
    assert len(tr) == cut and len(te) == n - cut
    assert np.array_equal(tr.dataset_dict["rewards"], data["rewards"][index[:cut]])
    assert np.array_equal(te.dataset_dict["observations"]["state"], data["observations"]["state"][index[cut:]])
    # filter by threshold keeps whole episodes; the unfinished tail is dropped
    thr = sorted(exp)[2]
    keep = np.zeros(n, bool)
    for s, e, r in zip(starts, ends, rets):
        if r >= thr:
            keep[s:e] = True
    ds.filter(threshold=thr)
    assert len(ds) == keep.sum() and np.array_equal(ds.dataset_dict["rewards"], data["rewards"][keep])
    ds2 = Dataset({k: (dict(v) if isinstance(v, dict) else v.copy()) for k, v in data.items()})
    ds2.filter(take_top=50)
    thr2 = np.percentile(rets, 50)
    assert len(ds2) == sum(e - s for s, e, r in zip(starts, ends, rets) if r >= thr2)
    with pytest.raises(AssertionError):
        ds2.filter()
    ds3 = Dataset({k: (dict(v) if isinstance(v, dict) else v.copy()) for k, v in data.items()})
    ds3.normalize_returns(scaling=10)
    assert np.allclose(ds3.dataset_dict["rewards"], data["rewards"] / (max(rets) - min(rets)) * 10, rtol=1e-6)


def test_oracle_sample_jax_and_upload():
    """oracle.replay.ring_upload / ring_sample_jax: block append and the split order `key, rng = split(rng)`."""
    from oracle import prng as op
    ob = orp.new_ring(((4,), np.float32), ((2,), np.float32), 12)
    rng = np.random.default_rng(2)
    blk = lambda n: dict(observations=rng.standard_normal((n, 4)).astype(np.float32),
                         next_observations=rng.standard_normal((n, 4)).astype(np.float32),
                         actions=rng.standard_normal((n, 2)).astype(np.float32),
                         rewards=np.arange(n, dtype=np.float32), masks=np.ones(n, np.float32), dones=np.zeros(n, bool))
    orp.ring_upload(ob, blk(5))
    orp.ring_upload(ob, blk(7))
    assert ob["size"] == 12 and ob["head"] == 0
    key = op.prng_key(42)
    rng2, mx, batch, indx = orp.ring_sample_jax(ob, key, 9)
    ks = op.split(key)
    assert np.array_equal(rng2, ks[1]) and np.array_equal(indx, op.randint(ks[0], (9,), 0, 12))
    assert mx == indx.max() and np.array_equal(batch["rewards"], ob["data"]["rewards"][indx])
    assert indx.min() >= 0 and indx.max() < 12
