"""BASELINE config 4's reward networks on the GPU vs the CPU restatement (oracle/reward.py): BinaryClassifier inference
(`DrQClassifierRewardLearner.classify_reward`), VICE `encode` / `vice_reward` / `update`.  Synthetic frozen backbone (the
pretrained one is not available offline: backbone parity unpinned)."""
import numpy as np
import pytest
import torch

from helpers import agent_to_oracle, assert_info_close, make_pixel_batch, pixel_spaces

pytestmark = pytest.mark.gpu
KEYS = ("wrist_1", "wrist_2")
H = W = 64
RLPD = dict(latent_dim=64, num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)


def _ts_np(ts):
    lay = ts.store.layout
    return dict(params=ts.store.numpy_tree(), mu=lay.unflatten(ts.mu.cpu().numpy(), wrap=lambda a: a.copy()),
                nu=lay.unflatten(ts.nu.cpu().numpy(), wrap=lambda a: a.copy()), count=ts.step)


def test_classifier_reward_learner(dev, prec):
    from oracle import reward as orw
    from serl_amd.agents import DrQClassifierRewardLearner
    obs_space, act_space = pixel_spaces(KEYS, H, W, 1, 19, 7)
    agent = DrQClassifierRewardLearner.create(3, obs_space, act_space, pixel_keys=KEYS, encoder="mobilenet", **RLPD)
    assert agent.target_entropy == -7.0
    ostate, ocfg = agent_to_oracle(agent)
    rng = np.random.default_rng(0)
    # classify_reward on a single observation scaled to [0, 1] (the scripts' call) and on a batch
    imgs = {k: rng.integers(0, 256, size=(5, H, W, 3), dtype=np.uint8) for k in KEYS}
    img01 = {k: v.astype(np.float32) / np.float32(255.0) for k, v in imgs.items()}
    want = orw.binary_classifier(agent.classifier.store.numpy_tree(), img01, ocfg["backbone"], torch.float64).numpy()
    got = agent.classify_reward(img01)
    assert got.shape == (5, 1) and np.abs(got - want).max() <= 2e-4
    one = agent.classify_reward({k: v[2] for k, v in img01.items()})
    assert one.shape == (1,) and abs(float(one[0]) - float(want[2, 0])) <= 2e-4
    assert np.abs(agent.classify_reward(imgs) - want).max() <= 2e-4            # uint8 input: /255 on load
    # update: key-first augmentation keys, rng not advanced by them
    B, U = 4, 2
    batch = make_pixel_batch(rng, B * U, KEYS, H, W, 1, 19, 7, packed=True)
    collect = []
    ostate, oinfo = orw.classifier_reward_update(ostate, ocfg, batch, U, collect=collect)
    new, info = agent.update(batch, U, pixel_keys=KEYS)
    ws = new._ws(B)
    for side, okey in (("obs", "observations"), ("next", "next_observations")):
        for cam in KEYS:
            got = ws.tensor(f"aug_{side}_{cam}", (B * U, H, W, 3), torch.uint8).cpu().numpy()
            assert np.array_equal(got.reshape(B * U, H, W, 3, 1), collect[0][okey][cam])
    assert_info_close(info, oinfo, rtol=1e-3 if prec == "highest" else 5e-3)
    assert np.array_equal(new.rng, ostate["rng"])
    assert new.classifier is agent.classifier                                   # never updated by this learner


def test_vice_learner_reward_and_update(dev, prec):
    from oracle import reward as orw
    from serl_amd.agents import VICELearner
    obs_space, act_space = pixel_spaces(KEYS, H, W, 1, 19, 7)
    with pytest.raises(NotImplementedError):
        VICELearner.create(0, obs_space, act_space, pixel_keys=KEYS)             # encoder="d4pg": NameError upstream
    agent = VICELearner.create(5, obs_space, act_space, pixel_keys=KEYS, encoder="mobilenet", vice_dropout_rate=0.1, **RLPD)
    assert list(agent.vice_classifiers) == list(KEYS)
    ostate, ocfg = agent_to_oracle(agent)
    ostate["vice"] = {k: _ts_np(v) for k, v in agent.vice_classifiers.items()}
    rng = np.random.default_rng(1)
    obs = {k: rng.integers(0, 256, size=(6, H, W, 3, 1), dtype=np.uint8) for k in KEYS}
    # encode
    f = agent.encode(obs["wrist_1"]).cpu().numpy()
    fo = orw.encode(ocfg, obs["wrist_1"], torch.float64).numpy()
    assert f.shape == fo.shape == (6, 2, 2, 960)
    assert np.abs(f - fo).max() <= (2e-5 if prec == "highest" else 2e-4) * np.abs(fo).max()
    # vice_reward
    agent2, rews = agent.vice_reward(obs, KEYS)
    ostate2, orews = orw.vice_reward(ostate, ocfg, obs, KEYS, torch.float64)
    assert list(rews) == ["wrist_1", "wrist_2", "mean"]
    for k in rews:
        assert np.abs(rews[k].cpu().numpy() - orews[k].numpy()).max() <= 2e-4, k
    assert np.array_equal(agent2.rng, ostate2["rng"]) and not np.array_equal(agent2.rng, agent.rng)
    # update: rewards replaced by the logit of the mean VICE reward of the AUGMENTED observations
    B, U = 4, 2
    batch = make_pixel_batch(rng, B * U, KEYS, H, W, 1, 19, 7, packed=True)
    collect = []
    ostate, oinfo = orw.vice_update(ostate, ocfg, batch, U, collect=collect)
    new, info = agent.update(batch, U, pixel_keys=KEYS)
    assert list(info)[-1] == "vice_rews"
    assert_info_close(info, oinfo, rtol=1e-3 if prec == "highest" else 5e-3)
    assert np.array_equal(new.rng, ostate["rng"])


@pytest.mark.parametrize("layer_norm", [True, False])
def test_vice_update_classifier(dev, layer_norm):
    """vice_learner.py:316-474 at precision "highest": goal sampling, augmentation, mixup, label smoothing, mixup BCE and the
    gradient penalty, whose parameter gradient is a hand-derived double backward on the GPU and torch's
    autograd-of-autograd in the oracle.  Gradients of both cameras vs the float64 oracle, logged scalars, rng, Adam step."""
    import serl_amd
    from helpers import assert_tree_close
    from oracle import reward as orw
    from serl_amd.agents import VICELearner
    serl_amd.set_matmul_precision("highest")
    try:
        obs_space, act_space = pixel_spaces(KEYS, H, W, 1, 19, 7)
        rng = np.random.default_rng(2)
        pool = {k: rng.integers(0, 256, size=(9, H, W, 3, 1), dtype=np.uint8) for k in KEYS}
        kw = dict(RLPD, critic_layer_norm=layer_norm)
        agent = VICELearner.create(8, obs_space, act_space, pixel_keys=KEYS, encoder="mobilenet", vice_dropout_rate=0.1,
                                   vice_goal_pool=pool, **kw)
        ostate, ocfg = agent_to_oracle(agent)
        ostate["vice"] = {k: _ts_np(v) for k, v in agent.vice_classifiers.items()}
        ocfg.update(vice_lr=3e-4, vice_dropout_rate=0.1)
        B = 6
        batch = make_pixel_batch(rng, B, KEYS, H, W, 1, 19, 7, packed=True)
        collect = []
        ostate, oinfo = orw.update_classifier(ostate, ocfg, batch, KEYS, pool, dtype=torch.float64, collect=collect)
        new, info = agent.update_classifier(batch, KEYS)
        assert set(info) == set(oinfo)
        for k in oinfo:
            assert abs(info[k] - oinfo[k]) <= 1e-3 * max(abs(oinfo[k]), 1e-2), (k, info[k], oinfo[k])
        assert np.array_equal(new.rng, ostate["rng"])
        for c in collect:
            got = agent._shared["vice_last_grads"][c["camera"]].numpy_tree()
            assert_tree_close(got, c["grads"], rtol=2e-3, l2_rtol=2e-3, what=f"classifier grads {c['camera']}")
        for k in KEYS:
            assert new.vice_classifiers[k].step == 1 and agent.vice_classifiers[k].step == 0
            got = new.vice_classifiers[k].store.numpy_tree()
            assert_tree_close(got, ostate["vice"][k]["params"], atol=2.5 * 3e-4, mean_atol=2e-5, what=f"params {k}")
    finally:
        serl_amd.set_matmul_precision("bf16_3x")


def test_config4_full_size_vice_update_properties(dev):
    """BASELINE config 4 at FULL size: 2 x 128^2 cameras, stack 1, batch 256, UTD 20 (5120 rows per call), frozen
    MobileNetV3 encoder, VICE rewards inside `VICELearner.update` (vice_learner.py:277-314 -> sac_learner.py:221-238).
    The CPU oracle would need minutes, so size-independent properties (the pattern of
    test_full_size_update_properties), which exercise the 256-image backbone chunking and the `features ready` hook at
    5120 rows:
      1. augmented uint8 rows bit-exact vs the oracle's crop with the `key, rng = split` key order of this learner;
      2. the rewards the update used (logged as their mean, 'vice_rews') equal the logit of `vice_reward` evaluated by
         the public method on the same augmented observations;
      3. the same call twice is bit-identical in every parameter, moment and logged scalar;
      4. rng chain: one split per camera in vice_reward, then 4 per critic step + 2 (aug keys are not written back);
      5. everything finite, step counters advanced."""
    from oracle import learner as ol
    from oracle import prng as oprng
    from serl_amd.agents import VICELearner
    HH = WW = 128
    B, U = 256, 20
    n = B * U
    obs_space, act_space = pixel_spaces(KEYS, HH, WW, 1, 19, 7)
    agent = VICELearner.create(13, obs_space, act_space, pixel_keys=KEYS, encoder="mobilenet", vice_dropout_rate=0.1, **RLPD)
    rng = np.random.default_rng(4)
    batch = make_pixel_batch(rng, n, KEYS, HH, WW, 1, 19, 7, packed=True)
    a1, i1 = agent.update(batch, U, pixel_keys=KEYS)
    i1 = dict(i1)
    # 1. augmentation, every 41st row
    rows = np.arange(0, n, 41)
    ws = a1._ws(B)
    key, r = oprng.split(agent.rng)
    aug_obs = {}
    for side, sl in (("obs", slice(0, 1)), ("next", slice(1, 2))):
        kk = key
        for cam in KEYS:
            kcam, kk = oprng.split(kk)
            offs = ol.crop_offsets_fast(kcam, n)[rows]
            want = ol.random_crop_batch(batch["observations"][cam][rows][..., sl], offs)
            full = ws.tensor(f"aug_{side}_{cam}", (n, HH, WW, 3), torch.uint8)
            got = full[torch.as_tensor(rows, device=dev)]
            assert np.array_equal(got.cpu().numpy().reshape(len(rows), HH, WW, 3, 1), want), f"aug {side} {cam}"
            if side == "obs":
                aug_obs[cam] = full.cpu().numpy().reshape(n, HH, WW, 3, 1)
        key, r = oprng.split(r)
    # 2. rewards of the update == public vice_reward on the augmented observations
    _, rews = agent.vice_reward(aug_obs, KEYS)
    m = rews["mean"].double().cpu().numpy()
    want_mean = float(np.log(m / (1 - m)).mean())
    assert abs(i1["vice_rews"] - want_mean) <= 1e-4 * max(abs(want_mean), 1e-2), (i1["vice_rews"], want_mean)
    # 3. bit-identical repeat
    a2, i2 = agent.update(batch, U, pixel_keys=KEYS)
    assert i1 == dict(i2), (i1, dict(i2))
    for name in ("actor", "critic", "target_critic"):
        t1, t2 = getattr(a1, name), getattr(a2, name)
        assert torch.equal(t1.store.flat, t2.store.flat), f"{name} parameters differ between two identical calls"
        if name != "target_critic":
            assert torch.equal(t1.mu, t2.mu) and torch.equal(t1.nu, t2.nu), f"{name} Adam moments differ"
    assert torch.equal(a1.temp.buf, a2.temp.buf)
    # 4. rng chain
    e = agent.rng
    for _ in KEYS:
        _, e = oprng.split(e)
    for _ in range(4 * U + 2):
        _, e = oprng.split(e)
    assert np.array_equal(a1.rng, e)
    # 5.
    assert all(np.isfinite(v) for v in i1.values())
    assert bool(torch.isfinite(a1.critic.store.flat).all()) and bool(torch.isfinite(a1.actor.store.flat).all())
    assert int(a1.critic.step) == int(agent.critic.step) + U and int(a1.actor.step) == int(agent.actor.step) + 1
    for k in KEYS:
        assert a1.vice_classifiers[k] is agent.vice_classifiers[k]          # update() never trains the classifiers
