"""CPU tests of host logic: layouts / flax leaf names, initialisers, FrozenDict and space shims, agent
construction guards, and closed-form checks of the oracle's own ops."""
import math

import numpy as np
import pytest
import torch

from helpers import pixel_spaces, state_spaces
from oracle import learner as ol


def _e1_cfg(**kw):
    from serl_amd.networks.spec import NetConfig
    base = dict(action_dim=7, num_qs=10, critic_layer_norm=True, pixel_keys=("wrist_1", "wrist_2"), state_dim=19,
                H=128, W=128, stack=3, cnn_features=(32, 64, 128, 256), cnn_filters=(3, 3, 3, 3),
                cnn_strides=(2, 2, 2, 2), latent_dim=64)
    base.update(kw)
    return NetConfig(**base)


def test_layouts_and_names():
    from serl_amd.networks.spec import actor_layout, critic_layout, init_tree
    cfg = _e1_cfg()
    cl, al = critic_layout(cfg), actor_layout(cfg)
    n_critic = sum(int(np.prod(s)) for _, (o, s) in cl.entries.items())
    assert n_critic == 5246858                      # ~5.25 M
    assert cfg.enc_dim == 4096 and [L.OH for L in cfg.conv_layers()] == [63, 31, 15, 7]
    assert cl.prefix_size("image_encoder") == al.prefix_size("image_encoder") > 980992
    assert cl.shape("Ensemble_0", "VmapStateActionValue_0", "MLP_0", "Dense_0", "kernel") == (10, 583, 256)
    assert cl.shape("image_encoder", "Conv_4", "kernel") == (3, 3, 9, 32)          # camera 2, layer 1
    assert cl.shape("image_encoder", "SpatialLearnedEmbeddings_1", "kernel") == (7, 7, 256, 8)
    assert al.shape("Normal_0", "OutputDenseLogStd", "kernel") == (256, 7) and al.shape("Dense_0", "kernel") == (4096, 512)
    for _, (off, _s) in cl.entries.items():
        assert off % 64 == 0
    tree = init_tree(cl, [1, 2])
    flat = cl.flatten(tree)
    back = cl.unflatten(flat)
    k = back["Ensemble_0"]["VmapStateActionValue_0"]["MLP_0"]["Dense_0"]["kernel"]
    assert np.array_equal(k, tree["Ensemble_0"]["VmapStateActionValue_0"]["MLP_0"]["Dense_0"]["kernel"])
    lim = math.sqrt(6 / (583 + 256))
    assert abs(k).max() <= lim and abs(k).max() > 0.95 * lim and not np.array_equal(k[0], k[1])
    assert np.all(tree["LayerNorm_0"]["scale"] == 1) and np.all(tree["Dense_0"]["bias"] == 0)
    sle = tree["image_encoder"]["SpatialLearnedEmbeddings_0"]["kernel"]
    std = math.sqrt(1 / (7 * 7 * 256)) / 0.87962566103423978
    assert abs(sle).max() <= 2 * std + 1e-9 and abs(sle.std() / std - 0.8796) < 0.02
    with pytest.raises(ValueError):
        cl.flatten({**tree, "Dense_0": {"kernel": np.zeros((3, 3)), "bias": tree["Dense_0"]["bias"]}})
    # state-based names
    from serl_amd.networks.spec import NetConfig
    scfg = NetConfig(action_dim=6, num_qs=2, obs_dim=17)
    assert critic_layout(scfg).shape("VmapStateActionValue_0", "MLP_0", "Dense_0", "kernel") == (2, 23, 256)
    assert actor_layout(scfg).shape("OutputDenseMean", "bias") == (6,)


def test_frozen_dict_and_spaces():
    from serl_amd.frozen_dict import FrozenDict, freeze
    from serl_amd.spaces import Box, Dict
    d = freeze(dict(a=1, b=dict(c=2)))
    assert isinstance(d["b"], FrozenDict) and d["b"]["c"] == 2 and "a" in d and len(d) == 2
    d2 = d.copy(add_or_replace={"a": 5})
    assert d2["a"] == 5 and d["a"] == 1
    rest, v = d.pop("a")
    assert v == 1 and "a" not in rest and d.unfreeze() == dict(a=1, b=dict(c=2))
    with pytest.raises(TypeError):
        d.x = 3
    b = Box(0, 255, (4, 4, 3, 2), np.uint8)
    assert b.sample().shape == (4, 4, 3, 2) and b.low[..., 0].shape == (4, 4, 3)
    sp = Dict(dict(state=Box(-np.inf, np.inf, (3,), np.float32)))
    assert sp.sample()["state"].dtype == np.float32


def test_create_guards_without_gpu():
    from serl_amd.agents import DrQLearner, SACLearner
    obs_space, act_space = pixel_spaces(("a",), 16, 16, 1, 4, 2)
    with pytest.raises(NotImplementedError):
        DrQLearner.create(0, obs_space, act_space, pixel_keys=("a",), encoder="resnet")
    with pytest.raises(TypeError):
        DrQLearner.create(0, state_spaces(4, 2)[0], act_space)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            SACLearner.create(0, *state_spaces(4, 2))


# ----------------------------------------------------------------------------- oracle closed-form checks
def test_oracle_crop_matches_numpy_pad_and_vectorised_offsets():
    from oracle import prng
    key = prng.prng_key(9)
    offs = ol.crop_offsets(key, 33)
    assert np.array_equal(offs, ol.crop_offsets_fast(key, 33))
    img = np.random.default_rng(0).integers(0, 256, size=(33, 12, 10, 3, 2), dtype=np.uint8)
    out = ol.random_crop_batch(img, offs)
    yy = np.clip(np.arange(12)[None, :] + offs[:, :1] - 4, 0, 11)
    xx = np.clip(np.arange(10)[None, :] + offs[:, 1:] - 4, 0, 9)
    ref = img[np.arange(33)[:, None, None], yy[:, :, None], xx[:, None, :]]
    assert np.array_equal(out, ref)


def test_oracle_logprob_adam_polyak_layernorm():
    from scipy.stats import norm
    rng = np.random.default_rng(1)
    mean = torch.tensor(rng.standard_normal((5, 3))); lstd = torch.tensor(rng.uniform(-2, 1, (5, 3)))
    x = torch.tensor(rng.standard_normal((5, 3)))
    lp = ol.tanh_normal_log_prob(mean, lstd, x).numpy()
    a = np.tanh(x.numpy())
    ref = norm.logpdf(x.numpy(), mean.numpy(), np.exp(lstd.numpy())).sum(-1) - np.log(1 - a * a).sum(-1)
    np.testing.assert_allclose(lp, ref, rtol=1e-9)
    # Adam: scalar recurrence
    ts = dict(params=dict(w=np.float32([1.0])), mu=dict(w=np.float32([0.0])), nu=dict(w=np.float32([0.0])), count=0)
    m = v = 0.0
    p = 1.0
    for t, g in enumerate([0.5, -0.25, 2.0], 1):
        ts = ol.adam_step(ts, dict(w=np.float32([g])), 1e-2)
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        p -= 1e-2 * (m / (1 - 0.9 ** t)) / (math.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
        assert abs(float(ts["params"]["w"][0]) - p) < 1e-6 and ts["count"] == t
    assert abs(float(ol.polyak(dict(w=np.float32([2.0])), dict(w=np.float32([1.0])), 0.005)["w"][0]) - 1.005) < 1e-7
    # LayerNorm (eps 1e-6, biased variance)
    h = torch.tensor(rng.standard_normal((4, 16)))
    y = ol.layer_norm(dict(scale=torch.ones(16, dtype=torch.float64), bias=torch.zeros(16, dtype=torch.float64)), h)
    ref = (h - h.mean(-1, keepdim=True)) / torch.sqrt(h.var(-1, unbiased=False, keepdim=True) + 1e-6)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-9, atol=1e-12)


def test_oracle_sac_update_quirks():
    """UTD slices are contiguous; the actor step uses the LAST minibatch; critic info is the last step's."""
    from serl_amd.networks.spec import NetConfig, actor_layout, critic_layout, init_tree
    from oracle import prng
    cfg = NetConfig(action_dim=2, num_qs=3, obs_dim=4, hidden_dims=(8, 8), critic_layer_norm=True)
    cp, ap = init_tree(critic_layout(cfg), [0]), init_tree(actor_layout(cfg), [1])
    z = lambda t: ol.tree_map(np.zeros_like, t)
    state = dict(rng=prng.prng_key(1), actor=dict(params=ap, mu=z(ap), nu=z(ap), count=0),
                 critic=dict(params=cp, mu=z(cp), nu=z(cp), count=0), target_critic=dict(params=cp),
                 temp=dict(params=dict(log_temp=np.float32(0)), mu=dict(log_temp=np.float32(0)),
                           nu=dict(log_temp=np.float32(0)), count=0))
    ocfg = dict(discount=0.99, tau=0.005, target_entropy=-1.0, num_qs=3, num_min_qs=2, backup_entropy=True,
                actor_lr=3e-4, critic_lr=3e-4, temp_lr=3e-4, critic_layer_norm=True, pixel_keys=(), cnn_strides=())
    from helpers import make_state_batch
    batch = make_state_batch(np.random.default_rng(0), 12, 4, 2)
    col = []
    new, info = ol.sac_update(state, ocfg, batch, 3, collect=col)
    assert len(col) == 4 and new["critic"]["count"] == 3 and new["actor"]["count"] == 1 and new["temp"]["count"] == 1
    assert set(info) == {"actor_loss", "entropy", "critic_loss", "q", "temperature", "temperature_loss"}
    assert col[0]["qs"].shape == (3, 4) and len(col[0]["subsample"]) == 2
    # rng advanced by 4 splits per critic step + 2 for the actor
    r = state["rng"]
    for _ in range(3 * 4 + 2):
        r = prng.split(r)[1]
    assert np.array_equal(new["rng"], r)
    with pytest.raises(AssertionError):
        ol.sac_update(state, ocfg, batch, 5)


def test_oracle_ddpg_update_and_noise():
    """ddpg_learner.py: one critic, actor loss = -mean q, no temperature; exploration noise = 0.05 x the policy
    sample's own standard-normal draw (both keys come from split(self.rng), :99 and :102)."""
    from serl_amd.networks.spec import NetConfig, actor_layout, critic_layout, init_tree
    from oracle import prng
    cfg = NetConfig(action_dim=2, num_qs=1, obs_dim=4, hidden_dims=(8, 8), critic_layer_norm=False)
    cp, ap = init_tree(critic_layout(cfg), [0]), init_tree(actor_layout(cfg), [1])
    z = lambda t: ol.tree_map(np.zeros_like, t)
    state = dict(rng=prng.prng_key(3), actor=dict(params=ap, mu=z(ap), nu=z(ap), count=0),
                 critic=dict(params=cp, mu=z(cp), nu=z(cp), count=0), target_critic=dict(params=cp))
    ocfg = dict(discount=0.99, tau=0.005, num_qs=1, num_min_qs=1, backup_entropy=False, actor_lr=3e-4,
                critic_lr=3e-4, critic_layer_norm=False, pixel_keys=(), cnn_strides=())
    from helpers import make_state_batch
    batch = make_state_batch(np.random.default_rng(0), 12, 4, 2)
    col = []
    new, info = ol.ddpg_update(state, ocfg, batch, 3, collect=col)
    assert list(info) == ["actor_loss", "critic_loss", "q"] and new["critic"]["count"] == 3 and new["actor"]["count"] == 1
    assert col[0]["qs"].shape == (1, 4) and list(col[0]["subsample"]) == [0]
    # actor loss = -mean q(obs, a) at the post-UTD critic, a = the collected policy sample
    mini = ol.tree_map(lambda a: a[8:], batch)
    q = ol.critic_qs(ol.to_torch(new["critic"]["params"], torch.float32), mini["observations"],
                     torch.tensor(col[-1]["actions"]), ocfg, torch.float32)
    assert abs(info["actor_loss"] + float(q.mean())) < 1e-6
    r = state["rng"]
    for _ in range(3 * 4 + 2):
        r = prng.split(r)[1]
    assert np.array_equal(new["rng"], r)
    ob = np.random.default_rng(1).standard_normal(4).astype(np.float32)
    a, st2 = ol.ddpg_sample_actions(new, ocfg, ob)
    a0, st3 = ol.sample_actions(new, ocfg, ob)
    key = prng.split(new["rng"])[0]
    np.testing.assert_allclose(a - a0, 0.05 * prng.normal(key, (2,)), atol=1e-6)
    assert np.array_equal(st2["rng"], st3["rng"])


def test_update_info_is_a_lazy_mapping():
    """update() returns its scalars as a Mapping that reads the device buffer on first access (like the reference's
    jax Arrays); keys, order, dict() / == / iteration behave like the reference's dict."""
    from serl_amd.agents.sac.sac_learner import UpdateInfo
    raw = torch.tensor([8.0, 4.0, -6.0, 0.0, 0.25, 0.5, 1.5, 0.0])      # loss sum, q sum, actor-loss sum, -, temp, temp loss, entropy
    info = UpdateInfo(raw, global_batch=2, num_qs=2)
    assert info._host is None                                            # nothing fetched yet
    assert list(info) == ["actor_loss", "entropy", "critic_loss", "q", "temperature", "temperature_loss"]
    assert info["critic_loss"] == 2.0 and info["q"] == 1.0 and info["actor_loss"] == -3.0 and info["entropy"] == 1.5
    assert dict(info) == {"actor_loss": -3.0, "entropy": 1.5, "critic_loss": 2.0, "q": 1.0, "temperature": 0.25,
                          "temperature_loss": 0.5}
    assert info == dict(info) and len(info) == 6 and {f"training/{k}": v for k, v in info.items()}
    ddpg = UpdateInfo(raw, 2, 1, keys=("actor_loss", "critic_loss", "q"))
    assert list(ddpg.items()) == [("actor_loss", -3.0), ("critic_loss", 4.0), ("q", 2.0)]
