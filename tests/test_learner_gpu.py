"""End-to-end GPU parity: N `update()` calls of the HIP learners vs the CPU oracle from identical
parameters, batches and RNG.

Stated tolerances (float path): losses / Q / entropy / temperature rel 1e-3 after N updates
(north_star); per-leaf gradients: critic 2e-4 of the leaf's max magnitude against the float64 oracle when
the step is comparable (UTD 1), else 1e-3 against the float32 oracle (both sides are fp32 and drift apart
through Adam's sign-like first steps), actor 5e-4 at UTD 1 and 1e-2 after several critic steps (its
Q-gradient is a mean over the ensemble that nearly cancels at init, so it amplifies that drift);
augmentation and indices bit-exact.
The tight per-entry gradient bounds hold at matmul precision "highest" (exact fp32 MFMA).  At the default
"bf16_3x" (each product accurate to ~2^-17) a pre-activation within ~1e-5 of zero can land on the other side
of a ReLU, which moves single gradient entries by a whole term; gradients are then bounded per leaf in relative
L2 norm (2e-2; 3e-2 after several Adam steps) and by 5e-2 of the leaf's max magnitude.  Losses / Q / entropy / temperature at "bf16_3x": rel 1e-3
for every update() call with one critic step, started from the product's own state (the oracle is re-synchronised
before each call), rel 5e-3 when the call holds several Adam steps (UTD > 1): Adam's sign-like first steps turn a
1e-6 gradient difference into a +-lr parameter difference, so two correct implementations separate at that rate.
Parameters after Adam are compared loosely (Adam's first steps are ~sign(g): a gradient that is noise
around 0 may move a weight by +-lr in either implementation)."""
import numpy as np
import pytest
import torch

from helpers import (agent_to_oracle, assert_info_close, assert_tree_close, make_pixel_batch, make_state_batch,
                     pixel_spaces, state_spaces)

pytestmark = pytest.mark.gpu


def _check_params(agent, ostate, lr, steps):
    for name in ("actor", "critic", "target_critic"):
        got = getattr(agent, name).store.numpy_tree()
        ref = ostate[name]["params"]
        assert_tree_close(got, ref, atol=2.5 * lr * steps, mean_atol=2e-5, what=f"{name} params")


def _check_grads(agent, name, ref64, ref32, prec, rtol, noise_ref=None):
    got = agent._shared[name].numpy_tree()
    if prec == "highest":
        if noise_ref is not None:
            assert_tree_close(got, ref64, rtol=rtol, what=name, noise_ref=noise_ref)
        else:
            assert_tree_close(got, ref32, rtol=rtol, what=name)
    else:
        ref = ref64 if noise_ref is not None else ref32
        assert_tree_close(got, ref, rtol=5e-2, l2_rtol=2e-2 if noise_ref is not None else 3e-2, what=name)


@pytest.mark.parametrize("cfgname", ["sac_c1", "rlpd_u1", "rlpd_c2", "redq_m3", "droq_ln", "droq_plain"])
def test_state_learner_matches_oracle(dev, cfgname, prec):
    from oracle import learner as ol
    from serl_amd.agents import SACLearner
    obs_space, act_space = state_spaces(17, 6)
    if cfgname == "sac_c1":       # BASELINE config 1
        kw = dict(num_qs=2, critic_layer_norm=False, backup_entropy=True, init_temperature=1.0)
        B, U = 256, 1
    elif cfgname == "rlpd_u1":    # RLPD networks, one critic step per call: tight gradient check
        kw = dict(num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
        B, U = 256, 1
    elif cfgname == "rlpd_c2":    # BASELINE config 2 (UTD shortened for the CPU oracle)
        kw = dict(num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
        B, U = 256, 4
    elif cfgname.startswith("droq"):   # critic_dropout_rate (sac_learner.py:47, mlp.py:26-29): Dense -> Dropout -> [LN] -> ReLU in
        # the target / critic / actor-step applies; mask stream build-defined (oracle.learner.critic_dropout)
        kw = dict(num_qs=2 if cfgname == "droq_plain" else 4, num_min_qs=None if cfgname == "droq_plain" else 2,
                  critic_layer_norm=cfgname == "droq_ln", critic_dropout_rate=0.01 if cfgname == "droq_ln" else 0.2,
                  backup_entropy=True, init_temperature=0.5)
        B, U = 96, (1 if cfgname == "droq_ln" else 2)
    else:                          # general subsample path (non-arithmetic index sets)
        kw = dict(num_qs=6, num_min_qs=3, critic_layer_norm=True, backup_entropy=True, init_temperature=0.5)
        B, U = 64, 3
    agent = SACLearner.create(42, obs_space, act_space, **kw)
    ostate, ocfg = agent_to_oracle(agent)
    rng = np.random.default_rng(0)
    info_tol = 1e-3 if (prec == "highest" or U == 1) else 5e-3
    for it in range(3):
        batch = make_state_batch(rng, B * U, 17, 6)
        collect, c64 = [], []
        if prec != "highest":
            ostate, _ = agent_to_oracle(agent)                                        # per-call parity (see module doc)
        ol.sac_update(ostate, ocfg, batch, U, dtype=torch.float64, collect=c64)      # float64 gradients
        ostate, oinfo = ol.sac_update(ostate, ocfg, batch, U, collect=collect)
        agent, info = agent.update(batch, U)
        assert_info_close(info, oinfo, rtol=info_tol)
        assert np.array_equal(agent.rng, ostate["rng"])
        if U == 1:   # with U > 1 the f64 run's parameters drift from the f32 run's before the last step
            _check_grads(agent, "critic_grads", c64[-2]["grads"], None, prec, 2e-4, noise_ref=collect[-2]["grads"])
            _check_grads(agent, "actor_grads", c64[-1]["grads"], None, prec, 5e-4, noise_ref=collect[-1]["grads"])
        else:
            _check_grads(agent, "critic_grads", None, collect[-2]["grads"], prec, 1e-3)
            _check_grads(agent, "actor_grads", None, collect[-1]["grads"], prec, 1e-2)
    _check_params(agent, ostate, 3e-4, 3 * U)
    if prec != "highest":
        ostate, _ = agent_to_oracle(agent)
    # acting
    ob = rng.standard_normal(17).astype(np.float32)
    a, agent2 = agent.sample_actions(ob)
    oa, ostate2 = ol.sample_actions(ostate, ocfg, ob)
    assert np.allclose(a, oa, atol=2e-4) and np.array_equal(agent2.rng, ostate2["rng"])
    assert np.allclose(agent.eval_actions(ob), ol.eval_actions(ostate, ocfg, ob), atol=2e-4)
    # immutability: the old agent still acts with its old parameters
    assert np.array_equal(agent.rng, ostate["rng"])


@pytest.mark.parametrize("variant", ["e1_packed", "e1_packed_utd2", "e0_unpacked_small", "e0_small_droq"])
def test_drq_learner_matches_oracle(dev, variant, prec):
    from oracle import learner as ol
    from serl_amd.agents import DrQLearner
    if variant.startswith("e1_packed"):   # BASELINE config 3 shapes, RLPD settings, tiny batch for the CPU oracle
        H = W = 128; S = 3; B, U = 8, (2 if variant.endswith("utd2") else 1)
        kw = dict(cnn_features=(32, 64, 128, 256), cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64,
                  num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
        packed = True
    else:                           # DrQLearner defaults (stride 2,1,1,1), stack 1, unpacked batch
        H = W = 32; S = 1; B, U = 6, 1
        kw = dict(num_qs=2, critic_layer_norm=False, backup_entropy=True)
        if variant.endswith("droq"):           # pixel learner with critic_dropout_rate (drq_learner.py:47,100-106)
            kw = dict(num_qs=4, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, critic_dropout_rate=0.05)
        packed = False
    keys = ("wrist_1", "wrist_2")
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = DrQLearner.create(42, obs_space, act_space, pixel_keys=keys, **kw)
    ostate, ocfg = agent_to_oracle(agent)
    rng = np.random.default_rng(1)
    for it in range(2):
        batch = make_pixel_batch(rng, B * U, keys, H, W, S, 19, 7, packed=packed)
        collect, c64, csync = [], [], []
        if U == 1:   # single-step gradients from the product's own current state: tight, vs float64
            osync, _ = agent_to_oracle(agent)
            ol.drq_update(osync, ocfg, batch, U, dtype=torch.float64, collect=c64)
            ol.drq_update(osync, ocfg, batch, U, collect=csync)
        if prec != "highest":
            ostate, _ = agent_to_oracle(agent)                                     # per-call parity (see module doc)
        ostate, oinfo = ol.drq_update(ostate, ocfg, batch, U, collect=collect)     # free-running trajectory at "highest"
        agent, info = agent.update(batch, U, pixel_keys=keys)
        # augmentation bit-exact (uint8)
        ws = agent._ws(B)
        for side, okey in (("obs", "observations"), ("next", "next_observations")):
            for cam in keys:
                got = ws.tensor(f"aug_{side}_{cam}", (B * U, H, W, 3 * S), torch.uint8).cpu().numpy()
                assert np.array_equal(got.reshape(B * U, H, W, 3, S), collect[0][okey][cam]), f"aug {side} {cam}"
        assert_info_close(info, oinfo, rtol=1e-3 if (prec == "highest" or U == 1) else 5e-3)
        assert np.array_equal(agent.rng, ostate["rng"])
        if U == 1:
            _check_grads(agent, "critic_grads", c64[-2]["grads"], None, prec, 3e-4, noise_ref=csync[-2]["grads"])
            _check_grads(agent, "actor_grads", c64[-1]["grads"], None, prec, 5e-4, noise_ref=csync[-1]["grads"])
    _check_params(agent, ostate, 3e-4, 2 * U)
    ob = {k: rng.integers(0, 256, size=(H, W, 3, S), dtype=np.uint8) for k in keys}
    ob["state"] = rng.standard_normal(19).astype(np.float32)
    osync, _ = agent_to_oracle(agent)          # acting parity from the product's own parameters
    a, agent2 = agent.sample_actions(ob)
    oa, osync2 = ol.sample_actions(osync, ocfg, ob)
    assert np.allclose(a, oa, atol=5e-4) and np.array_equal(agent2.rng, osync2["rng"])
    assert np.allclose(agent.eval_actions(ob), ol.eval_actions(osync, ocfg, ob), atol=5e-4)


def test_device_replay_feeds_update(dev):
    """MemoryEfficientReplayBuffer(device=cuda) iterator -> lazy FrameWindows -> fused gather+crop in update():
    same result as the host buffer + packed numpy batch (and both equal the oracle's sampled windows)."""
    from oracle import replay as orp
    from serl_amd.agents import DrQLearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    keys = ("wrist_1", "wrist_2")
    H = W = 32; S = 3
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    bufs = [MemoryEfficientReplayBuffer(obs_space, act_space, 200, pixel_keys=keys, device=d) for d in (None, "cuda")]
    spec = {k: ((H, W, 3, S), np.uint8) for k in keys}
    spec["state"] = ((19,), np.float32)
    obuf = orp.new_frame_buffer(spec, ((7,), np.float32), 200, keys)
    rng = np.random.default_rng(3)
    from helpers import rollout_transitions
    for tr in rollout_transitions(rng, 260, keys, H, W, S, 19, 7, episode_len=23):   # wraps the ring
        for b in bufs:
            b.insert(tr)
        orp.frame_insert(obuf, tr)
    assert np.array_equal(bufs[0]._is_correct_index, obuf["valid"]) and np.array_equal(bufs[1]._is_correct_index, obuf["valid"])
    for b in bufs:
        b.seed(7)
    orng = orp.make_rng(7)
    args = dict(batch_size=24, pack_obs_and_next_obs=True, demo_batch_size=12, demo_size=50)
    host = bufs[0].sample(**args)
    dev_np = bufs[1].sample(**args)            # device gather + D2H
    oidx = orp.frame_sample_indices(obuf, orng, 24, 12, 50)
    obatch = orp.frame_gather(obuf, oidx, True)
    for k in keys:
        assert np.array_equal(host["observations"][k], obatch["observations"][k])
        assert np.array_equal(dev_np["observations"][k], obatch["observations"][k])
    assert np.array_equal(host["actions"], obatch["actions"])
    # lazy iterator batch -> update == packed host batch -> update
    for b in bufs:
        b.seed(11)
    it = bufs[1].get_iterator(sample_args=dict(batch_size=12, pack_obs_and_next_obs=True))
    lazy = next(it)
    hostb = bufs[0].sample(batch_size=12, pack_obs_and_next_obs=True)
    agent = DrQLearner.create(0, obs_space, act_space, pixel_keys=keys, num_qs=2)
    a1, i1 = agent.update(lazy, 2, pixel_keys=keys)
    a2, i2 = agent.update(hostb, 2, pixel_keys=keys)
    assert i1 == i2
    assert torch.equal(a1.critic.store.flat, a2.critic.store.flat)


def test_full_size_update_properties(dev):
    """BASELINE config 3 with the RLPD settings at FULL size (2 x 128^2 cameras, stack 3, batch 256, 10/2 critics,
    UTD 20: 5120 rows per call), where the CPU oracle would take minutes -- size-independent properties instead:
      1. the augmented uint8 images of a strided sample of the 5120 rows are bit-exact against the oracle's crop of
         the host buffer's windows (index stream, frame-stack gather, _unpack, per-row threefry offsets, edge clamp);
      2. the same update() run twice from the same agent and batch is bit-identical in every parameter, optimizer
         moment and logged scalar (the parameter-gradient kernels run on side streams; split-K slabs are reduced in
         a fixed order -- a race or an atomic accumulation would show here);
      3. the rng chain advances exactly as the oracle's split sequence; all outputs are finite."""
    from oracle import learner as ol
    from oracle import prng as oprng
    from serl_amd.agents import DrQLearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    from helpers import rollout_transitions
    keys = ("wrist_1", "wrist_2")
    H = W = 128; S = 3; B, U = 256, 20
    n = B * U
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    bufs = [MemoryEfficientReplayBuffer(obs_space, act_space, 1500, pixel_keys=keys, device=d) for d in (None, "cuda")]
    rng = np.random.default_rng(5)
    for tr in rollout_transitions(rng, 1200, keys, H, W, S, 19, 7, episode_len=100):
        for b in bufs:
            b.insert(tr)
    args = dict(batch_size=n, pack_obs_and_next_obs=True, demo_batch_size=n // 2, demo_size=300)
    for b in bufs:
        b.seed(3)
    lazy = next(bufs[1].get_iterator(queue_size=1, sample_args=args))
    host = bufs[0].sample(**args)
    agent = DrQLearner.create(7, obs_space, act_space, pixel_keys=keys, cnn_features=(32, 64, 128, 256),
                              cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64, num_qs=10,
                              num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
    a1, i1 = agent.update(lazy, U, pixel_keys=keys)
    # 1. augmentation, every 37th row
    rows = np.arange(0, n, 37)
    ws = a1._ws(B)
    r, key = oprng.split(agent.rng)
    for side, sl in (("obs", slice(0, S)), ("next", slice(1, S + 1))):
        kk = key
        for cam in keys:
            kcam, kk = oprng.split(kk)
            offs = ol.crop_offsets_fast(kcam, n)[rows]
            want = ol.random_crop_batch(host["observations"][cam][rows][..., sl], offs)
            got = ws.tensor(f"aug_{side}_{cam}", (n, H, W, 3 * S), torch.uint8)[torch.as_tensor(rows, device=dev)]
            assert np.array_equal(got.cpu().numpy().reshape(len(rows), H, W, 3, S), want), f"aug {side} {cam}"
        r, key = oprng.split(r)
    # 2. bit-identical repeat
    a2, i2 = agent.update(lazy, U, pixel_keys=keys)
    assert i1 == i2, (i1, i2)
    for name in ("actor", "critic", "target_critic"):
        t1, t2 = getattr(a1, name), getattr(a2, name)
        assert torch.equal(t1.store.flat, t2.store.flat), f"{name} parameters differ between two identical calls"
        if name != "target_critic":
            assert torch.equal(t1.mu, t2.mu) and torch.equal(t1.nu, t2.nu), f"{name} Adam moments differ"
    assert torch.equal(a1.temp.buf, a2.temp.buf)
    # 3. rng chain: drq_learner.py:174-177 (two splits), sac_learner.py:166-209 (4 per critic step), :121-122 (2)
    e = agent.rng
    e, _ = oprng.split(e)
    e, _ = oprng.split(e)
    for _ in range(4 * U + 2):
        _, e = oprng.split(e)
    assert np.array_equal(a1.rng, e)
    assert all(np.isfinite(v) for v in i1.values())
    assert bool(torch.isfinite(a1.critic.store.flat).all()) and bool(torch.isfinite(a1.actor.store.flat).all())
    assert int(a1.critic.step) == int(agent.critic.step) + U and int(a1.actor.step) == int(agent.actor.step) + 1


def test_checkpoint_roundtrip_through_agent(dev, tmp_path):
    """The reference's save/restore pattern (cable_route_...py:164-170,184-207): save each TrainState with its prefix,
    restore into a freshly created agent with agent.replace(...): acting and the next update are bit-identical."""
    from serl_amd.agents import DrQLearner
    from serl_amd.utils import restore_checkpoint_, save_checkpoint
    keys = ("wrist_1", "wrist_2")
    H = W = 32; S = 1
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    kw = dict(pixel_keys=keys, num_qs=4, num_min_qs=2, critic_layer_norm=True, backup_entropy=False)
    rng = np.random.default_rng(0)
    agent = DrQLearner.create(1, obs_space, act_space, **kw)
    agent, _ = agent.update(make_pixel_batch(rng, 8, keys, H, W, S, 19, 7, packed=True), 2, pixel_keys=keys)
    for name in ("actor", "critic", "target_critic", "temp"):
        save_checkpoint(tmp_path, getattr(agent, name), step=20000, prefix=f"{name}_", keep=1000)
    fresh = DrQLearner.create(99, obs_space, act_space, **kw)
    fresh = fresh.replace(rng=agent.rng, **{name: restore_checkpoint_(str(tmp_path / f"{name}_20000"), getattr(fresh, name), None)
                                            for name in ("actor", "critic", "target_critic", "temp")})
    ob = {k: rng.integers(0, 256, size=(H, W, 3, S), dtype=np.uint8) for k in keys}
    ob["state"] = rng.standard_normal(19).astype(np.float32)
    assert np.array_equal(agent.eval_actions(ob), fresh.eval_actions(ob))
    batch = make_pixel_batch(rng, 8, keys, H, W, S, 19, 7, packed=True)
    a1, i1 = agent.update(batch, 2, pixel_keys=keys)
    a2, i2 = fresh.update(batch, 2, pixel_keys=keys)
    assert i1 == i2
    assert torch.equal(a1.critic.store.flat, a2.critic.store.flat) and torch.equal(a1.actor.store.flat, a2.actor.store.flat)
    assert torch.equal(a1.temp.buf, a2.temp.buf) and a1.critic.step == a2.critic.step


@pytest.mark.parametrize("layer_norm", [False, True])
def test_ddpg_learner_matches_oracle(dev, layer_norm, prec):
    """DDPGLearner (serl/agents/ddpg/ddpg_learner.py): one critic, -mean(q) actor loss, no temperature, info keys
    actor_loss / critic_loss / q, exploration noise drawn with the policy sample's key."""
    from oracle import learner as ol
    from serl_amd.agents import DDPGLearner
    obs_space, act_space = state_spaces(17, 6)
    agent = DDPGLearner.create(42, obs_space, act_space, critic_layer_norm=layer_norm)
    assert agent.num_qs == 1
    rng = np.random.default_rng(0)
    B, U = 64, 2
    for it in range(3):
        ostate, ocfg = agent_to_oracle(agent)
        ocfg.update(num_qs=1, num_min_qs=1, backup_entropy=False)
        batch = make_state_batch(rng, B * U, 17, 6)
        ostate, oinfo = ol.ddpg_update(ostate, ocfg, batch, U)
        agent, info = agent.update(batch, U)
        assert list(info) == ["actor_loss", "critic_loss", "q"]
        assert_info_close(info, oinfo, rtol=1e-3 if prec == "highest" else 5e-3)
        assert np.array_equal(agent.rng, ostate["rng"])
    _check_params(agent, ostate, 3e-4, U)
    ob = rng.standard_normal(17).astype(np.float32)
    osync, ocfg = agent_to_oracle(agent)
    a, agent2 = agent.sample_actions(ob)
    oa, osync2 = ol.ddpg_sample_actions(osync, ocfg, ob)
    assert np.allclose(a, oa, atol=5e-4) and np.array_equal(agent2.rng, osync2["rng"])
    assert np.abs(a).max() > 0 and a.shape == (6,)
    assert np.allclose(agent.eval_actions(ob), ol.eval_actions(osync, ocfg, ob), atol=5e-4)


@pytest.mark.parametrize("packed", [True, False])
def test_pixel_ddpg_learner_matches_oracle(dev, packed, prec):
    """PixelDDPGLearner (serl/agents/ddpg/pixel_ddpg_learner.py:148-181): DrQ's augmentation with `key, rng = split`
    key order, the rng NOT advanced by the augmentation splits, then the DDPG update (1 critic, 3 info keys)."""
    from oracle import learner as ol
    from serl_amd.agents import PixelDDPGLearner
    keys = ("wrist_1", "wrist_2")
    H = W = 32; S = 2; B, U = 6, 2
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = PixelDDPGLearner.create(5, obs_space, act_space, pixel_keys=keys, critic_layer_norm=True)
    assert agent.num_qs == 1
    rng = np.random.default_rng(3)
    for it in range(2):
        ostate, ocfg = agent_to_oracle(agent)
        ocfg.update(num_qs=1, num_min_qs=1, backup_entropy=False)
        batch = make_pixel_batch(rng, B * U, keys, H, W, S, 19, 7, packed=packed)
        collect = []
        ostate, oinfo = ol.pixel_ddpg_update(ostate, ocfg, batch, U, collect=collect)
        agent, info = agent.update(batch, U, pixel_keys=keys)
        ws = agent._ws(B)
        for side, okey in (("obs", "observations"), ("next", "next_observations")):
            for cam in keys:
                got = ws.tensor(f"aug_{side}_{cam}", (B * U, H, W, 3 * S), torch.uint8).cpu().numpy()
                assert np.array_equal(got.reshape(B * U, H, W, 3, S), collect[0][okey][cam]), f"aug {side} {cam}"
        assert list(info) == ["actor_loss", "critic_loss", "q"]
        assert_info_close(info, oinfo, rtol=1e-3 if prec == "highest" else 5e-3)
        assert np.array_equal(agent.rng, ostate["rng"])
    _check_params(agent, ostate, 3e-4, U)
    ob = {k: rng.integers(0, 256, size=(H, W, 3, S), dtype=np.uint8) for k in keys}
    ob["state"] = rng.standard_normal(19).astype(np.float32)
    osync, ocfg = agent_to_oracle(agent)
    a, agent2 = agent.sample_actions(ob)
    oa, osync2 = ol.ddpg_sample_actions(osync, ocfg, ob)
    assert np.allclose(a, oa, atol=5e-4) and np.array_equal(agent2.rng, osync2["rng"])


@pytest.mark.parametrize("repeat", [0, 1, 2])
def test_learner_thread_and_acting_thread(dev, repeat):
    """The reference's threading pattern (serl_examples/*: learner_thread + main loop): one thread calls update()
    while another keeps acting with an OLDER agent copy.  Agents are immutable, so the acting thread's actions must
    equal those of the same calls made with nothing else running, and the learner's result must equal a solo run."""
    import threading
    from serl_amd.agents import DrQLearner
    keys = ("wrist_1", "wrist_2")
    H = W = 32; S = 1
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    kw = dict(pixel_keys=keys, num_qs=4, num_min_qs=2, critic_layer_norm=True, backup_entropy=False)
    rng = np.random.default_rng(0)
    agent0 = DrQLearner.create(3, obs_space, act_space, **kw)
    batches = [make_pixel_batch(rng, 16, keys, H, W, S, 19, 7, packed=True) for _ in range(4)]
    obs = [dict({k: rng.integers(0, 256, size=(H, W, 3, S), dtype=np.uint8) for k in keys},
                state=rng.standard_normal(19).astype(np.float32)) for _ in range(40)]

    def learn(agent):
        for b in batches:
            agent, info = agent.update(b, 4, pixel_keys=keys)
        return agent, dict(info)

    def act(agent):
        out = []
        for ob in obs:
            a, agent = agent.sample_actions(ob)
            out.append(a)
        return np.stack(out)

    solo_agent, solo_info = learn(agent0)
    solo_actions = act(agent0)
    res, errs = {}, []

    def run(name, fn):
        try:
            torch.cuda.set_device(0)
            res[name] = fn(agent0)
        except Exception as e:      # noqa: BLE001
            errs.append((name, repr(e)))
    ts = [threading.Thread(target=run, args=("learn", learn)), threading.Thread(target=run, args=("act", act))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs, errs
    assert np.array_equal(res["act"], solo_actions)
    assert res["learn"][1] == solo_info
    assert torch.equal(res["learn"][0].critic.store.flat, solo_agent.critic.store.flat)
    assert torch.equal(res["learn"][0].actor.store.flat, solo_agent.actor.store.flat)


def test_drq_pixel_keys_in_caller_order(dev):
    """pixel_keys=("wrist_2", "wrist_1"): the augmentation keys are split in the caller's order
    (drq_learner.py:131-137) while the encoder visits the cameras in sorted order with per-camera weights
    (two_d4pg_encoder.py:23-29).  One update vs the oracle, augmented images bit-exact."""
    from oracle import learner as ol
    from serl_amd.agents import DrQLearner
    keys = ("wrist_2", "wrist_1")
    H = W = 32; S = 2; B, U = 6, 1
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = DrQLearner.create(3, obs_space, act_space, pixel_keys=keys, num_qs=4, num_min_qs=2, critic_layer_norm=True)
    ostate, ocfg = agent_to_oracle(agent)
    assert tuple(ocfg["pixel_keys"]) == keys
    batch = make_pixel_batch(np.random.default_rng(4), B * U, keys, H, W, S, 19, 7, packed=True)
    collect = []
    ostate, oinfo = ol.drq_update(ostate, ocfg, batch, U, collect=collect)
    agent, info = agent.update(batch, U, pixel_keys=keys)
    ws = agent._ws(B)
    for side, okey in (("obs", "observations"), ("next", "next_observations")):
        for cam in keys:
            got = ws.tensor(f"aug_{side}_{cam}", (B * U, H, W, 3 * S), torch.uint8).cpu().numpy()
            assert np.array_equal(got.reshape(B * U, H, W, 3, S), collect[0][okey][cam]), f"aug {side} {cam}"
    assert_info_close(info, oinfo, rtol=1e-3)
    assert np.array_equal(agent.rng, ostate["rng"])
    # swapping the two cameras' images must change the result (per-camera weights): guards against a silent re-sort
    swapped = dict(batch, observations={"wrist_1": batch["observations"]["wrist_2"],
                                        "wrist_2": batch["observations"]["wrist_1"],
                                        "state": batch["observations"]["state"]})
    agent0 = DrQLearner.create(3, obs_space, act_space, pixel_keys=keys, num_qs=4, num_min_qs=2, critic_layer_norm=True)
    _, info_sw = agent0.update(swapped, U, pixel_keys=keys)
    assert abs(info_sw["critic_loss"] - info["critic_loss"]) > 1e-6


def test_update_does_not_touch_the_input_agent(dev):
    """Functional semantics (sac_learner.py:221-238 returns a new agent): update() leaves no attribute on `self`."""
    from serl_amd.agents import DrQLearner
    keys = ("pixels",)
    obs_space, act_space = pixel_spaces(keys, 32, 32, 1, 5, 3)
    agent = DrQLearner.create(0, obs_space, act_space, pixel_keys=keys)
    before = dict(vars(agent))
    snap = {n: getattr(agent, n).store.flat.clone() for n in ("actor", "critic", "target_critic")}
    agent.update(make_pixel_batch(np.random.default_rng(0), 8, keys, 32, 32, 1, 5, 3), 2, pixel_keys=keys)
    assert vars(agent).keys() == before.keys() and all(vars(agent)[k] is before[k] for k in before)
    for n, t in snap.items():
        assert torch.equal(getattr(agent, n).store.flat, t)


def test_insert_between_sample_and_update(dev):
    """The reference gathers pixels at sample time (memory_efficient_replay_buffer.py:155-160): inserts that land
    between `next(iterator)` and `update()` must not change the batch.  The device buffer's lazy windows are pinned
    when an insert is about to overwrite one of their rows."""
    from serl_amd.agents import DrQLearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    from helpers import rollout_transitions
    keys = ("wrist_1", "wrist_2")
    H = W = 32; S = 3
    cap = 120
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    bufs = [MemoryEfficientReplayBuffer(obs_space, act_space, cap, pixel_keys=keys, device=d) for d in (None, "cuda")]
    trs = list(rollout_transitions(np.random.default_rng(8), 400, keys, H, W, S, 19, 7, episode_len=17))
    for tr in trs[:300]:                                   # ring is full and has wrapped: every insert overwrites rows
        for b in bufs:
            b.insert(tr)
    for b in bufs:
        b.seed(5)
    args = dict(batch_size=64, pack_obs_and_next_obs=True)
    lazy = next(bufs[1].get_iterator(queue_size=1, sample_args=args))
    hostb = bufs[0].sample(**args)                         # snapshot, as the reference's sample() returns
    for tr in trs[300:340]:                                # 40 transitions: a third of the ring is rewritten
        for b in bufs:
            b.insert(tr)
    assert any(lazy["observations"][k].packed is not None for k in keys), "no sampled window was hit: weak test"
    agent = DrQLearner.create(0, obs_space, act_space, pixel_keys=keys, num_qs=2)
    a1, i1 = agent.update(lazy, 2, pixel_keys=keys)
    a2, i2 = agent.update(hostb, 2, pixel_keys=keys)
    assert i1 == i2
    assert torch.equal(a1.critic.store.flat, a2.critic.store.flat)


def test_config5_large_batch_update_properties(dev):
    """BASELINE config 5's per-GPU shape on one GPU: batch 4096, UTD 1 (2 x 128^2 cameras, stack 3, 10/2 critics).
    The CPU oracle would need minutes, so size-independent properties (the pattern of
    test_full_size_update_properties): augmented rows bit-exact vs the oracle's crop, a bit-identical repeat, the rng
    chain, finite outputs; plus linearity of the loss scale: critic_loss / q of the 4096-row step equal the row-weighted
    mean of the same quantities over its 16 contiguous 256-row slices evaluated at the same parameters."""
    from oracle import learner as ol
    from oracle import prng as oprng
    from serl_amd.agents import DrQLearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    from helpers import rollout_transitions
    keys = ("wrist_1", "wrist_2")
    H = W = 128; S = 3; B, U = 4096, 1
    n = B * U
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    bufs = [MemoryEfficientReplayBuffer(obs_space, act_space, 1500, pixel_keys=keys, device=d) for d in (None, "cuda")]
    rng = np.random.default_rng(6)
    for tr in rollout_transitions(rng, 1200, keys, H, W, S, 19, 7, episode_len=100):
        for b in bufs:
            b.insert(tr)
    args = dict(batch_size=n, pack_obs_and_next_obs=True, demo_batch_size=n // 2, demo_size=300)
    for b in bufs:
        b.seed(4)
    lazy = next(bufs[1].get_iterator(queue_size=1, sample_args=args))
    host = bufs[0].sample(**args)
    kw = dict(pixel_keys=keys, cnn_features=(32, 64, 128, 256), cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2),
              latent_dim=64, num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
    agent = DrQLearner.create(9, obs_space, act_space, **kw)
    a1, i1 = agent.update(lazy, U, pixel_keys=keys)
    rows = np.arange(0, n, 61)
    ws = a1._ws(B)
    r, key = oprng.split(agent.rng)
    for side, sl in (("obs", slice(0, S)), ("next", slice(1, S + 1))):
        kk = key
        for cam in keys:
            kcam, kk = oprng.split(kk)
            offs = ol.crop_offsets_fast(kcam, n)[rows]
            want = ol.random_crop_batch(host["observations"][cam][rows][..., sl], offs)
            got = ws.tensor(f"aug_{side}_{cam}", (n, H, W, 3 * S), torch.uint8)[torch.as_tensor(rows, device=dev)]
            assert np.array_equal(got.cpu().numpy().reshape(len(rows), H, W, 3, S), want), f"aug {side} {cam}"
        r, key = oprng.split(r)
    a2, i2 = agent.update(lazy, U, pixel_keys=keys)
    assert i1 == i2, (i1, i2)
    for name in ("actor", "critic", "target_critic"):
        assert torch.equal(getattr(a1, name).store.flat, getattr(a2, name).store.flat), name
    e = agent.rng
    e, _ = oprng.split(e)
    e, _ = oprng.split(e)
    for _ in range(4 * U + 2):
        _, e = oprng.split(e)
    assert np.array_equal(a1.rng, e)
    assert all(np.isfinite(v) for v in i1.values())
    assert bool(torch.isfinite(a1.critic.store.flat).all()) and bool(torch.isfinite(a1.actor.store.flat).all())


# Free-running drift bounds (relative, per logged scalar and call, against the float64 oracle over the same calls).
# north_star's bar is 1e-3.  Measured (gpurun_out/free_running_drift_*.json, copied to profiles/r02_drift_*.json):
#   state shape (C2 networks, 24 Adam steps): "highest" 6.7e-5, "bf16_3x" 2.9e-4; float32 oracle 5e-7  -> 1e-3 holds.
#   pixel shape (C3 networks, batch 8): after the first call (2 Adam steps) "highest" 3e-6, "bf16_3x" 5e-3 (actor_loss;
#   q 2.3e-4); after 12 Adam steps "highest" 3.1e-2 (actor_loss; q 4.2e-3), "bf16_3x" 5.1e-2 (q 2.5e-2), and the float32
#   ORACLE is itself 5.5e-4 away from the float64 oracle and growing ~10x per two calls.
# So for the pixel learner the free-running difference between ANY two fp32-level implementations grows chaotically
# (Adam's first, sign-like steps turn the sign of a noise-level gradient entry into a 2*lr parameter difference; a
# pre-activation that lands on the other side of a ReLU moves a gradient entry by a whole term), and 1e-3 "after N steps"
# can only be stated per call from a synchronised state -- which test_drq_learner_matches_oracle asserts at both
# precisions.  The pixel bounds below are 2-3x the measured drift: regression guards, not parity claims.
DRIFT_BOUND = {("c2_state", "highest"): 1e-3, ("c2_state", "bf16_3x"): 1e-3,
               ("c3_pixel", "highest"): 1e-1, ("c3_pixel", "bf16_3x"): 1.5e-1}
FIRST_CALL_BOUND = {("c3_pixel", "highest"): 1e-3, ("c3_pixel", "bf16_3x"): 1e-2}


@pytest.mark.parametrize("shape", ["c2_state", "c3_pixel"])
def test_free_running_drift(dev, shape, prec):
    """north_star's tolerance ("rel 1e-3 on losses and Q after N steps with a fixed RNG stream"), measured: N = 6
    consecutive update() calls with NO re-synchronisation of the oracle, at both matmul precisions, next to the drift of
    the float32 oracle against the float64 oracle over the same calls (two correct implementations of the same
    update).  Bounds: DRIFT_BOUND above.  The measured numbers go to gpurun_out/free_running_drift_<shape>_<prec>.json
    (quoted in DESIGN.md 5)."""
    import json
    import os
    from oracle import learner as ol
    rng = np.random.default_rng(12)
    N = 6
    if shape == "c2_state":        # BASELINE config 2 networks; UTD shortened to 4 for the CPU oracle (24 Adam steps)
        from serl_amd.agents import SACLearner
        obs_space, act_space = state_spaces(17, 6)
        agent = SACLearner.create(42, obs_space, act_space, num_qs=10, num_min_qs=2, critic_layer_norm=True,
                                  backup_entropy=False, init_temperature=0.1)
        B, U = 256, 4
        mk = lambda: make_state_batch(rng, B * U, 17, 6)
        oupd = lambda st, cfg, b, dt: ol.sac_update(st, cfg, b, U, dtype=dt)
        upd = lambda a, b: a.update(b, U)
    else:                          # BASELINE config 3 shapes (2 x 128^2, stack 3, E1 encoder, 10/2 critics), batch 8, UTD 2
        from serl_amd.agents import DrQLearner
        keys = ("wrist_1", "wrist_2")
        H = W = 128; S = 3; B, U = 8, 2
        obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
        agent = DrQLearner.create(42, obs_space, act_space, pixel_keys=keys, cnn_features=(32, 64, 128, 256),
                                  cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64, num_qs=10,
                                  num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
        mk = lambda: make_pixel_batch(rng, B * U, keys, H, W, S, 19, 7, packed=True)
        oupd = lambda st, cfg, b, dt: ol.drq_update(st, cfg, b, U, dtype=dt)
        upd = lambda a, b: a.update(b, U, pixel_keys=keys)
    o32, ocfg = agent_to_oracle(agent)
    o64, _ = agent_to_oracle(agent)
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-2)
    rows = []
    for it in range(N):
        batch = mk()
        o64, i64 = oupd(o64, ocfg, batch, torch.float64)
        o32, i32 = oupd(o32, ocfg, batch, torch.float32)
        agent, info = upd(agent, batch)
        rows.append({"call": it + 1, "adam_steps": (it + 1) * U,
                     "product_vs_f64": {k: rel(info[k], i64[k]) for k in i64},
                     "f32_oracle_vs_f64": {k: rel(i32[k], i64[k]) for k in i64}})
        assert np.array_equal(agent.rng, o64["rng"])
    worst_p = max(max(r["product_vs_f64"].values()) for r in rows)
    worst_o = max(max(r["f32_oracle_vs_f64"].values()) for r in rows)
    out = {"shape": shape, "precision": prec, "calls": N, "utd": U, "batch": B, "worst_product_vs_f64": worst_p,
           "worst_f32_oracle_vs_f64": worst_o, "rows": rows}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"free_running_drift_{shape}_{prec}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"free-running drift {shape} {prec}: product {worst_p:.3e}  fp32-oracle {worst_o:.3e}")
    for r in rows:
        bound = DRIFT_BOUND[(shape, prec)]
        if r["call"] == 1:
            bound = FIRST_CALL_BOUND.get((shape, prec), bound)
        for k, v in r["product_vs_f64"].items():
            assert v <= max(bound, 4.0 * r["f32_oracle_vs_f64"][k]), \
                f"call {r['call']} {k}: product drift {v:.3e} > {bound:.1e} (fp32 oracle {r['f32_oracle_vs_f64'][k]:.3e})"


@pytest.mark.parametrize("utd", [1, 2])
def test_drq_mobilenet_encoder_matches_oracle(dev, utd, prec):
    """BASELINE config 4's learner: DrQ-RLPD with encoder="mobilenet" (TwoMobileNetEncoder: ImageNet normalisation,
    frozen MobileNetV3-Large backbone, trainable SpatialLearnedEmbeddings, Dropout(0.1) in the training=True applies of
    the critic and target critic; two_mobilenet_encoder.py:19-34, sac_learner.py:177-206,129-134).  Synthetic backbone
    weights (pretrained ones are not available offline: backbone parity unpinned), dropout mask stream build-defined
    (oracle/learner.py encoder_dropout).  Two update() calls vs the oracle, augmented images bit-exact."""
    from oracle import learner as ol
    from serl_amd.agents import DrQLearner
    keys = ("wrist_1", "wrist_2")
    H = W = 64; S = 1; B, U = 4, utd
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = DrQLearner.create(11, obs_space, act_space, pixel_keys=keys, encoder="mobilenet", latent_dim=64, num_qs=10,
                              num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
    assert agent._cfg.enc_dim == 2 * 960 * 8 and agent._cfg.sle_shape == (2, 2, 960)
    assert set(agent.critic.params["image_encoder"]) == {"SpatialLearnedEmbeddings_0", "SpatialLearnedEmbeddings_1"}
    rng = np.random.default_rng(7)
    info_tol = 1e-3 if (prec == "highest" or U == 1) else 5e-3
    for it in range(2):
        ostate, ocfg = agent_to_oracle(agent)          # per-call parity from the product's own state
        batch = make_pixel_batch(rng, B * U, keys, H, W, S, 19, 7, packed=True)
        collect, c64 = [], []
        if U == 1:
            ol.drq_update(ostate, ocfg, batch, U, dtype=torch.float64, collect=c64)
        ostate, oinfo = ol.drq_update(ostate, ocfg, batch, U, collect=collect)
        agent, info = agent.update(batch, U, pixel_keys=keys)
        ws = agent._ws(B)
        for side, okey in (("obs", "observations"), ("next", "next_observations")):
            for cam in keys:
                got = ws.tensor(f"aug_{side}_{cam}", (B * U, H, W, 3 * S), torch.uint8).cpu().numpy()
                assert np.array_equal(got.reshape(B * U, H, W, 3, S), collect[0][okey][cam]), f"aug {side} {cam}"
        assert_info_close(info, oinfo, rtol=info_tol)
        assert np.array_equal(agent.rng, ostate["rng"])
        if U == 1:
            _check_grads(agent, "critic_grads", c64[-2]["grads"], None, prec, 3e-4, noise_ref=collect[-2]["grads"])
            _check_grads(agent, "actor_grads", c64[-1]["grads"], None, prec, 5e-4, noise_ref=collect[-1]["grads"])
    _check_params(agent, ostate, 3e-4, U)
    ob = {k: rng.integers(0, 256, size=(H, W, 3, S), dtype=np.uint8) for k in keys}
    ob["state"] = rng.standard_normal(19).astype(np.float32)
    osync, ocfg = agent_to_oracle(agent)
    a, agent2 = agent.sample_actions(ob)
    oa, osync2 = ol.sample_actions(osync, ocfg, ob)
    assert np.allclose(a, oa, atol=5e-4) and np.array_equal(agent2.rng, osync2["rng"])
    assert np.allclose(agent.eval_actions(ob), ol.eval_actions(osync, ocfg, ob), atol=5e-4)


@pytest.mark.parametrize("kind", ["state_ln", "state_plain", "pixel"])
def test_fused_ensemble_forward_bit_identical(dev, kind, monkeypatch):
    """csrc/ensemble.hip (the critic ensemble's forward in one launch) keeps the products, the k order and the LayerNorm /
    head reductions of the four launches it replaces: two agents created from the same seed, one with the fused kernel and
    one with SERL_AMD_ENS_FUSED=0, must hold identical bits after two update() calls (REDQ subsample with a negative member
    stride included), at a batch that is not a multiple of the kernel's 32-row blocks."""
    from serl_amd.agents import DrQLearner, SACLearner
    import serl_amd
    assert serl_amd.get_matmul_precision() == "bf16_3x"
    rng = np.random.default_rng(3)
    if kind == "pixel":
        keys = ("wrist_1", "wrist_2")
        obs_space, act_space = pixel_spaces(keys, 32, 32, 1, 19, 7)
        mk = lambda: DrQLearner.create(5, obs_space, act_space, pixel_keys=keys, num_qs=10, num_min_qs=2,
                                       critic_layer_norm=True, backup_entropy=False)
        batches = [make_pixel_batch(rng, 2 * 20, keys, 32, 32, 1, 19, 7, packed=True) for _ in range(2)]
        upd = lambda a, b: a.update(b, 2, pixel_keys=keys)
    else:
        obs_space, act_space = state_spaces(17, 6)
        ln = kind == "state_ln"
        mk = lambda: SACLearner.create(5, obs_space, act_space, num_qs=10 if ln else 2, num_min_qs=2 if ln else None,
                                       critic_layer_norm=ln)
        batches = [make_state_batch(rng, 3 * 50, 17, 6) for _ in range(2)]
        upd = lambda a, b: a.update(b, 3)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SERL_AMD_ENS_FUSED", fused)        # read when the native handle is created
        agent = mk()
        infos = []
        for b in batches:
            agent, info = upd(agent, b)
            infos.append(dict(info))
        out[fused] = (agent, infos)
    for name in ("actor", "critic", "target_critic"):
        assert torch.equal(getattr(out["1"][0], name).store.flat, getattr(out["0"][0], name).store.flat), name
    assert out["1"][1] == out["0"][1]


def test_full_size_patch_vs_im2col_update(dev):
    """The headline shapes (2 x 128^2 cameras, stack 3, E1 encoder, batch 256, 10/2 critics) through update() with the
    LDS-patch kernels (default) and with the im2col kernels they replace (serl_convs_set_patch_kernels(0)), from the same
    agent and batch: at this size every persistent patch block walks many items.  The patch forward / data-gradient
    kernels are bit-identical to im2col and the weight gradients of conv1 (fused into conv2's data gradient), conv2 and conv3
    only sum their pixels in another order, so after ONE critic step every gradient leaf except the first three conv layers'
    kernel / bias must hold the same bits and those must agree to the weight gradient's 2e-5 summation tolerance; after the one-step call the logged scalars agree to 1e-5,
    after a two-step call to 2e-4 (Adam's sign-like first step may move a weight whose gradient is summation noise by 2 lr: parameters
    are bounded like everywhere else in this file).
    The last layer's forward kernel (csrc/pconv4.h) orders its K loop by input-pixel parity class, so its outputs differ
    from the im2col kernel's in the last bit of the 16-bit S-format mantissa and everything behind it moves a little: the
    bit-level comparison above runs with that layer on the im2col kernel in both arms (mode 8 against mode 0), and the
    default (mode 1) is compared with mode 8 at 2e-3 of each gradient leaf's largest entry and 1e-4 on the logged scalars."""
    from serl_amd import ops
    from serl_amd.agents import DrQLearner
    keys = ("wrist_1", "wrist_2")
    H = W = 128; S = 3; B = 256
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = DrQLearner.create(21, obs_space, act_space, pixel_keys=keys, cnn_features=(32, 64, 128, 256),
                              cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64, num_qs=10,
                              num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
    rng = np.random.default_rng(2)
    batch2 = make_pixel_batch(rng, B * 2, keys, H, W, S, 19, 7, packed=True)
    batch1 = {k: ({kk: vv[:B] for kk, vv in v.items()} if isinstance(v, dict) else v[:B]) for k, v in batch2.items()}
    res = {}
    try:
        for mode in (8, 0, 1):
            ops.convs_set_patch_kernels(mode)
            a1, i1 = agent.update(batch1, 1, pixel_keys=keys)
            i1 = dict(i1)
            torch.cuda.synchronize()
            grads = {n: agent._shared[n].numpy_tree() for n in ("critic_grads", "actor_grads")}
            a2, i2 = agent.update(batch2, 2, pixel_keys=keys)
            res[mode] = (grads, i1, dict(i2), {n: getattr(a2, n).store.numpy_tree() for n in ("actor", "critic", "target_critic")})
    finally:
        ops.convs_set_patch_kernels(1)
    from helpers import _leaves
    gd = dict(_leaves(res[1][0]["critic_grads"]))
    res[1], res_default = res[8], res[1]
    g1, g0 = dict(_leaves(res[1][0]["critic_grads"])), dict(_leaves(res[0][0]["critic_grads"]))
    assert set(g1) == set(g0) == set(gd)
    for path in g1:      # default kernels (incl. the last layer's parity-class forward) against mode 8
        assert np.isfinite(gd[path]).all()
        assert np.abs(gd[path] - g1[path]).max() <= 2e-3 * max(float(np.abs(g1[path]).max()), 1e-12), f"default vs mode 8: {'/'.join(path)}"
    for k in res[1][1]:
        a, b = float(res_default[1][k]), float(res[1][1][k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-2), f"default vs mode 8: info {k}: {a} vs {b}"
    for n in ("actor", "critic", "target_critic"):
        assert_tree_close(res_default[3][n], res[1][3][n], atol=2.5 * 3e-4 * 2, mean_atol=1e-5, what=f"default vs mode 8: {n} params")
    nsummed = 0
    for path in g0:
        a, b = g1[path], g0[path]
        # first three conv layers of camera 1 / camera 2 (two_d4pg_encoder.py:30-38): the patch weight gradients of conv2 and
        # conv3 and conv1's weight gradient inside conv2's data gradient (serl_convs_dgrad_wgrad1) sum their pixels per
        # persistent block
        if path[-2] in ("Conv_0", "Conv_1", "Conv_2", "Conv_4", "Conv_5", "Conv_6"):
            nsummed += 1
            assert np.abs(a - b).max() <= 2e-5 * max(float(np.abs(b).max()), 1e-12), f"{'/'.join(path)}"
        else:
            assert np.array_equal(a, b), f"gradient leaf {'/'.join(path)} differs between patch and im2col kernels"
    assert nsummed == 12
    # the actor step runs on the critic AFTER its Adam step (sac_learner.py:235), i.e. behind conv2's re-ordered sums
    assert_tree_close(res[1][0]["actor_grads"], res[0][0]["actor_grads"], rtol=5e-4, what="actor grads")
    for k in res[0][1]:
        for which in (1, 2):
            a, b = float(res[1][which][k]), float(res[0][which][k])
            # call 1: one critic step from identical parameters; call 2: behind Adam's first, sign-like steps, which
            # turn conv2's re-ordered sums into +-lr differences of single weights (measured 4e-5 on actor_loss)
            tol = 1e-5 if which == 1 else 2e-4
            assert abs(a - b) <= tol * max(abs(b), 1e-2), f"info {k} (call {which}): {a} vs {b}"
    for n in ("actor", "critic", "target_critic"):
        assert_tree_close(res[1][3][n], res[0][3][n], atol=2.5 * 3e-4 * 2, mean_atol=2e-6, what=f"{n} params")


def test_config2_utd20_two_buffers_matches_oracle(dev, prec):
    """BASELINE config 2 at its real UTD: state RLPD (10/2 LayerNorm critics, no entropy backup), batch 256, UTD 20, every
    256-row minibatch half online / half demo rows drawn from two `ReplayBuffer`s (serl/data/replay_buffer.py:44-79,
    dataset.py:81-104), against the oracle fed by the oracle's own ring buffers with the same seeds
    (sac_learner.py:221-238).  The sampled rows must be identical; logged scalars rel 1e-3 ("highest") / 5e-3
    ("bf16_3x", 20 Adam steps per call from a re-synchronised state); rng chains equal."""
    from oracle import learner as ol
    from oracle import replay as orp
    from serl_amd.agents import SACLearner
    from serl_amd.data import ReplayBuffer
    obs_space, act_space = state_spaces(17, 6)
    B, U = 256, 20
    half = B // 2
    rng = np.random.default_rng(9)
    bufs = {"online": ReplayBuffer(obs_space, act_space, 3000), "demo": ReplayBuffer(obs_space, act_space, 1000)}
    obufs = {k: orp.new_ring(((17,), np.float32), ((6,), np.float32), cap) for k, cap in (("online", 3000), ("demo", 1000))}
    for name, n in (("online", 3500), ("demo", 700)):              # the online ring has wrapped, the demo one is part-filled
        for _ in range(n):
            tr = dict(observations=rng.standard_normal(17).astype(np.float32),
                      next_observations=rng.standard_normal(17).astype(np.float32),
                      actions=rng.uniform(-1, 1, 6).astype(np.float32), rewards=np.float32(rng.standard_normal()),
                      masks=np.float32(rng.random() < 0.99), dones=False)
            bufs[name].insert(tr)
            orp.ring_insert(obufs[name], tr)
    orngs = {}
    for i, k in enumerate(bufs):
        bufs[k].seed(100 + i)
        orngs[k] = orp.make_rng(100 + i)

    def mix(on, de):        # per minibatch: 128 online rows then 128 demo rows
        out = {}
        for k in ("observations", "next_observations", "actions", "rewards", "masks", "dones"):
            a, b = np.asarray(on[k]), np.asarray(de[k])
            out[k] = np.concatenate([np.concatenate([a[i * half:(i + 1) * half], b[i * half:(i + 1) * half]]) for i in range(U)])
        return out
    agent = SACLearner.create(42, obs_space, act_space, num_qs=10, num_min_qs=2, critic_layer_norm=True,
                              backup_entropy=False, init_temperature=0.1)
    ostate, ocfg = agent_to_oracle(agent)
    for it in range(2):
        batch = mix(bufs["online"].sample(half * U), bufs["demo"].sample(half * U))
        obatch = mix(orp.ring_sample(obufs["online"], orngs["online"], half * U)[0],
                     orp.ring_sample(obufs["demo"], orngs["demo"], half * U)[0])
        for k in batch:
            assert np.array_equal(batch[k], obatch[k]), f"sampled {k} differ from the oracle's buffers"
        if prec != "highest":
            ostate, _ = agent_to_oracle(agent)
        ostate, oinfo = ol.sac_update(ostate, ocfg, obatch, U)
        agent, info = agent.update(batch, U)
        assert_info_close(info, oinfo, rtol=1e-3 if prec == "highest" else 5e-3)
        assert np.array_equal(agent.rng, ostate["rng"])
    _check_params(agent, ostate, 3e-4, 2 * U)


@pytest.mark.parametrize("kind", ["state_ln", "state_plain", "pixel"])
def test_fused_ensemble_backward_matches_unfused(dev, kind, monkeypatch):
    """csrc/ensemble.hip ens_bwd_kernel + ens_colsum_kernel (the critic ensemble's backward in two launches) against the
    launch-by-launch path (SERL_AMD_ENS_BWD_FUSED=0: ln_act_bwd, GEMM, ln_act_bwd, four column-sum launches): same bf16_3x
    products, LayerNorm backward arithmetic and fixed-order sums, only the summation ORDER of the 256-term contraction and
    of the batch-column sums differs -> every critic / actor gradient leaf within 2e-6 (pixel: 2e-5) of its scale after one step, at a
    batch that is not a multiple of the kernel's 32-row blocks; the oracle parity tests of this file run the fused path."""
    from serl_amd.agents import DrQLearner, SACLearner
    rng = np.random.default_rng(5)
    if kind == "pixel":
        keys = ("wrist_1", "wrist_2")
        obs_space, act_space = pixel_spaces(keys, 32, 32, 1, 19, 7)
        mk = lambda: DrQLearner.create(5, obs_space, act_space, pixel_keys=keys, num_qs=10, num_min_qs=2,
                                       critic_layer_norm=True, backup_entropy=False)
        batch = make_pixel_batch(rng, 44, keys, 32, 32, 1, 19, 7, packed=True)
        upd = lambda a: a.update(batch, 1, pixel_keys=keys)
    else:
        obs_space, act_space = state_spaces(17, 6)
        ln = kind == "state_ln"
        mk = lambda: SACLearner.create(5, obs_space, act_space, num_qs=10 if ln else 2, num_min_qs=2 if ln else None,
                                       critic_layer_norm=ln)
        batch = make_state_batch(rng, 150, 17, 6)
        upd = lambda a: a.update(batch, 1)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SERL_AMD_ENS_BWD_FUSED", fused)        # read when the native handle is created
        agent = mk()
        new, info = upd(agent)
        info = dict(info)
        torch.cuda.synchronize()
        out[fused] = ({n: agent._shared[n].numpy_tree() for n in ("critic_grads", "actor_grads")}, info)
    # (pixel: the conv stack's backward amplifies the re-ordered sums; the actor step runs behind the critic's Adam step,
    # whose sign-like first update turns them into +-lr differences of single weights)
    # (state: the fused path sums the first kernel's gradient in exact fp32 instead of bf16_3x products: 2^-16 per product)
    assert_tree_close(out["1"][0]["critic_grads"], out["0"][0]["critic_grads"], rtol=2e-5, what="critic_grads")
    assert_tree_close(out["1"][0]["actor_grads"], out["0"][0]["actor_grads"], rtol=2e-4, what="actor_grads")
    for k in out["0"][1]:
        a, b = out["1"][1][k], out["0"][1][k]
        assert abs(a - b) <= 1e-5 * max(abs(b), 1e-2), (k, a, b)


@pytest.mark.parametrize("kind", ["state", "pixel"])
def test_fused_policy_forward_matches_unfused(dev, kind, monkeypatch):
    """csrc/ensemble.hip, policy epilogue (TanhNormal(MLP) forward in one launch: Dense + ReLU twice with the bf16_3x MFMA
    products of the GEMM launches, the two Dense heads as fp32 dot products, jax.random.normal at the global element index,
    tanh, log-prob) against the four launches it replaces (SERL_AMD_POL_FUSED=0): sampled and mode actions within 2e-5,
    the rng bit-identical, one update's logged scalars within 1e-4 -- at a batch that is not a multiple of 32 rows."""
    from serl_amd.agents import DrQLearner, SACLearner
    rng = np.random.default_rng(6)
    if kind == "pixel":
        keys = ("wrist_1", "wrist_2")
        obs_space, act_space = pixel_spaces(keys, 32, 32, 1, 19, 7)
        mk = lambda: DrQLearner.create(7, obs_space, act_space, pixel_keys=keys, num_qs=4, num_min_qs=2,
                                       critic_layer_norm=True, backup_entropy=True)
        batch = make_pixel_batch(rng, 2 * 21, keys, 32, 32, 1, 19, 7, packed=True)
        upd = lambda a: a.update(batch, 2, pixel_keys=keys)
        ob = {k: rng.integers(0, 256, size=(32, 32, 3, 1), dtype=np.uint8) for k in keys}
        ob["state"] = rng.standard_normal(19).astype(np.float32)
    else:
        obs_space, act_space = state_spaces(17, 6)
        mk = lambda: SACLearner.create(7, obs_space, act_space, num_qs=2, backup_entropy=True)
        batch = make_state_batch(rng, 3 * 50, 17, 6)
        upd = lambda a: a.update(batch, 3)
        ob = rng.standard_normal(17).astype(np.float32)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SERL_AMD_POL_FUSED", fused)        # read when the native handle is created
        agent = mk()
        a, agent2 = agent.sample_actions(ob)
        m = agent.eval_actions(ob)
        new, info = upd(agent)
        out[fused] = (a, m, agent2.rng, dict(info), new.rng)
    assert np.abs(out["1"][0] - out["0"][0]).max() <= 2e-5 and np.abs(out["1"][1] - out["0"][1]).max() <= 2e-5
    assert np.array_equal(out["1"][2], out["0"][2]) and np.array_equal(out["1"][4], out["0"][4])
    for k in out["0"][3]:
        a, b = out["1"][3][k], out["0"][3][k]
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-2), (k, a, b)
