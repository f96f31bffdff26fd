"""Behaviour-cloning learners (serl/agents/bc/pixel_bc_learner.py, pixel_hybrid_bc_learner.py) on the GPU vs the CPU
restatement oracle/bc.py: augmented observations bit-exact, nll / bce rel 1e-3, gradients (through the actor's image
encoder) against the float64 oracle, rng chain, acting.  Parity unpinned (oracle/bc.py header); the two reference lines
that cannot run in the snapshot are restated by their evident intent on both sides."""
import numpy as np
import pytest
import torch

from helpers import assert_info_close, assert_tree_close, make_pixel_batch, pixel_spaces

pytestmark = pytest.mark.gpu


def _ts(t):
    lay = t.store.layout
    return dict(params=t.store.numpy_tree(), mu=lay.unflatten(t.mu.cpu().numpy(), wrap=lambda a: a.copy()),
                nu=lay.unflatten(t.nu.cpu().numpy(), wrap=lambda a: a.copy()), count=t.step)


def _to_oracle(agent):
    cfg = agent._cfg
    state = dict(rng=np.array(agent.rng, dtype=np.uint32), actor=_ts(agent.actor))
    if hasattr(agent, "discrete_actor"):
        state["discrete_actor"] = _ts(agent.discrete_actor)
    ocfg = dict(actor_lr=agent.actor.lr, pixel_keys=cfg.pixel_keys, cnn_strides=cfg.cnn_strides)
    if cfg.encoder == "mobilenet":
        ocfg.update(encoder="mobilenet", backbone=agent._shared["backbone"].tree(), encoder_dropout=cfg.encoder_dropout)
    return state, ocfg


def _grads_close(got, ref64, ref32, prec, what):
    if prec == "highest":
        assert_tree_close(got, ref64, rtol=5e-4, what=what, noise_ref=ref32)
    else:
        assert_tree_close(got, ref64, rtol=5e-2, l2_rtol=2e-2, what=what)


@pytest.mark.parametrize("variant", ["bc_e1_packed", "hybrid_e1_packed", "hybrid_e0_unpacked", "hybrid_mobilenet"])
def test_bc_learner_matches_oracle(dev, variant, prec):
    from oracle import bc as ob
    from serl_amd.agents import PixelBCLearner, PixelHybridBCLearner
    hybrid = variant.startswith("hybrid")
    keys = ("wrist_1", "wrist_2")
    A = 7
    if variant.endswith("e1_packed"):
        H = W = 128; S = 3; B = 6; packed = True
        kw = dict(cnn_features=(32, 64, 128, 256), cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64)
    elif variant.endswith("mobilenet"):
        H = W = 64; S = 1; B = 4; packed = True
        kw = dict(encoder="mobilenet", latent_dim=64)
    else:                                   # create() defaults: 32-32-32-32, strides 2,1,1,1
        H = W = 32; S = 1; B = 5; packed = False
        kw = {}
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, A)
    cls = PixelHybridBCLearner if hybrid else PixelBCLearner
    agent = cls.create(3, obs_space, act_space, pixel_keys=keys, **kw)
    assert agent._cfg.action_dim == (A - 1 if hybrid else A)
    if hybrid:
        assert set(agent.discrete_actor.params) == {"image_encoder", "Dense_0", "LayerNorm_0", "Dense_1", "LayerNorm_1",
                                                    "OneDimOutput_0"}
    rng = np.random.default_rng(5)
    enc0 = None
    for it in range(2):
        batch = make_pixel_batch(rng, B, keys, H, W, S, 19, A, packed=packed)
        batch["actions"][0, 0] = 1.0            # hits the clip to 1 - 1e-5
        batch["actions"][1, 1] = -1.0
        ostate, ocfg = _to_oracle(agent)        # per-call parity from the product's own state
        c64, c32 = [], []
        ob.bc_update(ostate, ocfg, batch, hybrid, dtype=torch.float64, collect=c64)
        onew, oinfo = ob.bc_update(ostate, ocfg, batch, hybrid, collect=c32)
        if hybrid:
            enc0 = agent.discrete_actor.store.numpy_tree()["image_encoder"]
            new_agent, info = agent.update(batch, 1, pixel_keys=keys)
        else:
            new_agent, info = agent.update(batch, pixel_keys=keys)
        ws = new_agent._ws(B)
        for cam in keys:
            got = ws.tensor(f"aug_obs_{cam}", (B, H, W, 3 * S), torch.uint8).cpu().numpy()
            assert np.array_equal(got.reshape(B, H, W, 3, S), c32[0]["observations"][cam]), f"aug obs {cam}"
        assert_info_close(info, oinfo, rtol=1e-3)
        assert np.array_equal(new_agent.rng, onew["rng"]) and not np.array_equal(new_agent.rng, agent.rng)
        _grads_close(new_agent._shared["actor_grads"].numpy_tree(), c64[0]["actor_grads"], c32[0]["actor_grads"], prec,
                     "actor grads")
        g = new_agent._shared["actor_grads"].numpy_tree()["image_encoder"]
        assert any(np.abs(v["kernel"]).max() > 0 for v in g.values()), "the BC actor's encoder must be trained"
        lr = agent.actor.lr
        assert_tree_close(new_agent.actor.store.numpy_tree(), onew["actor"]["params"], atol=2.5 * lr, mean_atol=2e-5,
                          what="actor params")
        if hybrid:
            _grads_close(new_agent._shared["disc_grads"].numpy_tree(), c64[0]["disc_grads"], c32[0]["disc_grads"], prec,
                         "discrete actor grads")
            assert_tree_close(new_agent.discrete_actor.store.numpy_tree(), onew["discrete_actor"]["params"],
                              atol=2.5 * lr, mean_atol=2e-5, what="discrete actor params")
            enc1 = new_agent.discrete_actor.store.numpy_tree()["image_encoder"]      # stop_gradient: never moves
            for k in enc0:
                for leaf in enc0[k]:
                    assert np.array_equal(enc0[k][leaf], enc1[k][leaf]), f"discrete actor encoder {k}/{leaf} moved"
        # the input agent is untouched (immutable agents)
        assert agent.actor.step == it and new_agent.actor.step == it + 1
        agent = new_agent
    o = {k: rng.integers(0, 256, size=(H, W, 3, S), dtype=np.uint8) for k in keys}
    o["state"] = rng.standard_normal(19).astype(np.float32)
    ostate, ocfg = _to_oracle(agent)
    if hybrid:
        assert np.allclose(agent.compute_actions(o), ob.compute_actions(ostate, ocfg, o), atol=1e-3)
        ea = agent.eval_actions(o)
        assert ea.shape == (A,) and ea[-1] in (0.5, -0.5) and np.allclose(ea[:-1], ob.eval_actions(ostate, ocfg, o)[:-1], atol=5e-4)
    else:
        from oracle import learner as ol
        a, agent2 = agent.sample_actions(o)
        assert a.shape == (A,) and np.all(np.abs(a) <= 1) and not np.array_equal(agent2.rng, agent.rng)
        mean = ob.policy_params(ol.to_torch(ostate["actor"]["params"], torch.float32), ol._add_batch(o), ocfg,
                                torch.float32, None)[0]
        assert np.allclose(agent.eval_actions(o), torch.tanh(mean[0]).detach().numpy(), atol=5e-4)


def test_bc_losses_kernels(dev):
    """serl_tanh_gaussian_nll / serl_bce_logits against torch autograd in float64."""
    from serl_amd import _lib, ops
    rng = np.random.default_rng(0)
    B, A, ld = 37, 6, 7
    ms = torch.tensor(rng.standard_normal((B, 2 * A)).astype(np.float32) * 1.5, device=dev)
    ms[0, A] = 3.0; ms[1, A + 1] = -25.0                       # clipped log-stds: zero gradient
    act = torch.tensor(rng.uniform(-1, 1, (B, ld)).astype(np.float32), device=dev)
    act[2, 0] = 1.0; act[3, 1] = -1.0
    dms = torch.zeros((B, 2 * A), device=dev); info = torch.zeros(2, device=dev)
    st = ops.current_stream()
    _lib.call("serl_tanh_gaussian_nll", ms.data_ptr(), ms.data_ptr() + 4 * A, 2 * A, act.data_ptr(), ld, B, A, 1.0 / B,
              dms.data_ptr(), dms.data_ptr() + 4 * A, 2 * A, info.data_ptr(), st)
    m64 = ms.detach().cpu().double().requires_grad_(True)
    a64 = torch.clamp(act.cpu()[:, :A], -1 + 1e-5, 1 - 1e-5).double()
    x = torch.atanh(a64)
    ls = torch.clamp(m64[:, A:], -20, 2)
    z = (x - m64[:, :A]) * torch.exp(-ls)
    lp = (-0.5 * z * z - ls - 0.5 * np.log(2 * np.pi)).sum(-1) - (2 * (np.log(2.0) - x - torch.nn.functional.softplus(-2 * x))).sum(-1)
    nll = -lp.mean()
    nll.backward()
    assert abs(float(info[0]) / B - float(nll.detach())) <= 1e-5 * abs(float(nll.detach()))
    assert np.allclose(dms.cpu().numpy(), m64.grad.numpy(), rtol=2e-4, atol=1e-6)
    logits = torch.tensor(rng.standard_normal(B).astype(np.float32) * 3, device=dev)
    dl = torch.zeros(B, device=dev)
    _lib.call("serl_bce_logits", logits.data_ptr(), act.data_ptr() + 4 * A, ld, B, 1.0 / B, dl.data_ptr(),
              info.data_ptr() + 4, st)
    l64 = logits.cpu().double().requires_grad_(True)
    y = (act.cpu()[:, A] > 0).double()
    bce = torch.nn.functional.binary_cross_entropy_with_logits(l64, y)
    bce.backward()
    assert abs(float(info[1]) / B - float(bce.detach())) <= 1e-5 * abs(float(bce.detach()))
    assert np.allclose(dl.cpu().numpy(), l64.grad.numpy(), rtol=1e-4, atol=1e-7)
