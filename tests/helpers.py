"""Shared test helpers: synthetic batches, product-agent -> oracle-state export,
tree comparison."""
import numpy as np


def state_spaces(obs_dim, act_dim):
    from serl_amd.spaces import Box
    return Box(-np.inf, np.inf, (obs_dim,), np.float32), Box(-1.0, 1.0, (act_dim,), np.float32)


def pixel_spaces(keys, H, W, S, state_dim, act_dim):
    from serl_amd.spaces import Box, Dict
    sp = {k: Box(0, 255, (H, W, 3, S), np.uint8) for k in keys}
    sp["state"] = Box(-np.inf, np.inf, (state_dim,), np.float32)
    return Dict(sp), Box(-1.0, 1.0, (act_dim,), np.float32)


def make_state_batch(rng, n, obs_dim, act_dim):
    return dict(
        observations=rng.standard_normal((n, obs_dim)).astype(np.float32),
        next_observations=rng.standard_normal((n, obs_dim)).astype(np.float32),
        actions=rng.uniform(-1, 1, (n, act_dim)).astype(np.float32),
        rewards=rng.standard_normal(n).astype(np.float32),
        masks=(rng.random(n) < 0.99).astype(np.float32),
        dones=np.zeros(n, bool),
    )


def _frames(rng, n, H, W, T):
    """uint8 noise + a low-frequency pattern so that shifted crops are distinguishable."""
    yy, xx = np.mgrid[0:H, 0:W]
    base = ((yy * 3 + xx * 5) % 251).astype(np.int32)
    x = rng.integers(0, 64, size=(n, H, W, 3, T), dtype=np.int32) + base[None, :, :, None, None] // 2
    return np.clip(x, 0, 255).astype(np.uint8)


def make_pixel_batch(rng, n, keys, H, W, S, state_dim, act_dim, packed=True):
    obs, nxt = {}, {}
    for k in keys:
        win = _frames(rng, n, H, W, S + 1)
        if packed:
            obs[k] = win
        else:
            obs[k] = np.ascontiguousarray(win[..., :-1])
            nxt[k] = np.ascontiguousarray(win[..., 1:])
    obs["state"] = rng.standard_normal((n, state_dim)).astype(np.float32)
    nxt["state"] = rng.standard_normal((n, state_dim)).astype(np.float32)
    return dict(observations=obs, next_observations=nxt,
                actions=rng.uniform(-1, 1, (n, act_dim)).astype(np.float32),
                rewards=rng.standard_normal(n).astype(np.float32),
                masks=(rng.random(n) < 0.99).astype(np.float32), dones=np.zeros(n, bool))


def rollout_transitions(rng, n, keys, H, W, S, state_dim, act_dim, episode_len=100):
    """Transitions of a fake env with a FrameStack wrapper: obs = last S frames (stacked on the last axis)."""
    t = 0
    stack = None
    for i in range(n):
        if t == 0:
            f0 = {k: rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for k in keys}
            stack = {k: np.stack([f0[k]] * S, axis=-1) for k in keys}
        new = {k: rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for k in keys}
        nstack = {k: np.concatenate([stack[k][..., 1:], new[k][..., None]], axis=-1) for k in keys}
        done = (t + 1 == episode_len)
        obs = dict(stack, state=rng.standard_normal(state_dim).astype(np.float32))
        nxt = dict(nstack, state=rng.standard_normal(state_dim).astype(np.float32))
        yield dict(observations=obs, next_observations=nxt,
                   actions=rng.uniform(-1, 1, act_dim).astype(np.float32),
                   rewards=np.float32(rng.standard_normal()), masks=np.float32(0.0 if done else 1.0), dones=bool(done))
        stack = nstack
        t = 0 if done else t + 1


def _zeros_like_tree(t):
    if isinstance(t, dict):
        return {k: _zeros_like_tree(v) for k, v in t.items()}
    return np.zeros_like(t)


def agent_to_oracle(agent):
    """(oracle state, oracle cfg) holding copies of the product agent's parameters and hyper-parameters."""
    cfg = agent._cfg

    def ts(t):
        lay = t.store.layout
        p = t.store.numpy_tree()
        if t.mu is None:
            return dict(params=p)
        mu = lay.unflatten(t.mu.cpu().numpy(), wrap=lambda a: a.copy())
        nu = lay.unflatten(t.nu.cpu().numpy(), wrap=lambda a: a.copy())
        return dict(params=p, mu=mu, nu=nu, count=t.step)
    tb = agent.temp.buf.cpu().numpy()
    state = dict(rng=np.array(agent.rng, dtype=np.uint32), actor=ts(agent.actor), critic=ts(agent.critic),
                 target_critic=ts(agent.target_critic),
                 temp=dict(params=dict(log_temp=np.float32(tb[0])), mu=dict(log_temp=np.float32(tb[1])),
                           nu=dict(log_temp=np.float32(tb[2])), count=agent.temp.step))
    ocfg = dict(discount=agent.discount, tau=agent.tau, target_entropy=agent.target_entropy, num_qs=agent.num_qs,
                num_min_qs=agent.num_min_qs, backup_entropy=agent.backup_entropy, actor_lr=agent.actor.lr,
                critic_lr=agent.critic.lr, temp_lr=agent.temp.lr, critic_layer_norm=cfg.critic_layer_norm,
                pixel_keys=cfg.pixel_keys, cnn_strides=cfg.cnn_strides,
                critic_dropout_rate=getattr(agent, "critic_dropout_rate", None))
    if getattr(cfg, "encoder", "d4pg") == "mobilenet":
        ocfg.update(encoder="mobilenet", backbone=agent._shared["backbone"].tree(), encoder_dropout=cfg.encoder_dropout)
    return state, ocfg


def assert_info_close(info, ref, rtol):
    assert set(info) == set(ref), (sorted(info), sorted(ref))
    for k in ref:
        a, b = float(info[k]), float(ref[k])
        assert abs(a - b) <= rtol * max(abs(b), 1e-2), f"{k}: {a} vs {b}"


def _leaves(tree, prefix=()):
    if isinstance(tree, dict) or hasattr(tree, "items"):
        out = []
        for k, v in tree.items():
            out += _leaves(v, prefix + (k,))
        return out
    return [(prefix, np.asarray(tree))]


def assert_tree_close(got, ref, rtol=None, atol=None, mean_atol=None, what="", noise_ref=None, noise_mult=4.0,
                      l2_rtol=None):
    """l2_rtol: bound on ||got - ref||_2 / ||ref||_2 per leaf (robust to single ReLU-mask flips, which move one
    gradient entry by a whole term however small the perturbation that caused the flip).
    noise_ref: the same quantity from the float32 oracle when `ref` is the float64 oracle; the allowed
    error is then max(rtol*scale, noise_mult * |f32 oracle - f64 oracle|_max) per leaf."""
    g, r = dict(_leaves(got)), dict(_leaves(ref))
    nz = dict(_leaves(noise_ref)) if noise_ref is not None else None
    assert set(g) == set(r), f"{what}: leaf sets differ {set(g) ^ set(r)}"
    for path in r:
        a, b = g[path].astype(np.float64), r[path].astype(np.float64)
        assert a.shape == b.shape, f"{what} {path}: {a.shape} vs {b.shape}"
        d = np.abs(a - b)
        if rtol is not None:
            scale = max(float(np.abs(b).max()), 1e-12)
            tol = rtol * scale + 1e-9
            if nz is not None:
                tol = max(tol, noise_mult * float(np.abs(nz[path].astype(np.float64) - b).max()))
            assert d.max() <= tol, f"{what} {'/'.join(path)}: max diff {d.max():.3e} > {tol:.3e} (scale {scale:.3e})"
        if l2_rtol is not None:
            nrm = float(np.sqrt((b * b).sum()))
            e2 = float(np.sqrt((d * d).sum()))
            assert e2 <= l2_rtol * max(nrm, 1e-12) + 1e-9, f"{what} {'/'.join(path)}: L2 rel err {e2 / max(nrm, 1e-12):.3e} > {l2_rtol:.1e}"
        if atol is not None:
            assert d.max() <= atol, f"{what} {'/'.join(path)}: max diff {d.max():.3e} > {atol:.3e}"
        if mean_atol is not None:
            assert d.mean() <= mean_atol, f"{what} {'/'.join(path)}: mean diff {d.mean():.3e} > {mean_atol:.3e}"
