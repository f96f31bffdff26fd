"""ORACLE (test infrastructure only) -- CPU restatement of SERL's learner update.

torch-CPU tensors + autograd are used as the gradient oracle; every function cites the
reference file:line it follows (paths relative to /root/reference).  Nothing under
`serl_amd/` imports this file.  dtype is float32 by default, float64 on request (tight
per-op checks).

PARITY UNPINNED: the reference ships no tests / golden vectors for this path and JAX,
Flax, Optax and TFP are not installable here, so this restatement is the
oracle of record; its random streams are pinned as far as oracle/prng.py says.

State layout (all leaves numpy arrays, names = flax auto-names):
    state = dict(rng=uint32[2],
                 actor=dict(params=..., mu=..., nu=..., count=int),
                 critic=dict(...), target_critic=dict(params=...),
                 temp=dict(params=dict(log_temp=()), mu=..., nu=..., count=int))
    cfg   = dict(discount, tau, target_entropy, num_qs, num_min_qs, backup_entropy,
                 actor_lr, critic_lr, temp_lr, critic_layer_norm,
                 pixel_keys=(), cnn_strides=(), ...)
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import prng

LN_EPS = 1e-6          # flax.linen.LayerNorm default
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # optax.adam defaults


# --------------------------------------------------------------------------- tree helpers
def tree_map(fn, *trees):
    t0 = trees[0]
    if isinstance(t0, dict):
        return {k: tree_map(fn, *[t[k] for t in trees]) for k in t0}
    return fn(*trees)


def tree_leaves(tree, prefix=()):
    if isinstance(tree, dict):
        out = []
        for k in tree:
            out += tree_leaves(tree[k], prefix + (k,))
        return out
    return [(prefix, tree)]


def to_torch(tree, dtype, requires_grad=False):
    def cv(a):
        t = torch.tensor(np.asarray(a), dtype=dtype)
        if requires_grad:
            t.requires_grad_(True)
        return t
    return tree_map(cv, tree)


def to_numpy(tree):
    return tree_map(lambda t: t.detach().cpu().numpy().astype(np.float32), tree)


# --------------------------------------------------------------------------- layers
def dense(p, x):
    """flax nn.Dense: y = x @ kernel + bias."""
    return x @ p["kernel"] + p["bias"]


def layer_norm(p, x):
    """flax 0.7 nn.LayerNorm: var = max(0, E[x^2]-E[x]^2), eps 1e-6, scale+bias."""
    mean = x.mean(-1, keepdim=True)
    var = torch.clamp((x * x).mean(-1, keepdim=True) - mean * mean, min=0.0)
    mul = torch.rsqrt(var + LN_EPS) * p["scale"]
    return (x - mean) * mul + p["bias"]


def critic_dropout(x, key, layer, member, rate, shard=None):
    """nn.Dropout(rate)(x, deterministic=False) behind hidden Dense `layer` of ensemble member position `member`
    (serl/networks/mlp.py:26-29; the vmapped Ensemble splits the dropout rng per member, ensemble.py:11-16).  flax derives
    the rng by folding in the module path and the vmap axis (not restatable without flax): BUILD-DEFINED as
    fold_in(fold_in(key, layer), member) -- parity unpinned for the mask stream.  shard as in encoder_dropout."""
    keep = np.float32(1.0 - rate)
    B, Fd = x.shape
    off, total = shard if shard is not None else (0, B)
    mask = prng.bernoulli(prng.fold_in(prng.fold_in(key, layer), member), keep, (total, Fd))[off:off + B]
    return torch.where(torch.as_tensor(mask), x / float(keep), torch.zeros_like(x))


def mlp(p, x, use_layer_norm, activate_final=True, dropout=None):
    """serl/networks/mlp.py:18-35 (ReLU).  dropout: None (dropout_rate None in every shipped config, or training=False)
    or (key, member position, rate, shard): Dense -> Dropout -> [LayerNorm] -> ReLU."""
    n = len([k for k in p if k.startswith("Dense_")])
    for i in range(n):
        x = dense(p[f"Dense_{i}"], x)
        if i + 1 < n or activate_final:
            if dropout is not None:
                x = critic_dropout(x, dropout[0], i, dropout[1], dropout[2], dropout[3])
            if use_layer_norm:
                x = layer_norm(p[f"LayerNorm_{i}"], x)
            x = torch.relu(x)
    return x


def ensemble_q(p, obs, actions, use_layer_norm, dropout=None):
    """serl/networks/ensemble.py:8-22 + state_action_value.py:11-19.
    Every leaf of p has a leading ensemble axis; returns (E, B).  dropout: None or (key, rate, shard)."""
    inp = torch.cat([obs, actions], -1)
    some = p["Dense_0"]["kernel"]
    qs = []
    for e in range(some.shape[0]):
        pe = tree_map(lambda a: a[e], p)
        h = mlp(pe["MLP_0"], inp, use_layer_norm, True,
                None if dropout is None else (dropout[0], e, dropout[1], dropout[2]))
        qs.append(dense(pe["Dense_0"], h).squeeze(-1))
    return torch.stack(qs, 0)


def sle(p, feat):
    """serl/networks/spatial.py:13-27: (B,H,W,C) x (H,W,C,F) -> (B, C*F)."""
    out = torch.einsum("bhwc,hwcf->bcf", feat, p["kernel"])
    return out.reshape(feat.shape[0], -1)


def conv_relu(p, x, stride):
    """flax nn.Conv (NHWC, HWIO, VALID, bias) + ReLU (two_d4pg_encoder.py:30-38)."""
    w = p["kernel"].permute(3, 2, 0, 1)              # HWIO -> OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), w, p["bias"], stride=stride)
    return torch.relu(y.permute(0, 2, 3, 1))


def two_d4pg_encoder(p, images, strides):
    """serl/networks/encoders/two_d4pg_encoder.py:15-50. `images`: dict key -> (B,H,W,C)
    float; cameras are visited in sorted-key order, each with its own Conv_/SLE_ params."""
    outs = []
    nl = len(strides)
    for ci, key in enumerate(sorted(images)):
        t = images[key]
        for li, s in enumerate(strides):
            t = conv_relu(p[f"Conv_{ci * nl + li}"], t, s)
        outs.append(sle(p[f"SpatialLearnedEmbeddings_{ci}"], t))
    return torch.cat(outs, -1)


def encoder_dropout(x, key, ci, rate, shard=None):
    """nn.Dropout(rate)(x, deterministic=False) behind camera ci's SpatialLearnedEmbeddings
    (two_mobilenet_encoder.py:34): where(bernoulli(rng, 1 - rate, x.shape), x / (1 - rate), 0).  flax derives `rng`
    from rngs={"dropout": key} by folding in the module path (not restatable without flax): BUILD-DEFINED here as
    fold_in(key, ci) -- parity unpinned for the mask stream.  With shard=(row_offset, total_rows) the rows are
    rows [row_offset, row_offset + B) of the (total_rows, F) draw (data parallel: keyed by the global row)."""
    keep = np.float32(1.0 - rate)
    B, Fd = x.shape
    off, total = shard if shard is not None else (0, B)
    mask = prng.bernoulli(prng.fold_in(key, ci), keep, (total, Fd))[off:off + B]
    return torch.where(torch.as_tensor(mask), x / float(keep), torch.zeros_like(x))


def two_mobilenet_encoder(p, images01, cfg, dtype, dropout_key=None, shard=None, rate=None):
    """serl/networks/encoders/two_mobilenet_encoder.py:15-45: per camera (sorted keys) ImageNet-normalise the /255 image,
    frozen backbone (no gradient), SpatialLearnedEmbeddings(..., 8), Dropout(rate) when training; concat."""
    from . import mobilenet as mb
    outs = []
    rate = cfg.get("encoder_dropout", 0.1) if rate is None else rate
    for ci, key in enumerate(sorted(images01)):
        with torch.no_grad():
            feat = mb.forward(cfg["backbone"], mb.normalize(images01[key].to(torch.float32)).numpy(), dtype=dtype)
        t = sle(p[f"SpatialLearnedEmbeddings_{ci}"], feat)
        if dropout_key is not None:
            t = encoder_dropout(t, dropout_key, ci, rate, shard)
        outs.append(t)
    return torch.cat(outs, -1)


def pixel_trunk(p, obs, cfg, dtype, dropout_key=None, shard=None):
    """serl/networks/pixel_multiplexer.py:25-45 up to the concat. obs: dict with 'state'
    and uint8 pixel leaves (B,H,W,3,S).  dropout_key: rngs={"dropout": key} of an apply call with training=True (only
    the MobileNet encoder has a Dropout layer)."""
    images = {}
    for k in obs:
        if k == "state":
            continue
        x = torch.as_tensor(np.asarray(obs[k]))
        x = x.reshape(*x.shape[:-2], -1).to(torch.float32) / 255.0     # :27
        images[k] = x.to(dtype)
    if cfg.get("encoder", "d4pg") == "mobilenet":
        return two_mobilenet_encoder(p["image_encoder"], images, cfg, dtype, dropout_key, shard)
    x = two_d4pg_encoder(p["image_encoder"], images, cfg["cnn_strides"])
    return x


def pixel_latent(p, enc_out, state):
    x = torch.tanh(layer_norm(p["LayerNorm_0"], dense(p["Dense_0"], enc_out)))   # :34-36
    y = torch.tanh(layer_norm(p["LayerNorm_1"], dense(p["Dense_1"], state)))     # :39-43
    return torch.cat([x, y], -1)


def _is_pixel(cfg):
    return bool(cfg.get("pixel_keys"))


def actor_dist(p, obs, cfg, dtype):
    """(mean, log_std) of the tanh-Gaussian policy (tanh_normal.py:26-45)."""
    if _is_pixel(cfg):
        enc = pixel_trunk(p, obs, cfg, dtype).detach()          # stop_gradient, :31-33
        st = torch.as_tensor(np.asarray(obs["state"]), dtype=dtype)
        z = pixel_latent(p, enc, st)
        hp = p["Normal_0"]
    else:
        z = torch.as_tensor(np.asarray(obs), dtype=dtype)
        hp = p
    h = mlp(hp["MLP_0"], z, False, True)
    mean = dense(hp["OutputDenseMean"], h)
    log_std = torch.clamp(dense(hp["OutputDenseLogStd"], h), -20.0, 2.0)
    return mean, log_std


def critic_qs(p, obs, actions, cfg, dtype, dropout_key=None, shard=None):
    if _is_pixel(cfg):
        enc = pixel_trunk(p, obs, cfg, dtype, dropout_key, shard)
        st = torch.as_tensor(np.asarray(obs["state"]), dtype=dtype)
        z = pixel_latent(p, enc, st)
        ep = p["Ensemble_0"]["VmapStateActionValue_0"]
    else:
        z = torch.as_tensor(np.asarray(obs), dtype=dtype)
        ep = p["VmapStateActionValue_0"]
    rate = cfg.get("critic_dropout_rate") or 0.0                 # sac_learner.py:47,75-81: MLP(dropout_rate=...)
    drop = (dropout_key, rate, shard) if (rate > 0 and dropout_key is not None) else None
    return ensemble_q(ep, z, actions, cfg["critic_layer_norm"], drop)


# --------------------------------------------------------------------------- distribution
def tanh_normal_sample(mean, log_std, key, shard=None):
    """TFP MultivariateNormalDiag.sample -> Tanh bijector; eps = jax.random.normal(key, shape).
    shard=(row_offset, total_rows): this batch is rows [row_offset, row_offset+B) of a larger global batch
    whose noise is one (total_rows, A) draw (data-parallel restatement, not in the reference)."""
    if shard is None:
        eps = prng.normal(key, tuple(mean.shape))
    else:
        off, total = shard
        eps = prng.normal(key, (total,) + tuple(mean.shape[1:]))[off:off + mean.shape[0]]
    eps = torch.tensor(eps, dtype=mean.dtype)
    x = mean + torch.exp(log_std) * eps
    return torch.tanh(x), x


def tanh_normal_log_prob(mean, log_std, x):
    """log N(x; mean, sigma) summed over the event dim, minus the Tanh log-det-Jacobian
    2(log2 - x - softplus(-2x)) (TFP Tanh bijector), x = cached pre-tanh sample."""
    z = (x - mean) * torch.exp(-log_std)
    base = (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
    ldj = (2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))).sum(-1)
    return base - ldj


# --------------------------------------------------------------------------- optimiser
def adam_step(ts, grads, lr):
    """optax.adam(lr): scale_by_adam(b1=.9,b2=.999,eps=1e-8) then scale(-lr), float32."""
    count = ts["count"] + 1
    f32 = np.float32
    bc1 = f32(1) - f32(ADAM_B1) ** f32(count)
    bc2 = f32(1) - f32(ADAM_B2) ** f32(count)

    def upd(p, m, v, g):
        g = np.asarray(g, dtype=np.float32)
        m2 = (f32(ADAM_B1) * m + f32(1 - ADAM_B1) * g).astype(np.float32)
        v2 = (f32(ADAM_B2) * v + f32(1 - ADAM_B2) * g * g).astype(np.float32)
        u = (m2 / bc1) / (np.sqrt(v2 / bc2) + f32(ADAM_EPS))
        return (p - f32(lr) * u).astype(np.float32), m2, v2

    flat = tree_map(upd, ts["params"], ts["mu"], ts["nu"], grads)
    return dict(params=_pick(flat, 0), mu=_pick(flat, 1), nu=_pick(flat, 2), count=count)


def _pick(tree, i):
    if isinstance(tree, dict):
        return {k: _pick(v, i) for k, v in tree.items()}
    return tree[i]


def polyak(new, old, tau):
    """optax.incremental_update: tau*new + (1-tau)*old."""
    f32 = np.float32
    return tree_map(lambda n, o: (f32(tau) * n + f32(1 - tau) * o).astype(np.float32), new, old)


# --------------------------------------------------------------------------- RNG plumbing
def subsample_ensemble(key, params, num_sample, num_qs):
    """serl/networks/ensemble.py:25-37."""
    if num_sample is None:
        return params, np.arange(num_qs)
    indx = prng.choice_no_replace(key, num_qs, num_sample)
    if "Ensemble_0" in params:
        out = dict(params)
        out["Ensemble_0"] = tree_map(lambda a: a[indx], params["Ensemble_0"])
        return out, indx
    return tree_map(lambda a: a[indx], params), indx


def _slice_batch(batch, lo, hi):
    return tree_map(lambda a: a[lo:hi], batch)


# --------------------------------------------------------------------------- SAC update
def update_critic(state, cfg, batch, dtype=torch.float32, shard=None):
    """serl/agents/sac/sac_learner.py:160-219.  With shard=(row_offset, total_rows) the loss is this shard's
    share sum((q-t)^2)/(E*total_rows) of the global mean, so shard gradients add up to the global one."""
    rng = state["rng"]
    ap = to_torch(state["actor"]["params"], dtype)
    with torch.no_grad():
        mean, log_std = actor_dist(ap, batch["next_observations"], cfg, dtype)   # :162-164
        key, rng = prng.split(rng)
        next_a, next_x = tanh_normal_sample(mean, log_std, key, shard)           # :168-169
        key, rng = prng.split(rng)
        tparams, indx = subsample_ensemble(key, state["target_critic"]["params"],
                                           cfg["num_min_qs"], cfg["num_qs"])     # :172-175
        key, rng = prng.split(rng)                                               # dropout key of the target apply (:177-184)
        next_qs = critic_qs(to_torch(tparams, dtype), batch["next_observations"], next_a, cfg, dtype, key, shard)
        next_q = next_qs.min(0).values                                           # :185
        rew = torch.as_tensor(np.asarray(batch["rewards"]), dtype=dtype)
        msk = torch.as_tensor(np.asarray(batch["masks"]), dtype=dtype)
        target_q = rew + cfg["discount"] * msk * next_q                          # :187
        if cfg["backup_entropy"]:
            nlp = tanh_normal_log_prob(mean, log_std, next_x)
            alpha = math.exp(float(state["temp"]["params"]["log_temp"]))
            target_q = target_q - cfg["discount"] * msk * alpha * nlp            # :189-196
    key, rng = prng.split(rng)                                                   # :198 dropout key of the critic apply

    cp = to_torch(state["critic"]["params"], dtype, requires_grad=True)
    act = torch.as_tensor(np.asarray(batch["actions"]), dtype=dtype)
    qs = critic_qs(cp, batch["observations"], act, cfg, dtype, key, shard)
    loss = ((qs - target_q) ** 2).mean()                                          # :208
    if shard is not None:
        loss = ((qs - target_q) ** 2).sum() / (qs.shape[0] * shard[1])
    leaves = tree_leaves(cp)
    gl = torch.autograd.grad(loss, [t for _, t in leaves], allow_unused=True)
    grads = _rebuild(cp, [g if g is not None else torch.zeros_like(t)
                          for g, (_, t) in zip(gl, leaves)])
    grads_np = to_numpy(grads)
    critic = adam_step(state["critic"], grads_np, cfg["critic_lr"])               # :212
    target = dict(params=polyak(critic["params"], state["target_critic"]["params"], cfg["tau"]))
    new = dict(state)
    new.update(critic=critic, target_critic=target, rng=rng)
    info = dict(critic_loss=float(loss.detach()), q=float(qs.detach().mean()))
    aux = dict(grads=grads_np, target_q=target_q.numpy(), next_actions=next_a.numpy(),
               qs=qs.detach().numpy(), subsample=indx)
    return new, info, aux


def _rebuild(template, leaves_list):
    it = iter(leaves_list)

    def rec(t):
        if isinstance(t, dict):
            return {k: rec(v) for k, v in t.items()}
        return next(it)
    return rec(template)


def update_actor(state, cfg, batch, dtype=torch.float32):
    """serl/agents/sac/sac_learner.py:120-144."""
    key, rng = prng.split(state["rng"])
    key2, rng = prng.split(rng)
    ap = to_torch(state["actor"]["params"], dtype, requires_grad=True)
    cp = to_torch(state["critic"]["params"], dtype)
    mean, log_std = actor_dist(ap, batch["observations"], cfg, dtype)
    a, x = tanh_normal_sample(mean, log_std, key)
    logp = tanh_normal_log_prob(mean, log_std, x)
    qs = critic_qs(cp, batch["observations"], a, cfg, dtype, key2)              # :129-134 training=True, dropout=key2
    q = qs.mean(0)                                                               # :135
    alpha = math.exp(float(state["temp"]["params"]["log_temp"]))
    loss = (logp * alpha - q).mean()                                             # :136-138
    leaves = tree_leaves(ap)
    gl = torch.autograd.grad(loss, [t for _, t in leaves], allow_unused=True)
    grads = _rebuild(ap, [g if g is not None else torch.zeros_like(t)
                          for g, (_, t) in zip(gl, leaves)])
    grads_np = to_numpy(grads)
    actor = adam_step(state["actor"], grads_np, cfg["actor_lr"])
    new = dict(state)
    new.update(actor=actor, rng=rng)
    info = dict(actor_loss=float(loss.detach()), entropy=float(-logp.detach().mean()))
    return new, info, dict(grads=grads_np, actions=a.detach().numpy())


def update_temperature(state, cfg, entropy):
    """serl/agents/sac/sac_learner.py:146-158; temperature.py:9-14."""
    f32 = np.float32
    lt = f32(state["temp"]["params"]["log_temp"])
    temperature = f32(np.exp(lt))
    diff = f32(entropy) - f32(cfg["target_entropy"])
    temp_loss = temperature * diff
    g = dict(log_temp=np.asarray(temperature * diff, dtype=np.float32))          # d/dlog_temp
    temp = adam_step(state["temp"], g, cfg["temp_lr"])
    new = dict(state)
    new.update(temp=temp)
    return new, dict(temperature=float(temperature), temperature_loss=float(temp_loss))


def sac_update(state, cfg, batch, utd_ratio, dtype=torch.float32, collect=None):
    """serl/agents/sac/sac_learner.py:221-238."""
    n = len(np.asarray(batch["rewards"]))
    assert n % utd_ratio == 0
    b = n // utd_ratio
    new = state
    for i in range(utd_ratio):
        mini = _slice_batch(batch, b * i, b * (i + 1))
        new, critic_info, aux = update_critic(new, cfg, mini, dtype)
        if collect is not None:
            collect.append(aux)
    new, actor_info, aux_a = update_actor(new, cfg, mini, dtype)
    if collect is not None:
        collect.append(aux_a)
    new, temp_info = update_temperature(new, cfg, actor_info["entropy"])
    return new, {**actor_info, **critic_info, **temp_info}


# --------------------------------------------------------------------------- DDPG
def ddpg_update_actor(state, cfg, batch, dtype=torch.float32):
    """serl/agents/ddpg/ddpg_learner.py:117-144: loss = mean(-mean_members q(obs, a)), a ~ pi(.|obs)."""
    key, rng = prng.split(state["rng"])
    key2, rng = prng.split(rng)
    ap = to_torch(state["actor"]["params"], dtype, requires_grad=True)
    cp = to_torch(state["critic"]["params"], dtype)
    mean, log_std = actor_dist(ap, batch["observations"], cfg, dtype)
    a, _ = tanh_normal_sample(mean, log_std, key)
    q = critic_qs(cp, batch["observations"], a, cfg, dtype).mean(0)             # :137
    loss = (-q).mean()                                                          # :138
    leaves = tree_leaves(ap)
    gl = torch.autograd.grad(loss, [t for _, t in leaves], allow_unused=True)
    grads = _rebuild(ap, [g if g is not None else torch.zeros_like(t) for g, (_, t) in zip(gl, leaves)])
    grads_np = to_numpy(grads)
    new = dict(state)
    new.update(actor=adam_step(state["actor"], grads_np, cfg["actor_lr"]), rng=rng)
    return new, dict(actor_loss=float(loss.detach())), dict(grads=grads_np, actions=a.detach().numpy())


def ddpg_update(state, cfg, batch, utd_ratio, dtype=torch.float32, collect=None):
    """serl/agents/ddpg/ddpg_learner.py:202-225.  The critic step (:146-200) is sac_learner.update_critic with one
    critic, subsample_ensemble(key, params, 1, 1) and no entropy backup: cfg must say num_qs=1, num_min_qs=1,
    backup_entropy=False."""
    assert cfg["num_qs"] == 1 and cfg["num_min_qs"] == 1 and not cfg["backup_entropy"]
    n = len(np.asarray(batch["rewards"]))
    assert n % utd_ratio == 0                                                   # :216
    b = n // utd_ratio
    new = state
    for i in range(utd_ratio):
        mini = _slice_batch(batch, b * i, b * (i + 1))
        new, critic_info, aux = update_critic(new, cfg, mini, dtype)
        if collect is not None:
            collect.append(aux)
    new, actor_info, aux_a = ddpg_update_actor(new, cfg, mini, dtype)
    if collect is not None:
        collect.append(aux_a)
    return new, {**actor_info, **critic_info}


def ddpg_sample_actions(state, cfg, obs, dtype=torch.float32):
    """serl/agents/ddpg/ddpg_learner.py:91-115.  Both `split(self.rng)` calls (:99, :102) start from the SAME rng,
    so the exploration noise is drawn with the key of the policy sample."""
    key, rng = prng.split(state["rng"])
    ap = to_torch(state["actor"]["params"], dtype)
    with torch.no_grad():
        mean, log_std = actor_dist(ap, _add_batch(obs), cfg, dtype)
        a, _ = tanh_normal_sample(mean[0], log_std[0], key)
    a = a.numpy().astype(np.float32)
    key, rng = prng.split(state["rng"])                                         # :102 (self.rng again)
    a = a + prng.normal(key, a.shape).astype(a.dtype) * 0.05                    # :103
    new = dict(state)
    new["rng"] = rng
    return a, new


# --------------------------------------------------------------------------- DrQ wrapper
def unpack(batch):
    """serl/utils/commons.py:7-29."""
    obs = dict(batch["observations"])
    nxt = dict(batch["next_observations"])
    for k in list(obs.keys()):
        if k not in nxt:
            packed = obs[k]
            obs[k] = packed[..., :-1]
            nxt[k] = packed[..., 1:]
    out = dict(batch)
    out["observations"], out["next_observations"] = obs, nxt
    return out


def share_encoder(src_params, dst_params):
    """serl/utils/commons.py:32-47."""
    out = dict(dst_params)
    for k, v in src_params.items():
        if "encoder" in k:
            out[k] = v
    return out


def crop_offsets(key, n, padding=4):
    """keys = split(key, n); per sample randint(key_i, (2,), 0, 2*padding+1) -> (dy, dx)
    (serl/utils/augmentations.py:5-7, 44-47)."""
    keys = prng.split(key, n)
    offs = np.empty((n, 2), dtype=np.int32)
    for i in range(n):
        offs[i] = prng.randint(keys[i], (2,), 0, 2 * padding + 1)
    return offs


def crop_offsets_fast(key, n, padding=4):
    """Vectorised equivalent of crop_offsets (same bits; used at full batch sizes)."""
    keys = prng.split(key, n)                                   # (n,2)
    span = np.uint32(2 * padding + 1)
    # randint: k1,k2 = split(key_i); bits(k, (2,)) = threefry_2x32(k, [0,1])
    c = np.arange(4, dtype=np.uint32)
    y0, y1 = prng.threefry2x32(keys[:, 0:1], keys[:, 1:2], c[None, :2], c[None, 2:])
    k1 = np.stack([y0[:, 0], y0[:, 1]], 1)
    k2 = np.stack([y1[:, 0], y1[:, 1]], 1)

    def bits2(k):
        a, b = prng.threefry2x32(k[:, 0], k[:, 1], np.uint32(0), np.uint32(1))
        return np.stack([a, b], 1)
    hi, lo = bits2(k1), bits2(k2)
    mult = np.uint32(((1 << 16) % int(span)) ** 2 % int(span))
    with np.errstate(over="ignore"):
        off = ((hi % span) * mult + lo % span).astype(np.uint32) % span
    return off.astype(np.int32)


def random_crop_batch(imgs, offs, padding=4):
    """edge-pad H,W by `padding`, slice back at (dy,dx) (augmentations.py:8-11)."""
    imgs = np.asarray(imgs)
    out = np.empty_like(imgs)
    h, w = imgs.shape[1:3]
    pad = [(padding, padding), (padding, padding)] + [(0, 0)] * (imgs.ndim - 3)
    for i in range(imgs.shape[0]):
        p = np.pad(imgs[i], pad, mode="edge")
        dy, dx = int(offs[i, 0]), int(offs[i, 1])
        out[i] = p[dy:dy + h, dx:dx + w]
    return out


def data_augmentation(rng, observations, pixel_keys, fast=True):
    """serl/agents/drq/drq_learner.py:131-137 (no depth keys)."""
    obs = dict(observations)
    f = crop_offsets_fast if fast else crop_offsets
    for k in pixel_keys:
        key, rng = prng.split(rng)
        offs = f(key, len(obs[k]))
        obs[k] = random_crop_batch(obs[k], offs)
    return obs


def drq_update(state, cfg, batch, utd_ratio, dtype=torch.float32, collect=None):
    """serl/agents/drq/drq_learner.py:154-186."""
    pixel_keys = cfg["pixel_keys"]
    if pixel_keys[0] not in batch["next_observations"]:
        batch = unpack(batch)                                                     # :168-169
    new = dict(state)
    actor = dict(state["actor"])
    actor["params"] = share_encoder(state["critic"]["params"], actor["params"])   # :171-172
    new["actor"] = actor
    rng, key = prng.split(new["rng"])                                             # :174 (rng first!)
    obs = data_augmentation(key, batch["observations"], pixel_keys)
    rng, key = prng.split(rng)                                                    # :176
    nxt = data_augmentation(key, batch["next_observations"], pixel_keys)
    batch = dict(batch)
    batch["observations"], batch["next_observations"] = obs, nxt
    new["rng"] = rng                                                              # :185
    if collect is not None:
        collect.append(dict(observations=obs, next_observations=nxt))
    return sac_update(new, cfg, batch, utd_ratio, dtype, collect)


def pixel_ddpg_update(state, cfg, batch, utd_ratio, dtype=torch.float32, collect=None):
    """serl/agents/ddpg/pixel_ddpg_learner.py:148-181.  `key, rng = split(rng)` (key first, unlike DrQ), and the
    DDPG update runs on `new_agent`, whose rng is still the rng the call started with (:167,180)."""
    pixel_keys = cfg["pixel_keys"]
    if pixel_keys[0] not in batch["next_observations"]:
        batch = unpack(batch)                                                     # :163-164
    new = dict(state)
    actor = dict(state["actor"])
    actor["params"] = share_encoder(state["critic"]["params"], actor["params"])   # :166-167
    new["actor"] = actor
    key, rng = prng.split(new["rng"])                                             # :169
    obs = data_augmentation(key, batch["observations"], pixel_keys)
    key, rng = prng.split(rng)                                                    # :171
    nxt = data_augmentation(key, batch["next_observations"], pixel_keys)
    batch = dict(batch)
    batch["observations"], batch["next_observations"] = obs, nxt
    if collect is not None:
        collect.append(dict(observations=obs, next_observations=nxt))
    return ddpg_update(new, cfg, batch, utd_ratio, dtype, collect)                # :180 (new["rng"] untouched)


# --------------------------------------------------------------------------- acting
def sample_actions(state, cfg, obs, dtype=torch.float32):
    """serl/agents/agent.py:12-16,33-37 (single un-batched observation)."""
    key, rng = prng.split(state["rng"])
    ob = _add_batch(obs)
    with torch.no_grad():
        mean, log_std = actor_dist(to_torch(state["actor"]["params"], dtype), ob, cfg, dtype)
        a, _ = tanh_normal_sample(mean[0], log_std[0], key)
    new = dict(state)
    new["rng"] = rng
    return a.numpy().astype(np.float32), new


def eval_actions(state, cfg, obs, dtype=torch.float32):
    """serl/agents/agent.py:19-22,29-31: mode = tanh(mean)."""
    with torch.no_grad():
        mean, _ = actor_dist(to_torch(state["actor"]["params"], dtype), _add_batch(obs), cfg, dtype)
    return torch.tanh(mean[0]).numpy().astype(np.float32)


def _add_batch(obs):
    if isinstance(obs, dict):
        return {k: np.asarray(v)[None] for k, v in obs.items()}
    return np.asarray(obs)[None]
