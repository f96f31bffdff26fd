"""ORACLE (test infrastructure only) -- CPU restatement of the reward networks of BASELINE config 4:

  BinaryClassifier / classify_reward   serl/networks/binary_classifier.py:5-17,
                                       serl/agents/drq/drq_classifier_reward_learner.py:191-195
  VICE encode / vice_reward / update   serl/agents/vice/vice_learner.py:231-314
  VICE update_classifier               serl/agents/vice/vice_learner.py:316-474 (torch double backward as the oracle of
                                       the gradient penalty)
on the frozen backbone of oracle/mobilenet.py (synthetic weights; backbone parity unpinned).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this package."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import learner as ol
from . import mobilenet as mb
from . import prng


def _feat(backbone, img01, dtype):
    with torch.no_grad():
        return mb.forward(backbone, mb.normalize(torch.as_tensor(np.asarray(img01), dtype=torch.float32)).numpy(), dtype=dtype)


def binary_classifier(params, images01, backbone, dtype=torch.float32):
    """sigmoid(BinaryClassifier(obs, training=False)): images01 = {camera: (n,H,W,3) float in [0,1]} -> (n, 1)."""
    p = ol.to_torch(params, dtype)
    outs = []
    for ci, cam in enumerate(sorted(images01)):
        outs.append(ol.sle(p["image_encoder"][f"SpatialLearnedEmbeddings_{ci}"], _feat(backbone, images01[cam], dtype)))
    x = torch.cat(outs, -1)                                                     # two_mobilenet_encoder.py:39
    x = ol.dense(p["Dense_0"], x)                                               # binary_classifier.py:12 (Dropout: eval)
    x = torch.relu(ol.layer_norm(p["LayerNorm_0"], x))                          # :14-15
    return torch.sigmoid(ol.dense(p["Dense_1"], x))                             # :16, learner :192


def vice_logits(p, feat, layer_norm, drop1=None, drop2=None):
    """EncodedEncoder(OneDimOutput(MLP((256,), leaky_relu))) on encoded features (n,h,w,c) -> (n,).
    drop1 / drop2: multiplicative Dropout masks (already divided by keep) behind the spatial embeddings / the MLP's Dense
    (encoded_encoder.py:33-34, mlp.py:26-29); None = deterministic."""
    s = ol.sle(p["SpatialLearnedEmbeddings_0"], feat)                           # encoded_encoder.py:33
    if drop1 is not None:
        s = s * drop1
    t = torch.tanh(ol.layer_norm(p["LayerNorm_0"], ol.dense(p["Dense_0"], s)))  # :45-47
    o = p["OneDimOutput_0"]
    v = ol.dense(o["MLP_0"]["Dense_0"], t)                                      # mlp.py:24
    if drop2 is not None:
        v = v * drop2
    if layer_norm:
        v = ol.layer_norm(o["MLP_0"]["LayerNorm_0"], v)                         # :30-31
    r = torch.nn.functional.leaky_relu(v, 0.01)                                 # :33, vice_learner.py:156
    return ol.dense(o["Dense_0"], r).squeeze(-1)                                # one_d_output.py:19-20


def encode(cfg, pixels_u8, dtype=torch.float32):
    """vice_learner.py:231-244: reshape (n,H,W,3,1) -> (n,H,W,3), /255, normalise, frozen backbone."""
    x = np.asarray(pixels_u8)
    x = x.reshape(*x.shape[:-2], -1).astype(np.float32) / np.float32(255.0)
    return _feat(cfg["backbone"], x, dtype)


def vice_reward(state, cfg, observations, pixel_keys, dtype=torch.float32):
    """vice_learner.py:246-275 -> (state with the advanced rng, {camera: (n,) sigmoid, 'mean': (n,)})."""
    rng = state["rng"]
    rews = {}
    for k in pixel_keys:
        key, rng = prng.split(rng)
        p = ol.to_torch(state["vice"][k]["params"], dtype)
        with torch.no_grad():
            rews[k] = torch.sigmoid(vice_logits(p, encode(cfg, observations[k], dtype), cfg["critic_layer_norm"]))
    rews["mean"] = torch.stack([rews[k] for k in pixel_keys], 0).mean(0)
    new = dict(state)
    new["rng"] = rng
    return new, rews


def vice_update(state, cfg, batch, utd_ratio, dtype=torch.float32, collect=None):
    """vice_learner.py:277-314."""
    pixel_keys = cfg["pixel_keys"]
    if pixel_keys[0] not in batch["next_observations"]:
        batch = ol.unpack(batch)
    new = dict(state)
    actor = dict(state["actor"])
    actor["params"] = ol.share_encoder(state["critic"]["params"], actor["params"])
    new["actor"] = actor
    key, rng = prng.split(new["rng"])                                           # :295 (key first; rng never written back)
    obs = ol.data_augmentation(key, batch["observations"], pixel_keys)
    key, rng = prng.split(rng)
    nxt = ol.data_augmentation(key, batch["next_observations"], pixel_keys)
    new, rews = vice_reward(new, cfg, obs, pixel_keys, dtype)                    # :300 on the AUGMENTED observations
    m = rews["mean"]
    vice_rews = torch.log(m / (1 - m)).to(torch.float32).numpy()                 # :301
    batch = dict(batch)
    batch["observations"], batch["next_observations"], batch["rewards"] = obs, nxt, vice_rews
    if collect is not None:
        collect.append(dict(observations=obs, next_observations=nxt, rewards=vice_rews))
    out, info = ol.sac_update(new, cfg, batch, utd_ratio, dtype, collect)
    info = dict(info, vice_rews=float(vice_rews.mean()))
    return out, info


def classifier_reward_update(state, cfg, batch, utd_ratio, dtype=torch.float32, collect=None):
    """drq_classifier_reward_learner.py:197-219: DrQ's update with `key, rng = split(rng)` augmentation keys and the
    split rng dropped -- the same key plumbing as PixelDDPGLearner.update, followed by the full SAC update."""
    pixel_keys = cfg["pixel_keys"]
    if pixel_keys[0] not in batch["next_observations"]:
        batch = ol.unpack(batch)
    new = dict(state)
    actor = dict(state["actor"])
    actor["params"] = ol.share_encoder(state["critic"]["params"], actor["params"])
    new["actor"] = actor
    key, rng = prng.split(new["rng"])
    obs = ol.data_augmentation(key, batch["observations"], pixel_keys)
    key, rng = prng.split(rng)
    nxt = ol.data_augmentation(key, batch["next_observations"], pixel_keys)
    batch = dict(batch)
    batch["observations"], batch["next_observations"] = obs, nxt
    if collect is not None:
        collect.append(dict(observations=obs, next_observations=nxt))
    return ol.sac_update(new, cfg, batch, utd_ratio, dtype, collect)


# ------------------------------------------------------------------------------------------ VICE classifier training
def _dropout_mask(key, which, rate, shape, dtype):
    """Dropout mask / keep of the classifier's two Dropout layers (which = 0: behind the spatial embeddings,
    encoded_encoder.py:34; 1: inside the MLP, mlp.py:26-29).  BUILD-DEFINED stream: bernoulli(fold_in(key, which), keep,
    shape) -- flax folds the module path into rngs["dropout"], which is not restatable offline (parity unpinned)."""
    if not rate:
        return None
    keep = np.float32(1.0 - rate)
    m = prng.bernoulli(prng.fold_in(key, which), keep, shape)
    return torch.as_tensor(m.astype(np.float32) / keep, dtype=dtype)


def update_classifier(state, cfg, batch, pixel_keys, goal_pool, dtype=torch.float32, collect=None):
    """vice_learner.py:316-474.  state["vice"][camera] = dict(params, mu, nu, count); cfg: vice_lr, vice_dropout_rate,
    critic_layer_norm, backbone.  Deviations, both forced and documented in DESIGN.md: lam = jax.random.beta(key_0, 1, 1)
    (a gamma-rejection sampler) is drawn as uniform(key_0) -- Beta(1,1) IS the uniform distribution, the bit stream differs;
    Dropout mask streams are build-defined (_dropout_mask).  Python closes over the variable `key`, not its value, so both
    the mixup and the gradient-penalty apply use the SECOND `key, rng = split(rng, 2)` (:438,451) as dropout key; under
    vmap(grad(...)) with a closed-over rng every sample of the gradient-penalty batch gets the same (1, F) mask."""
    rng = state["rng"]
    if pixel_keys[0] not in batch["next_observations"]:
        batch = ol.unpack(batch)
    observations = batch["observations"]
    key, rng = prng.split(rng)
    aug_observations = ol.data_augmentation(key, observations, pixel_keys)        # :368-370
    rate = cfg.get("vice_dropout_rate")
    ln = cfg["critic_layer_norm"]
    new_vice, info = {}, {}
    for p_key in pixel_keys:
        pixels = np.asarray(observations[p_key])
        B = pixels.shape[0]
        key, rng = prng.split(rng)                                                # :377 (unused)
        key, rng = prng.split(rng)                                                # _sample :335-339
        indx = prng.randint(key, (B,), 0, len(goal_pool[p_key]))
        goal = np.asarray(goal_pool[p_key])[indx]
        goal_obs = dict(observations)
        goal_obs[p_key] = goal
        key, rng = prng.split(rng)                                                # :387
        aug_goal = ol.data_augmentation(key, goal_obs, pixel_keys)[p_key]
        all_pixels = np.concatenate([goal, aug_goal, pixels, np.asarray(aug_observations[p_key])], 0)   # :396-398
        y = np.concatenate([np.ones(2 * B), np.zeros(2 * B)]).astype(np.float32) * np.float32(0.8) + np.float32(0.1)   # :401-407
        enc = encode(cfg, all_pixels, dtype)                                      # :410
        k = prng.split(rng, 3)                                                    # :413
        key_0, key_1, rng = k[0], k[1], k[2]
        lam = float(prng.uniform(key_0, ()))                                      # beta(key_0, 1, 1): see docstring
        perm = prng.permutation(key_1, 4 * B)
        mix = lam * enc + (1 - lam) * enc[torch.as_tensor(perm.astype(np.int64))]   # :361
        y_a, y_b = torch.as_tensor(y, dtype=dtype), torch.as_tensor(y[perm], dtype=dtype)
        key, rng = prng.split(rng)                                                # :421
        eps = torch.as_tensor(prng.uniform(key, (2 * B,)), dtype=dtype)[:, None, None, None]
        gp_x = (eps * mix[:2 * B] + (1 - eps) * mix[2 * B:]).detach().requires_grad_(True)   # :428
        key, rng = prng.split(rng)                                                # :438 (rebound before use)
        key, rng = prng.split(rng)                                                # :451 the dropout key both applies see
        F8 = enc.shape[-1] * 8
        d1 = _dropout_mask(key, 0, rate, (4 * B, F8), dtype)
        d2 = _dropout_mask(key, 1, rate, (4 * B, 256), dtype)
        g1 = _dropout_mask(key, 0, rate, (1, F8), dtype)
        g2 = _dropout_mask(key, 1, rate, (1, 256), dtype)
        ts = state["vice"][p_key]
        p = ol.to_torch(ts["params"], dtype, requires_grad=True)
        y_hat = vice_logits(p, mix, ln, d1, d2)
        bce = lambda l: (torch.nn.functional.softplus(y_hat) - l * y_hat).mean()  # optax.sigmoid_binary_cross_entropy
        bce_loss = lam * bce(y_a) + (1 - lam) * bce(y_b)                          # :447-449
        yg = vice_logits(p, gp_x, ln, g1, g2)
        (grads_x,) = torch.autograd.grad(yg.sum(), gp_x, create_graph=True)       # vmap(grad(helper_fn)) :459-460
        gnorm = torch.sqrt((grads_x.reshape(2 * B, -1) ** 2 + 1e-6).sum(1))       # :463
        loss = bce_loss + 10 * ((gnorm - 1) ** 2).mean()                          # :464-465
        leaves = ol.tree_leaves(p)
        gl = torch.autograd.grad(loss, [t for _, t in leaves], allow_unused=True)
        grads = ol.to_numpy(ol._rebuild(p, [g if g is not None else torch.zeros_like(t) for g, (_, t) in zip(gl, leaves)]))
        new_vice[p_key] = ol.adam_step(ts, grads, cfg["vice_lr"])
        info[f"bce_loss_{p_key}"] = float(bce_loss.detach())
        info[f"grad_norm_{p_key}"] = float(gnorm.detach().mean())
        if collect is not None:
            collect.append(dict(camera=p_key, grads=grads, lam=lam, perm=perm, indx=indx, all_pixels=all_pixels,
                                gnorm=gnorm.detach().numpy()))
    new = dict(state)
    new["vice"] = new_vice
    new["rng"] = rng
    return new, info
