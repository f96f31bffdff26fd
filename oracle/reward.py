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
