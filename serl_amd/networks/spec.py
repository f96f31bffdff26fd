"""Static description (shapes, flax leaf names, flat layouts) and host-side initialisation of the
networks on the learner path.

Mirrors what `create()` builds in the reference:
  * critic  = [PixelMultiplexer(TwoD4PGEncoder) ->] Ensemble(StateActionValue(MLP))
      serl/agents/sac/sac_learner.py:75-89, serl/agents/drq/drq_learner.py:95-116
  * actor   = [PixelMultiplexer(stop_gradient) ->] TanhNormal(MLP)      sac_learner.py:66-73, drq_learner.py:78-93
Initialisers follow flax (xavier_uniform for Dense/Conv, lecun_normal for the spatial embeddings,
zeros/ones for bias/LayerNorm) but the *random stream* is numpy's (flax's path-folded init keys are
not restatable offline) -- parity tests therefore pass initial parameters explicitly.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from .params import Layout

TRUNK_DIM = 512        # hard-coded in serl/networks/pixel_multiplexer.py:34
SLE_FEATURES = 8       # two_d4pg_encoder.py:40


@dataclass(frozen=True)
class ConvLayer:
    IH: int; IW: int; C: int; K: int; S: int; F: int; OH: int; OW: int


@dataclass(frozen=True)
class NetConfig:
    action_dim: int
    hidden_dims: Tuple[int, ...] = (256, 256)
    num_qs: int = 2
    critic_layer_norm: bool = False
    # state-based
    obs_dim: int = 0
    # pixel-based
    pixel_keys: Tuple[str, ...] = ()          # as given by the caller (augmentation order)
    state_dim: int = 0
    H: int = 0; W: int = 0; stack: int = 0
    cnn_features: Tuple[int, ...] = (); cnn_filters: Tuple[int, ...] = (); cnn_strides: Tuple[int, ...] = ()
    latent_dim: int = 50
    # image encoder: "d4pg" (TwoD4PGEncoder: trainable conv stack) or "mobilenet" (TwoMobileNetEncoder: frozen
    # MobileNetV3-Large backbone whose (feat_h, feat_w, feat_c) feature map feeds the trainable spatial embeddings,
    # followed by Dropout(encoder_dropout) in training mode; two_mobilenet_encoder.py:10-45)
    encoder: str = "d4pg"
    feat_h: int = 0; feat_w: int = 0; feat_c: int = 0
    encoder_dropout: float = 0.1

    @property
    def pixel(self):
        return len(self.pixel_keys) > 0

    @property
    def cams(self):
        """Encoder visiting order = sorted keys (dict order after tree_map, two_d4pg_encoder.py:23)."""
        return tuple(sorted(self.pixel_keys))

    def conv_layers(self):
        out = []
        if self.encoder != "d4pg":
            return out
        h, w, c = self.H, self.W, 3 * self.stack
        for f, k, s in zip(self.cnn_features, self.cnn_filters, self.cnn_strides):
            oh, ow = (h - k) // s + 1, (w - k) // s + 1
            out.append(ConvLayer(h, w, c, k, s, f, oh, ow))
            h, w, c = oh, ow, f
        return out

    @property
    def sle_shape(self):
        """(H, W, C) of the feature map the spatial embeddings pool."""
        if self.encoder == "d4pg":
            last = self.conv_layers()[-1]
            return last.OH, last.OW, last.F
        return self.feat_h, self.feat_w, self.feat_c

    @property
    def enc_dim(self):
        return len(self.pixel_keys) * self.sle_shape[2] * SLE_FEATURES

    @property
    def latent(self):
        """Width of the vector the heads see."""
        return (TRUNK_DIM + self.latent_dim) if self.pixel else self.obs_dim


def _add_encoder_and_trunk(lay: Layout, cfg: NetConfig):
    layers = cfg.conv_layers()
    nl = len(layers)
    for ci in range(len(cfg.cams)):
        for li, L in enumerate(layers):
            name = f"Conv_{ci * nl + li}"
            lay.add(("image_encoder", name, "kernel"), (L.K, L.K, L.C, L.F))
            lay.add(("image_encoder", name, "bias"), (L.F,))
    fh, fw, fc = cfg.sle_shape
    for ci in range(len(cfg.cams)):
        lay.add(("image_encoder", f"SpatialLearnedEmbeddings_{ci}", "kernel"), (fh, fw, fc, SLE_FEATURES))
    lay.add(("Dense_0", "kernel"), (cfg.enc_dim, TRUNK_DIM)); lay.add(("Dense_0", "bias"), (TRUNK_DIM,))
    lay.add(("LayerNorm_0", "scale"), (TRUNK_DIM,)); lay.add(("LayerNorm_0", "bias"), (TRUNK_DIM,))
    lay.add(("Dense_1", "kernel"), (cfg.state_dim, cfg.latent_dim)); lay.add(("Dense_1", "bias"), (cfg.latent_dim,))
    lay.add(("LayerNorm_1", "scale"), (cfg.latent_dim,)); lay.add(("LayerNorm_1", "bias"), (cfg.latent_dim,))


def critic_head_prefix(cfg: NetConfig):
    return ("Ensemble_0", "VmapStateActionValue_0") if cfg.pixel else ("VmapStateActionValue_0",)


def actor_head_prefix(cfg: NetConfig):
    return ("Normal_0",) if cfg.pixel else ()


def critic_layout(cfg: NetConfig) -> Layout:
    lay = Layout()
    if cfg.pixel:
        _add_encoder_and_trunk(lay, cfg)
    pre = critic_head_prefix(cfg)
    E = cfg.num_qs
    din = cfg.latent + cfg.action_dim
    for i, h in enumerate(cfg.hidden_dims):
        lay.add(pre + ("MLP_0", f"Dense_{i}", "kernel"), (E, din, h))
        lay.add(pre + ("MLP_0", f"Dense_{i}", "bias"), (E, h))
        if cfg.critic_layer_norm:
            lay.add(pre + ("MLP_0", f"LayerNorm_{i}", "scale"), (E, h))
            lay.add(pre + ("MLP_0", f"LayerNorm_{i}", "bias"), (E, h))
        din = h
    lay.add(pre + ("Dense_0", "kernel"), (E, din, 1))
    lay.add(pre + ("Dense_0", "bias"), (E, 1))
    return lay


def discrete_actor_layout(cfg: NetConfig) -> Layout:
    """PixelMultiplexer(stop_gradient=True) -> OneDimOutput(MLP) -> Dense(1): the gripper head of PixelHybridBCLearner
    (serl/agents/bc/pixel_hybrid_bc_learner.py:91-106, serl/networks/one_d_output.py:7-21).  Same memory order as a
    one-member critic without LayerNorm and without the action input (the driver's "critic" slot)."""
    lay = Layout()
    _add_encoder_and_trunk(lay, cfg)
    pre = ("OneDimOutput_0",)
    din = cfg.latent
    for i, h in enumerate(cfg.hidden_dims):
        lay.add(pre + ("MLP_0", f"Dense_{i}", "kernel"), (din, h))
        lay.add(pre + ("MLP_0", f"Dense_{i}", "bias"), (h,))
        din = h
    lay.add(pre + ("Dense_0", "kernel"), (din, 1))
    lay.add(pre + ("Dense_0", "bias"), (1,))
    return lay


def actor_layout(cfg: NetConfig) -> Layout:
    lay = Layout()
    if cfg.pixel:
        _add_encoder_and_trunk(lay, cfg)
    pre = actor_head_prefix(cfg)
    din = cfg.latent
    for i, h in enumerate(cfg.hidden_dims):
        lay.add(pre + ("MLP_0", f"Dense_{i}", "kernel"), (din, h))
        lay.add(pre + ("MLP_0", f"Dense_{i}", "bias"), (h,))
        din = h
    # the two heads sit side by side so one batched GEMM (z = 2) evaluates both
    lay.add(pre + ("OutputDenseMean", "kernel"), (din, cfg.action_dim))
    lay.add(pre + ("OutputDenseLogStd", "kernel"), (din, cfg.action_dim))
    lay.add(pre + ("OutputDenseMean", "bias"), (cfg.action_dim,))
    lay.add(pre + ("OutputDenseLogStd", "bias"), (cfg.action_dim,))
    return lay


# ------------------------------------------------------------------------------------ initialisers
def _xavier_uniform(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _lecun_normal_trunc(rng, shape, fan_in):
    std = np.sqrt(1.0 / fan_in) / 0.87962566103423978
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def init_tree(layout: Layout, seed_seq) -> dict:
    """Fill a parameter tree for `layout` from its leaf names/shapes."""
    rng = np.random.default_rng(seed_seq)
    tree = {}
    for path, (_, shape) in layout.entries.items():
        leaf, parent = path[-1], path[-2]
        ens = any(p.startswith("VmapStateActionValue") for p in path)
        if parent.startswith("LayerNorm"):
            a = np.ones(shape, np.float32) if leaf == "scale" else np.zeros(shape, np.float32)
        elif leaf == "bias":
            a = np.zeros(shape, np.float32)
        elif parent.startswith("SpatialLearnedEmbeddings"):
            a = _lecun_normal_trunc(rng, shape, shape[0] * shape[1] * shape[2])
        elif parent.startswith("Conv"):
            rf = shape[0] * shape[1]
            a = _xavier_uniform(rng, shape, rf * shape[2], rf * shape[3])
        else:  # Dense kernels ((E,) in, out)
            a = _xavier_uniform(rng, shape, shape[-2], shape[-1])
        d = tree
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[leaf] = a
    return tree
