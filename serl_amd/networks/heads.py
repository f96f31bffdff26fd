"""Reward networks on top of the frozen MobileNetV3 features (BASELINE config 4), as launch sequences over flat
parameter stores (flax leaf names):

  BinaryClassifier   serl/networks/binary_classifier.py:5-17 with TwoMobileNetEncoder (two_mobilenet_encoder.py:19-45):
      per camera SLE(features) [Dropout(0.5) in training] -> concat -> Dense(128) -> [Dropout(0.1)] -> LayerNorm -> ReLU
      -> Dense(1).  Only inferred by the reference (drq_classifier_reward_learner.py:191-195).
  VICE classifier    EncodedEncoder(OneDimOutput(MLP((256,), leaky_relu, LayerNorm?)))  serl/networks/encoded_encoder.py:
      12-47, one_d_output.py:7-20, mlp.py:21-35, vice_learner.py:153-160,181-187: SLE(features) -> [Dropout] -> Dense(512)
      -> LayerNorm -> tanh -> Dense(256) -> [Dropout] -> [LayerNorm] -> leaky_relu -> Dense(1) -> squeeze.
Inference (training=False) here; the VICE classifier's training step is serl_amd/agents/vice/classifier_update.py."""
from __future__ import annotations

import torch

from .. import ops
from .params import Layout
from .spec import SLE_FEATURES, TRUNK_DIM

VICE_HIDDEN = 256          # vice_learner.py:157 hidden_dims=(256,)
BINARY_HIDDEN = 128        # drq_classifier_reward_learner.py:158 hidden_dim=128
ACT_LEAKY, ACT_SIGMOID = 3, 4


def binary_classifier_layout(ncam, fh, fw, fc, hidden=BINARY_HIDDEN) -> Layout:
    lay = Layout()
    for ci in range(ncam):
        lay.add(("image_encoder", f"SpatialLearnedEmbeddings_{ci}", "kernel"), (fh, fw, fc, SLE_FEATURES))
    lay.add(("Dense_0", "kernel"), (ncam * fc * SLE_FEATURES, hidden)); lay.add(("Dense_0", "bias"), (hidden,))
    lay.add(("LayerNorm_0", "scale"), (hidden,)); lay.add(("LayerNorm_0", "bias"), (hidden,))
    lay.add(("Dense_1", "kernel"), (hidden, 1)); lay.add(("Dense_1", "bias"), (1,))
    return lay


def vice_classifier_layout(fh, fw, fc, layer_norm, hidden=VICE_HIDDEN) -> Layout:
    lay = Layout()
    lay.add(("SpatialLearnedEmbeddings_0", "kernel"), (fh, fw, fc, SLE_FEATURES))
    lay.add(("Dense_0", "kernel"), (fc * SLE_FEATURES, TRUNK_DIM)); lay.add(("Dense_0", "bias"), (TRUNK_DIM,))
    lay.add(("LayerNorm_0", "scale"), (TRUNK_DIM,)); lay.add(("LayerNorm_0", "bias"), (TRUNK_DIM,))
    pre = ("OneDimOutput_0",)
    lay.add(pre + ("MLP_0", "Dense_0", "kernel"), (TRUNK_DIM, hidden)); lay.add(pre + ("MLP_0", "Dense_0", "bias"), (hidden,))
    if layer_norm:
        lay.add(pre + ("MLP_0", "LayerNorm_0", "scale"), (hidden,)); lay.add(pre + ("MLP_0", "LayerNorm_0", "bias"), (hidden,))
    lay.add(pre + ("Dense_0", "kernel"), (hidden, 1)); lay.add(pre + ("Dense_0", "bias"), (1,))
    return lay


class Scratch:
    """Named device buffers of a head evaluation (reused across calls of the same shape)."""

    def __init__(self, device):
        self.device, self._t = device, {}

    def get(self, name, shape):
        t = self._t.get(name)
        shape = tuple(int(x) for x in shape)
        if t is None or tuple(t.shape) != shape:
            t = self._t[name] = torch.empty(shape, dtype=torch.float32, device=self.device)
        return t


def binary_classifier_fwd(store, feats, n, fh, fw, fc, ws: Scratch, st, hidden=BINARY_HIDDEN):
    """feats: per camera (sorted order) device pointer of (n, fh, fw, fc) backbone features -> sigmoid(logit) (n, 1)."""
    ncam = len(feats)
    F8 = fc * SLE_FEATURES
    enc = ws.get("bc_enc", (n, ncam * F8))
    ops.sle_fwd(list(feats), [store.ptr("image_encoder", f"SpatialLearnedEmbeddings_{ci}", "kernel") for ci in range(ncam)],
                [enc.data_ptr() + 4 * ci * F8 for ci in range(ncam)], ncam * F8, n, fh * fw, fc, SLE_FEATURES, st)
    pre = ws.get("bc_pre", (n, hidden))
    ops.gemm(enc.data_ptr(), store.ptr("Dense_0", "kernel"), pre.data_ptr(), None, 1, n, hidden, ncam * F8,
             a_ms=ncam * F8, a_ks=1, b_ks=hidden, b_ns=1, c_ms=hidden, stream=st)
    h = ws.get("bc_h", (n, hidden))
    ops.ln_act_fwd(pre.data_ptr(), 1, 0, hidden, store.ptr("Dense_0", "bias"), store.ptr("LayerNorm_0", "scale"),
                   store.ptr("LayerNorm_0", "bias"), 0, h.data_ptr(), hidden, None, None, n, hidden, n, True, ops.ACT_RELU, st)
    out = ws.get("bc_out", (n, 1))
    ops.gemm(h.data_ptr(), store.ptr("Dense_1", "kernel"), out.data_ptr(), store.ptr("Dense_1", "bias"), 1, n, 1, hidden,
             a_ms=hidden, a_ks=1, b_ks=1, b_ns=1, c_ms=1, relu=ACT_SIGMOID, stream=st)
    return out


def vice_classifier_fwd(store, feat, n, fh, fw, fc, layer_norm, ws: Scratch, st, tag="", hidden=VICE_HIDDEN):
    """feat: device pointer of (n, fh, fw, fc) encoded features -> logits (n,) (training=False: no Dropout)."""
    F8 = fc * SLE_FEATURES
    s = ws.get("vc_s" + tag, (n, F8))
    ops.sle_fwd([feat], [store.ptr("SpatialLearnedEmbeddings_0", "kernel")], [s.data_ptr()], F8, n, fh * fw, fc,
                SLE_FEATURES, st)
    pre1 = ws.get("vc_pre1" + tag, (n, TRUNK_DIM))
    ops.gemm(s.data_ptr(), store.ptr("Dense_0", "kernel"), pre1.data_ptr(), None, 1, n, TRUNK_DIM, F8, a_ms=F8, a_ks=1,
             b_ks=TRUNK_DIM, b_ns=1, c_ms=TRUNK_DIM, stream=st)
    t = ws.get("vc_t" + tag, (n, TRUNK_DIM))
    ops.ln_act_fwd(pre1.data_ptr(), 1, 0, TRUNK_DIM, store.ptr("Dense_0", "bias"), store.ptr("LayerNorm_0", "scale"),
                   store.ptr("LayerNorm_0", "bias"), 0, t.data_ptr(), TRUNK_DIM, None, None, n, TRUNK_DIM, n, True,
                   ops.ACT_TANH, st)
    pre = ("OneDimOutput_0",)
    pre2 = ws.get("vc_pre2" + tag, (n, hidden))
    ops.gemm(t.data_ptr(), store.ptr(*pre, "MLP_0", "Dense_0", "kernel"), pre2.data_ptr(), None, 1, n, hidden, TRUNK_DIM,
             a_ms=TRUNK_DIM, a_ks=1, b_ks=hidden, b_ns=1, c_ms=hidden, stream=st)
    r = ws.get("vc_r" + tag, (n, hidden))
    ops.ln_act_fwd(pre2.data_ptr(), 1, 0, hidden, store.ptr(*pre, "MLP_0", "Dense_0", "bias"),
                   store.ptr(*pre, "MLP_0", "LayerNorm_0", "scale") if layer_norm else None,
                   store.ptr(*pre, "MLP_0", "LayerNorm_0", "bias") if layer_norm else None, 0, r.data_ptr(), hidden, None,
                   None, n, hidden, n, bool(layer_norm), ACT_LEAKY, st)
    y = ws.get("vc_y" + tag, (n,))
    ops.rowdot(r.data_ptr(), store.ptr(*pre, "Dense_0", "kernel"), 0, store.ptr(*pre, "Dense_0", "bias"), 0, y.data_ptr(), n,
               hidden, n, st)
    return y
