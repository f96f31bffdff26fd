"""Frozen MobileNetV3-Large backbone of `TwoMobileNetEncoder` / `MobileNetEncoder`
(serl/networks/encoders/two_mobilenet_encoder.py:10-45, mobilenet_encoder.py:9-48).

The reference builds it with `jeffnet.linen.create_model('tf_mobilenetv3_large_100', pretrained=True)`
(serl/agents/drq/drq_classifier_reward_learner.py:94-98, vice/vice_learner.py:95-97): an un-vendored dependency that
downloads its weights, unavailable offline.  This module defines the same published architecture (BatchNorm folded into
the convolutions, TensorFlow SAME padding, cut after the last 1x1 convolution: 960 channels at stride 32) over ONE flat
fp32 device buffer, evaluated by `serl_mbv3_forward` (csrc/mobilenet.hip).  `Backbone.synthetic(seed)` gives
variance-preserving random weights; `Backbone.from_tree(tree)` takes folded inference weights by leaf name
(`fold_batchnorm` folds a BatchNorm into its convolution) for users who can load the pretrained model elsewhere.
The backbone is frozen everywhere in the reference (module attribute / stop_gradient): forward only."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .params import ALIGN, _round_up

STEM_CH, HEAD_CH = 16, 960
# (inverted residual?, kernel, stride, expanded channels, output channels, squeeze-excite, activation)
BLOCKS = (
    (False, 3, 1, 16, 16, False, "relu"),
    (True, 3, 2, 64, 24, False, "relu"), (True, 3, 1, 72, 24, False, "relu"),
    (True, 5, 2, 72, 40, True, "relu"), (True, 5, 1, 120, 40, True, "relu"), (True, 5, 1, 120, 40, True, "relu"),
    (True, 3, 2, 240, 80, False, "hswish"), (True, 3, 1, 200, 80, False, "hswish"),
    (True, 3, 1, 184, 80, False, "hswish"), (True, 3, 1, 184, 80, False, "hswish"),
    (True, 3, 1, 480, 112, True, "hswish"), (True, 3, 1, 672, 112, True, "hswish"),
    (True, 5, 2, 672, 160, True, "hswish"), (True, 5, 1, 960, 160, True, "hswish"), (True, 5, 1, 960, 160, True, "hswish"),
)


def _se_channels(expanded):
    v = expanded * 0.25
    nv = max(8, int(v + 4) // 8 * 8)
    return nv + 8 if nv < 0.9 * v else nv


def leaf_shapes():
    """[(name, shape)] in the order of the flat buffer (= forward order, csrc/mobilenet.hip layout())."""
    out = [("stem/kernel", (3, 3, 3, STEM_CH)), ("stem/bias", (STEM_CH,))]
    cin = STEM_CH
    for i, (ir, k, s, exp, cout, se, act) in enumerate(BLOCKS):
        p = f"block{i}"
        if ir:
            out += [(f"{p}/expand/kernel", (cin, exp)), (f"{p}/expand/bias", (exp,))]
        out += [(f"{p}/dw/kernel", (k, k, exp)), (f"{p}/dw/bias", (exp,))]
        if se:
            r = _se_channels(exp)
            out += [(f"{p}/se_reduce/kernel", (exp, r)), (f"{p}/se_reduce/bias", (r,)),
                    (f"{p}/se_expand/kernel", (r, exp)), (f"{p}/se_expand/bias", (exp,))]
        out += [(f"{p}/project/kernel", (exp, cout)), (f"{p}/project/bias", (cout,))]
        cin = cout
    out += [("head/kernel", (cin, HEAD_CH)), ("head/bias", (HEAD_CH,))]
    return out


def _offsets():
    offs, at = {}, 0
    for name, shape in leaf_shapes():
        offs[name] = (at, shape)
        at += _round_up(int(np.prod(shape)), ALIGN)
    return offs, at


def fold_batchnorm(kernel, scale, bias, mean, var, eps=1e-3):
    """Fold an inference BatchNorm (tf_ models: eps 1e-3) into the preceding convolution -> (kernel', bias'); the output
    channel is the last axis of `kernel`."""
    inv = np.asarray(scale, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    return (np.asarray(kernel, np.float64) * inv).astype(np.float32), \
        (np.asarray(bias, np.float64) - np.asarray(mean, np.float64) * inv).astype(np.float32)


class Backbone:
    """The frozen feature extractor: flat device parameters + forward."""
    CHUNK = 1024         # images per launch sequence (bounds the activation workspace: ~1.9 MB per 128x128 image)

    def __init__(self, flat):
        self.flat = flat
        n = int(_lib.load().serl_mbv3_param_floats())
        if flat.numel() != n or flat.dtype != torch.float32:
            raise ValueError(f"MobileNetV3 backbone: {n} float32 parameters expected, got {flat.numel()} {flat.dtype}")
        offs, total = _offsets()
        assert total == n, "host / device layouts of the backbone differ"
        self._ws = None

    @classmethod
    def from_tree(cls, tree, device="cuda"):
        offs, total = _offsets()
        flat = np.zeros(total, np.float32)
        extra = set(tree) - set(offs)
        if extra:
            raise ValueError(f"unknown backbone leaves {sorted(extra)[:4]}")
        for name, (at, shape) in offs.items():
            a = np.asarray(tree[name], np.float32)
            if a.shape != shape:
                raise ValueError(f"backbone leaf {name}: shape {a.shape} != {shape}")
            flat[at:at + a.size] = a.ravel()
        return cls(torch.from_numpy(flat).to(device))

    @classmethod
    def synthetic(cls, seed=0, device="cuda"):
        """Random stand-in for the pretrained weights (there is no network to fetch them): variance-preserving normal
        kernels so that activations keep an O(1) scale through the 17 layers, small biases."""
        rng = np.random.default_rng([int(seed), 0x4D4256])
        tree = {}
        for name, shape in leaf_shapes():
            if name.endswith("bias"):
                tree[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
            else:
                fan_in = int(np.prod(shape[:-1])) if "/dw/" not in name else shape[0] * shape[1]
                gain = 1.0 if ("project" in name or "se_" in name) else np.sqrt(2.0)
                tree[name] = (gain / np.sqrt(fan_in) * rng.standard_normal(shape)).astype(np.float32)
        return cls.from_tree(tree, device)

    def tree(self):
        host = self.flat.cpu().numpy()
        offs, _ = _offsets()
        return {name: host[at:at + int(np.prod(shape))].reshape(shape).copy() for name, (at, shape) in offs.items()}

    @staticmethod
    def feature_shape(H, W):
        fh, fw, fc = C.c_int(), C.c_int(), C.c_int()
        _lib.call("serl_mbv3_feature_shape", int(H), int(W), C.byref(fh), C.byref(fw), C.byref(fc))
        return fh.value, fw.value, fc.value

    def forward(self, images, out=None, stream=None):
        """images: (n,H,W,3) device tensor, uint8 (0..255) or float32 already scaled to [0,1] -> (n,fh,fw,960) float32."""
        from .. import ops
        if images.dtype not in (torch.uint8, torch.float32) or images.ndim != 4 or images.shape[-1] != 3:
            raise TypeError(f"backbone input must be (n,H,W,3) uint8 or float32, got {tuple(images.shape)} {images.dtype}")
        images = images.contiguous()
        n, H, W, _ = images.shape
        fh, fw, fc = self.feature_shape(H, W)
        if out is None:
            out = torch.empty((n, fh, fw, fc), dtype=torch.float32, device=images.device)
        st = ops.current_stream() if stream is None else stream
        lib = _lib.load()
        chunk = min(self.CHUNK, n)
        need = int(lib.serl_mbv3_workspace_bytes(chunk, H, W))
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty(need // 4, dtype=torch.float32, device=images.device)
        esz = images.element_size()
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            _lib.call("serl_mbv3_forward", self.flat.data_ptr(), images.data_ptr() + lo * H * W * 3 * esz,
                      int(images.dtype == torch.uint8), m, H, W, self._ws.data_ptr(), self._ws.numel() * 4,
                      out.data_ptr() + lo * fh * fw * fc * 4, st)
        return out
