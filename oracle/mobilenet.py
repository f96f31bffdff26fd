"""ORACLE (test infrastructure only) -- CPU restatement of the frozen MobileNetV3-Large backbone behind
`TwoMobileNetEncoder` / `MobileNetEncoder` (serl/networks/encoders/two_mobilenet_encoder.py:19-34,
mobilenet_encoder.py:29-48) and of the reward heads that sit on it (serl/networks/binary_classifier.py:5-17,
encoded_encoder.py:12-47, one_d_output.py:7-20).

The backbone itself is NOT in /root/reference: the reference builds it with
`jeffnet.linen.create_model('tf_mobilenetv3_large_100', pretrained=True)` (drq_classifier_reward_learner.py:94-98,
vice_learner.py:95-97), an un-vendored git dependency (README.md:30, unpinned) that downloads its weights.  What is
restated here is the PUBLISHED architecture of that model name (Howard et al. 2019, "Searching for MobileNetV3", table 1,
in the `tf_` flavour: TensorFlow "SAME" padding, BatchNorm folded into the convolutions for inference, squeeze-excite
with a ReLU bottleneck of make_divisible(expanded/4, 8) channels and a hard-sigmoid gate on the EXPANDED tensor,
hard-swish in the stem and the last three stages), cut where a feature extractor cuts it: after the last 1x1 convolution
(960 channels, stride 32), before pooling.  Weights are synthetic (no network).  PARITY UNPINNED for the backbone: which
stage the jeffnet fork returns, and its trained weights, cannot be checked offline; everything downstream only needs a
frozen NHWC feature map, and is restated from the reference's own files.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this package."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # two_mobilenet_encoder.py:19, mobilenet_encoder.py:29
IMAGENET_STD = (0.229, 0.224, 0.225)      # :20 / :30
STEM_CH = 16
HEAD_CH = 960
# (kind, kernel, stride, expanded channels, output channels, squeeze-excite, activation)
BLOCKS = (
    ("ds", 3, 1, 16, 16, False, "relu"),
    ("ir", 3, 2, 64, 24, False, "relu"), ("ir", 3, 1, 72, 24, False, "relu"),
    ("ir", 5, 2, 72, 40, True, "relu"), ("ir", 5, 1, 120, 40, True, "relu"), ("ir", 5, 1, 120, 40, True, "relu"),
    ("ir", 3, 2, 240, 80, False, "hswish"), ("ir", 3, 1, 200, 80, False, "hswish"),
    ("ir", 3, 1, 184, 80, False, "hswish"), ("ir", 3, 1, 184, 80, False, "hswish"),
    ("ir", 3, 1, 480, 112, True, "hswish"), ("ir", 3, 1, 672, 112, True, "hswish"),
    ("ir", 5, 2, 672, 160, True, "hswish"), ("ir", 5, 1, 960, 160, True, "hswish"), ("ir", 5, 1, 960, 160, True, "hswish"),
)


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def se_channels(expanded):
    return make_divisible(expanded * 0.25, 8)


def leaf_shapes():
    """[(name, shape)] of the folded inference parameters, in forward order.  Pointwise kernels are (Cin, Cout), depthwise
    kernels (k, k, C), the stem (3, 3, 3, 16) HWIO."""
    out = [("stem/kernel", (3, 3, 3, STEM_CH)), ("stem/bias", (STEM_CH,))]
    cin = STEM_CH
    for i, (kind, k, s, exp, cout, se, act) in enumerate(BLOCKS):
        p = f"block{i}"
        if kind == "ir":
            out += [(f"{p}/expand/kernel", (cin, exp)), (f"{p}/expand/bias", (exp,))]
        out += [(f"{p}/dw/kernel", (k, k, exp)), (f"{p}/dw/bias", (exp,))]
        if se:
            r = se_channels(exp)
            out += [(f"{p}/se_reduce/kernel", (exp, r)), (f"{p}/se_reduce/bias", (r,)),
                    (f"{p}/se_expand/kernel", (r, exp)), (f"{p}/se_expand/bias", (exp,))]
        out += [(f"{p}/project/kernel", (exp, cout)), (f"{p}/project/bias", (cout,))]
        cin = cout
    out += [("head/kernel", (cin, HEAD_CH)), ("head/bias", (HEAD_CH,))]
    return out


def init_params(seed=0):
    """Synthetic stand-in for the pretrained weights: variance-preserving normal kernels (so activations keep an O(1)
    scale through the 17 layers), small biases."""
    rng = np.random.default_rng([seed, 0x4D4256])
    p = {}
    for name, shape in leaf_shapes():
        if name.endswith("bias"):
            p[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[:-1])) if "/dw/" not in name else shape[0] * shape[1]
            gain = 1.0 if "project" in name or "se_" in name else np.sqrt(2.0)
            p[name] = (gain / np.sqrt(fan_in) * rng.standard_normal(shape)).astype(np.float32)
    return p


def hswish(x):
    return x * torch.clamp(x + 3.0, 0.0, 6.0) / 6.0


def hsigmoid(x):
    return torch.clamp(x + 3.0, 0.0, 6.0) / 6.0


def _act(x, name):
    return torch.relu(x) if name == "relu" else hswish(x)


def same_pad(size, k, s):
    """TensorFlow 'SAME': (before, after, out)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2, out


def _conv_same(x, w, b, stride, groups=1):
    """x (n,C,H,W); w torch layout (O, I/groups, k, k); TF SAME padding."""
    k = w.shape[-1]
    pt, pb, _ = same_pad(x.shape[2], k, stride)
    pl, pr, _ = same_pad(x.shape[3], k, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, b, stride=stride, groups=groups)


def normalize(img01):
    """(x - mean) / std on a (..., 3) image already scaled to [0, 1] (two_mobilenet_encoder.py:30)."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=img01.dtype)
    std = torch.tensor(IMAGENET_STD, dtype=img01.dtype)
    return (img01 - mean) / std


def forward(params, x_nhwc, dtype=torch.float32):
    """Backbone on normalised NHWC images (n,H,W,3) -> (n, H/32, W/32, 960) feature map."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    x = torch.as_tensor(np.asarray(x_nhwc), dtype=dtype).permute(0, 3, 1, 2)
    x = hswish(_conv_same(x, t(params["stem/kernel"]).permute(3, 2, 0, 1), t(params["stem/bias"]), 2))
    for i, (kind, k, s, exp, cout, se, act) in enumerate(BLOCKS):
        p = f"block{i}"
        inp = x
        if kind == "ir":
            w = t(params[f"{p}/expand/kernel"])                                    # (Cin, exp)
            x = _act(F.conv2d(x, w.t()[:, :, None, None], t(params[f"{p}/expand/bias"])), act)
        wd = t(params[f"{p}/dw/kernel"]).permute(2, 0, 1)[:, None]                  # (C,1,k,k)
        x = _act(_conv_same(x, wd, t(params[f"{p}/dw/bias"]), s, groups=exp), act)
        if se:
            z = x.mean(dim=(2, 3))                                                  # (n, exp)
            z = torch.relu(z @ t(params[f"{p}/se_reduce/kernel"]) + t(params[f"{p}/se_reduce/bias"]))
            g = hsigmoid(z @ t(params[f"{p}/se_expand/kernel"]) + t(params[f"{p}/se_expand/bias"]))
            x = x * g[:, :, None, None]
        w = t(params[f"{p}/project/kernel"])
        x = F.conv2d(x, w.t()[:, :, None, None], t(params[f"{p}/project/bias"]))
        if s == 1 and inp.shape[1] == cout:
            x = x + inp
    x = hswish(F.conv2d(x, t(params["head/kernel"]).t()[:, :, None, None], t(params["head/bias"])))
    return x.permute(0, 2, 3, 1).contiguous()


def fold_batchnorm(kernel, scale, bias, mean, var, eps=1e-3):
    """Fold an inference BatchNorm (tf_ models: eps 1e-3) into the preceding convolution: returns (kernel', bias').
    The output channel is the last axis of `kernel` (depthwise kernels (k,k,C): the channel axis)."""
    inv = np.asarray(scale, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    return (np.asarray(kernel, np.float64) * inv).astype(np.float32), \
        (np.asarray(bias, np.float64) - np.asarray(mean, np.float64) * inv).astype(np.float32)
