"""Frozen MobileNetV3-Large backbone (csrc/mobilenet.hip) vs the CPU restatement (oracle/mobilenet.py) on the same
synthetic weights.  The backbone itself is parity-unpinned against the reference (jeffnet + pretrained weights are
unavailable offline); this pins the HIP kernels to the restated architecture: TF SAME padding with stride 2,
depthwise 3x3 / 5x5, squeeze-excite, hard-swish / hard-sigmoid, residuals."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_layout_matches_the_library():
    from oracle import mobilenet as om
    from serl_amd.networks import mobilenet as pm
    assert [(n, tuple(s)) for n, s in om.leaf_shapes()] == [(n, tuple(s)) for n, s in pm.leaf_shapes()]
    from serl_amd import _lib
    assert pm._offsets()[1] == _lib.load().serl_mbv3_param_floats()


# (8 images of 128 x 128: every layer of the first four blocks has >= 2048 rows, i.e. runs on the streaming pointwise kernel
#  at "bf16_3x" -- all of its (k16 steps, column tiles) instantiations)
@pytest.mark.parametrize("case", [(3, 128, 128, "u8"), (2, 128, 128, "f32"), (2, 96, 80, "u8"), (1, 224, 224, "f32"),
                                  (8, 128, 128, "u8")])
def test_backbone_matches_oracle(dev, case, prec):
    from oracle import mobilenet as om
    from serl_amd.networks.mobilenet import Backbone
    n, H, W, kind = case
    bb = Backbone.synthetic(3)
    params = bb.tree()
    rng = np.random.default_rng(n * H + W)
    u8 = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
    img01 = u8.astype(np.float32) / np.float32(255.0)
    x = torch.from_numpy(u8 if kind == "u8" else img01).to(dev)
    got = bb.forward(x).cpu().numpy()
    want = om.forward(params, om.normalize(torch.tensor(img01, dtype=torch.float64)).numpy(), dtype=torch.float64).numpy()
    assert got.shape == want.shape == (n, -(-H // 32), -(-W // 32), 960)
    scale = np.abs(want).max()
    tol = (2e-5 if prec == "highest" else 2e-4) * scale
    assert np.isfinite(got).all() and scale > 1e-2
    assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), scale)


@pytest.mark.parametrize("case", [(3, 128, 128), (2, 96, 80), (1, 70, 52), (9, 128, 128)])
def test_backbone_fused_expand_depthwise_bit_identical(dev, case):
    """The high-resolution blocks' 1x1 expand + depthwise conv in one kernel (expanded tensor in LDS) against the two
    launches: same fragments, same k order, same tap loop -> the same bits, for full and ragged tiles, odd sizes with
    asymmetric SAME padding and several images per persistent block."""
    from serl_amd import _lib
    from serl_amd.networks.mobilenet import Backbone
    n, H, W = case
    bb = Backbone.synthetic(5)
    x = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device=dev)
    lib = _lib.load()
    try:
        lib.serl_mbv3_set_fused(0)
        ref = bb.forward(x).clone()
        lib.serl_mbv3_set_fused(1)
        got = bb.forward(x)
        torch.cuda.synchronize()
    finally:
        lib.serl_mbv3_set_fused(1)
    assert torch.isfinite(got).all() and float(got.abs().max()) > 1e-3
    assert torch.equal(got, ref)


def test_backbone_chunks_and_from_tree(dev):
    from serl_amd.networks.mobilenet import Backbone
    bb = Backbone.synthetic(1)
    bb2 = Backbone.from_tree(bb.tree())
    assert torch.equal(bb.flat, bb2.flat)
    x = torch.randint(0, 256, (5, 64, 64, 3), dtype=torch.uint8, device=dev)
    full = bb.forward(x)
    bb2.CHUNK = 2
    assert torch.equal(bb2.forward(x), full)
    with pytest.raises(TypeError):
        bb.forward(x.to(torch.int32))
