"""serl_convs_wgrad_noalloc (include/serl_hip.h): the conv weight gradient with its split-K slabs in a file-scope
device array of the library -- for callers that cannot call hipMalloc (graph capture, serving loops).  It has no
counterpart in the reference; the bar is its sibling serl_convs_wgrad: bit-identical dW / dbias, on two calls in a
row, the second with a smaller input (the array keeps the first call's contents), and a clear error beyond the
array's size.  Collected last (file name) so that a failure here cannot hide other tests under `pytest -x`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(dev, ops, rng, G, B, IH, C, F, u8, K=3, stride=2):
    st = ops.current_stream()
    OH = (IH - K) // stride + 1
    xs, dzs, keep = [], [], []
    for g in range(G):
        if u8:
            x = torch.from_numpy(rng.integers(0, 256, size=(B, IH, IH, C), dtype=np.uint8)).to(dev)
        else:
            xf = torch.from_numpy(np.maximum(rng.standard_normal((B, IH, IH, C)), 0).astype(np.float32)).to(dev)
            x = torch.empty_like(xf)
            ops.split_f32(xf.data_ptr(), x.data_ptr(), B * IH * IH, C, st)
            keep.append(xf)
        dzf = torch.from_numpy(rng.standard_normal((B, OH, OH, F)).astype(np.float32)).to(dev)
        dz = torch.empty_like(dzf)
        ops.split_f32(dzf.data_ptr(), dz.data_ptr(), B * OH * OH, F, st)
        xs.append(x); dzs.append(dz); keep.append(dzf)
    need = ops.conv2d_wgrad_workspace_bytes(G, B, IH, IH, C, K, K, stride, F)
    out = []
    for variant in ("workspace", "noalloc"):
        dws = [torch.full((K, K, C, F), float("nan"), device=dev) for _ in range(G)]
        dbs = [torch.full((F,), float("nan"), device=dev) for _ in range(G)]
        args = ([t.data_ptr() for t in xs], [t.data_ptr() for t in dzs], [t.data_ptr() for t in dws],
                [t.data_ptr() for t in dbs])
        if variant == "workspace":
            wsp = torch.empty(need // 4, device=dev)
            ops.convs_wgrad(*args, wsp.data_ptr(), need, u8, B, IH, IH, C, K, K, stride, F, st)
        else:
            ops.convs_wgrad_noalloc(*args, u8, B, IH, IH, C, K, K, stride, F, st)
        torch.cuda.synchronize()
        out.append((dws, dbs))
    for (a, b) in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, b), "no-alloc weight gradient differs from serl_convs_wgrad"
    return need


def test_wgrad_noalloc_matches_workspace_variant(dev):
    from serl_amd import _lib, ops
    rng = np.random.default_rng(3)
    limit = _lib.load().serl_convs_wgrad_noalloc_max_bytes()
    assert limit == 64 << 20
    # first call: E1 conv3 at batch 64, 2 groups (large slabs); second call: a smaller problem reusing the array
    n1 = _case(dev, ops, rng, 2, 64, 31, 64, 128, False)
    n2 = _case(dev, ops, rng, 1, 3, 15, 8, 16, False)
    assert n2 < n1 <= limit
    _case(dev, ops, rng, 2, 4, 128, 9, 32, True)          # uint8 first layer
    # C = 32 -> F = 64 at IH = 63: the LDS-patch weight gradient (pconv_wgrad_kernel), whose slab count -- the summation
    # order -- must come from the geometry alone, not from the size of the workspace (exact-need here, 64 MiB there)
    _case(dev, ops, rng, 2, 16, 63, 32, 64, False)
    _case(dev, ops, rng, 2, 64, 63, 32, 64, False)
    _case(dev, ops, rng, 3, 5, 63, 32, 64, False)
    # beyond the array: rejected with a clear error, nothing launched
    big = ops.conv2d_wgrad_workspace_bytes(8, 64, 15, 15, 256, 3, 3, 2, 512)
    assert big > limit
    d = torch.zeros(16, device=dev)
    with pytest.raises(ValueError, match="67108864 available"):
        ops.convs_wgrad_noalloc([d.data_ptr()] * 8, [d.data_ptr()] * 8, [d.data_ptr()] * 8, [d.data_ptr()] * 8, False,
                                64, 15, 15, 256, 3, 3, 2, 512, ops.current_stream())
