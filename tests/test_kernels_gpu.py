"""GPU parity tests of every HIP kernel, called through the C ABI, against the CPU oracle
(oracle/learner.py, oracle/prng.py) or a float64 numpy/torch statement of the same op.

Tolerances: integer / byte work bit-exact; fp32 kernels rel 2e-5 of the output scale
(the fp32 MFMA is an exact fmaf chain; differences are summation order only); conv / GEMM at
precision "bf16_3x" rel 1e-4 of the output scale (<= 2^-16 relative error per product)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from serl_amd import ops
    return ops


def close(got, ref, rtol=2e-5, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(1e-30, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max()) / scale
    assert err <= rtol, f"{what}: max err {err:.3e} of scale {scale:.3e}"


def dv(a, dev, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


# ---------------------------------------------------------------------------------- igemm: conv
CONV_CASES = [
    # B, IH, IW, C, K, S, F, u8
    (3, 128, 128, 9, 3, 2, 32, True),     # E1 conv1 (uint8 input, K=81)
    (2, 63, 63, 32, 3, 2, 64, False),     # E1 conv2
    (2, 31, 31, 64, 3, 2, 128, False),    # E1 conv3
    (5, 15, 15, 128, 3, 2, 256, False),   # E1 conv4
    (2, 21, 19, 32, 3, 1, 32, False),     # E0-style stride 1, non-square
    (2, 128, 128, 9, 3, 2, 32, True),     # E0 conv1 at the full 128x128 frame (DrQLearner defaults: strides 2,1,1,1)
    (1, 63, 63, 32, 3, 1, 32, False),     # E0 conv2 at full size: stride 1, 63 -> 61
    (1, 59, 59, 32, 3, 1, 32, False),     # E0 conv4 at full size: 59 -> 57
    (1, 9, 9, 3, 2, 1, 5, False),         # odd sizes, filter 2
]


def _conv_ref(x, w, b, stride):
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    wt = torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)
    y = torch.nn.functional.conv2d(xt, wt, torch.tensor(b, dtype=torch.float64), stride=stride)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case, prec):
    ops = _ops()
    tol = 2e-5 if prec == "highest" else 1e-4
    B, IH, IW, Cin, K, S, F, u8 = case
    rng = np.random.default_rng(1)
    G = 2
    st = ops.current_stream()
    xs, ws, bs, ys, refs = [], [], [], [], []
    for g in range(G):
        if u8:
            x = rng.integers(0, 256, size=(B, IH, IW, Cin), dtype=np.uint8)
            xf = (x.astype(np.float32) / np.float32(255.0))
        else:
            x = np.maximum(rng.standard_normal((B, IH, IW, Cin)).astype(np.float32), 0)
            xf = x
        w = (rng.standard_normal((K, K, Cin, F)) * 0.1).astype(np.float32)
        b = (rng.standard_normal(F) * 0.1).astype(np.float32)
        xs.append(dv(x, dev)); ws.append(dv(w, dev)); bs.append(dv(b, dev))
        OH, OW = (IH - K) // S + 1, (IW - K) // S + 1
        ys.append(torch.full((B, OH, OW, F), float("nan"), device=dev))
        refs.append((xf, w, b))
    ops.conv2d_fwd([t.data_ptr() for t in xs], [t.data_ptr() for t in ws], [t.data_ptr() for t in bs],
                   [t.data_ptr() for t in ys], u8, B, IH, IW, Cin, K, K, S, F, True, st)
    torch.cuda.synchronize()
    dzs, dxs, dws, dbs = [], [], [], []
    for g in range(G):
        xf, w, b = refs[g]
        yref = torch.relu(_conv_ref(xf, w, b, S))
        close(ys[g].cpu().numpy(), yref.numpy(), rtol=tol, what=f"conv fwd g{g}")
        dz = rng.standard_normal((B, OH, OW, F)).astype(np.float32)
        dz *= (yref.numpy() > 0)                      # already ReLU-masked, as the chain delivers it
        dzs.append(dv(dz, dev))
        dxs.append(torch.full((B, IH, IW, Cin), float("nan"), device=dev))
        dws.append(torch.full((K, K, Cin, F), float("nan"), device=dev))
        dbs.append(torch.full((F,), float("nan"), device=dev))
    # --- wgrad (+ bias grad)
    wsb = ops.conv2d_wgrad_workspace_bytes(G, B, IH, IW, Cin, K, K, S, F)
    wsp = torch.empty(wsb // 4, device=dev)
    ops.conv2d_wgrad([t.data_ptr() for t in xs], [t.data_ptr() for t in dzs], [t.data_ptr() for t in dws],
                     [t.data_ptr() for t in dbs], wsp.data_ptr(), wsb, u8, B, IH, IW, Cin, K, K, S, F, st)
    # --- dgrad with the ReLU mask of this layer's input
    if not u8:
        ops.conv2d_dgrad([t.data_ptr() for t in dzs], [t.data_ptr() for t in ws], [t.data_ptr() for t in xs],
                         [t.data_ptr() for t in dxs], B, IH, IW, Cin, K, K, S, F, st)
    torch.cuda.synchronize()
    for g in range(G):
        xf, w, b = refs[g]
        xt = torch.tensor(xf, dtype=torch.float64, requires_grad=True)
        wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), bt, stride=S).permute(0, 2, 3, 1)
        (y * torch.tensor(dzs[g].cpu().numpy(), dtype=torch.float64)).sum().backward()
        close(dws[g].cpu().numpy(), wt.grad.numpy(), rtol=tol, what=f"conv wgrad g{g}")
        close(dbs[g].cpu().numpy(), bt.grad.numpy(), rtol=tol, what=f"conv bias grad g{g}")
        if not u8:
            ref_dx = xt.grad.numpy() * (xf > 0)
            close(dxs[g].cpu().numpy(), ref_dx, rtol=tol, what=f"conv dgrad g{g}")


# ---------------------------------------------------------------------------------- sgemm: S-format conv stack
def _to_s(ops, t, rows, cols, st):
    """fp32 device tensor -> (S-format device buffer, the fp32 values it represents)."""
    buf = torch.empty(rows * cols * 2, dtype=torch.int16, device=t.device)
    ops.split_f32(t.data_ptr(), buf.data_ptr(), rows, cols, st)
    back = torch.empty(rows * cols, dtype=torch.float32, device=t.device)
    ops.merge_split(buf.data_ptr(), back.data_ptr(), rows, cols, st)
    torch.cuda.synchronize()
    return buf, back.cpu().numpy()


def _from_s(ops, buf, rows, cols, st):
    out = torch.empty(rows * cols, dtype=torch.float32, device=buf.device)
    ops.merge_split(buf.data_ptr(), out.data_ptr(), rows, cols, st)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_split_merge_roundtrip(dev):
    ops = _ops()
    st = ops.current_stream()
    x = np.random.default_rng(0).standard_normal((37, 24)).astype(np.float32) * 3
    _, back = _to_s(ops, dv(x, dev), 37, 24, st)
    assert np.abs(back.reshape(37, 24) - x).max() <= 2.0 ** -16 * np.abs(x).max()     # 16 mantissa bits kept


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[6] % 8 == 0 and (c[7] or c[3] % 8 == 0)])
def test_convs_fwd_dgrad_wgrad(dev, case):
    """Same checks as test_conv_fwd_dgrad_wgrad on the S-format entry points; the float64 reference is fed the
    values the S-format inputs actually hold. Tolerance 1e-4 of the output scale (bf16_3x products + the 2^-17
    rounding of the S-format outputs)."""
    ops = _ops()
    tol = 1e-4
    B, IH, IW, Cin, K, S, F, u8 = case
    rng = np.random.default_rng(1)
    G = 2
    st = ops.current_stream()
    OH, OW = (IH - K) // S + 1, (IW - K) // S + 1
    xs, wfs, wds, bs, ys, refs, keep = [], [], [], [], [], [], []
    wb = ops.convs_weight_bytes(u8, K, K, Cin, F)
    for g in range(G):
        w = (rng.standard_normal((K, K, Cin, F)) * 0.1).astype(np.float32)
        b = (rng.standard_normal(F) * 0.1).astype(np.float32)
        if u8:
            x = rng.integers(0, 256, size=(B, IH, IW, Cin), dtype=np.uint8)
            xf = (x.astype(np.float32) / np.float32(255.0))
            xs.append(dv(x, dev))
        else:
            x = np.maximum(rng.standard_normal((B, IH, IW, Cin)).astype(np.float32), 0)
            buf, xf = _to_s(ops, dv(x, dev), B * IH * IW, Cin, st)
            xf = xf.reshape(B, IH, IW, Cin)
            xs.append(buf)
        W = dv(w, dev)
        wf = torch.full((wb // 2,), -1, dtype=torch.int16, device=dev)
        wd = torch.full((K * K * Cin * F * 2,), -1, dtype=torch.int16, device=dev)
        keep.append(W)
        wfs.append(wf); wds.append(wd); bs.append(dv(b, dev))
        ys.append(torch.full((B * OH * OW * F * 2,), -1, dtype=torch.int16, device=dev))
        refs.append((xf, w, b))
    ops.convs_prep_weights([t.data_ptr() for t in keep], [t.data_ptr() for t in wfs], [t.data_ptr() for t in wds],
                           u8, K, K, Cin, F, st)
    rmask0 = torch.full((B * OH * OW * F // 8,), 0xAA, dtype=torch.uint8, device=dev)   # group 0 only
    ops.convs_fwd([t.data_ptr() for t in xs], [t.data_ptr() for t in wfs], [t.data_ptr() for t in bs],
                  [t.data_ptr() for t in ys], u8, B, IH, IW, Cin, K, K, S, F, True, st,
                  rmasks=[rmask0.data_ptr(), 0])
    torch.cuda.synchronize()
    y0 = _from_s(ops, ys[0], B * OH * OW, F, st).reshape(-1, F)
    np.testing.assert_array_equal(rmask0.cpu().numpy(), np.packbits(y0 != 0, axis=1, bitorder="little").reshape(-1))
    dzs, dzf, dxs, dws, dbs = [], [], [], [], []
    for g in range(G):
        xf, w, b = refs[g]
        yref = torch.relu(_conv_ref(xf, w, b, S))
        got = _from_s(ops, ys[g], B * OH * OW, F, st).reshape(B, OH, OW, F)
        close(got, yref.numpy(), rtol=tol, what=f"convs fwd g{g}")
        dz = rng.standard_normal((B, OH, OW, F)).astype(np.float32)
        dz *= (yref.numpy() > 0)
        buf, dzv = _to_s(ops, dv(dz, dev), B * OH * OW, F, st)
        dzs.append(buf); dzf.append(dzv.reshape(B, OH, OW, F))
        dxs.append(torch.full((B * IH * IW * Cin * 2,), -1, dtype=torch.int16, device=dev))
        dws.append(torch.full((K, K, Cin, F), float("nan"), device=dev))
        dbs.append(torch.full((F,), float("nan"), device=dev))
    wsb = ops.conv2d_wgrad_workspace_bytes(G, B, IH, IW, Cin, K, K, S, F)
    wsp = torch.empty(wsb // 4, device=dev)
    ops.convs_wgrad([t.data_ptr() for t in xs], [t.data_ptr() for t in dzs], [t.data_ptr() for t in dws],
                    [t.data_ptr() for t in dbs], wsp.data_ptr(), wsb, u8, B, IH, IW, Cin, K, K, S, F, st)
    if not u8:
        ops.convs_dgrad([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], [t.data_ptr() for t in xs],
                        [t.data_ptr() for t in dxs], B, IH, IW, Cin, K, K, S, F, st)
        # the same data gradient masked by the byte mask of x instead of x's hi plane: bit-identical
        xm = [dv(np.packbits(refs[g][0].reshape(-1, Cin) != 0, axis=1, bitorder="little").reshape(-1), dev)
              for g in range(G)]
        dxb = [torch.full_like(t, -1) for t in dxs]
        ops.convs_dgrad([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], [t.data_ptr() for t in xm],
                        [t.data_ptr() for t in dxb], B, IH, IW, Cin, K, K, S, F, st, xact_bits=True)
        torch.cuda.synchronize()
        for g in range(G):
            assert torch.equal(dxb[g], dxs[g]), f"byte-mask dgrad differs from hi-plane-mask dgrad, group {g}"
    torch.cuda.synchronize()
    for g in range(G):
        xf, w, b = refs[g]
        xt = torch.tensor(xf, dtype=torch.float64, requires_grad=True)
        wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), bt, stride=S).permute(0, 2, 3, 1)
        (y * torch.tensor(dzf[g], dtype=torch.float64)).sum().backward()
        close(dws[g].cpu().numpy(), wt.grad.numpy(), rtol=tol, what=f"convs wgrad g{g}")
        close(dbs[g].cpu().numpy(), bt.grad.numpy(), rtol=tol, what=f"convs bias grad g{g}")
        if not u8:
            ref_dx = xt.grad.numpy() * (xf > 0)
            got = _from_s(ops, dxs[g], B * IH * IW, Cin, st).reshape(B, IH, IW, Cin)
            close(got, ref_dx, rtol=tol, what=f"convs dgrad g{g}")


@pytest.mark.parametrize("case", [(3, 63, 63, 32, 64), (2, 64, 63, 32, 64), (3, 31, 31, 64, 128), (2, 29, 32, 64, 128),
                                  (1, 9, 11, 32, 64), (5, 15, 15, 128, 256), (1, 16, 13, 128, 256), (1, 7, 7, 64, 136),
                                  (40, 31, 31, 64, 128), (30, 13, 9, 64, 128),
                                  # C=32 -> 64 with >= 4 items per persistent block (85 blocks per group): the double-
                                  # buffered patch, the counted vmcnt waits with the previous item's stores in flight and
                                  # the weight gradient's accumulators kept over items (pconv.h) past the first item
                                  (40, 63, 63, 32, 64), (33, 64, 63, 32, 64), (64, 63, 63, 32, 64)])
def test_patch_conv_bit_identical_to_im2col(dev, case):
    """The LDS-patch kernels (csrc/pconv.h) and the LDS-DMA ring GEMM (csrc/sgemm_ring.h, layers with F >= 128) keep the
    k order and the MFMA sequence of the im2col kernels: outputs, ReLU byte masks and data gradients must be
    bit-identical, over full and ragged bands / row tiles, odd/even widths, a row narrower than the lane tile and a
    channel count that is not a multiple of the column tile."""
    ops = _ops()
    B, IH, IW, Cin, F = case
    K, S, G = 3, 2, 3
    rng = np.random.default_rng(7)
    st = ops.current_stream()
    OH, OW = (IH - K) // S + 1, (IW - K) // S + 1
    wb = ops.convs_weight_bytes(False, K, K, Cin, F)
    xs, wfs, wds, bs, keep, dzs, xm = [], [], [], [], [], [], []
    for g in range(G):
        x = np.maximum(rng.standard_normal((B, IH, IW, Cin)).astype(np.float32), 0)
        buf, _ = _to_s(ops, dv(x, dev), B * IH * IW, Cin, st)
        xs.append(buf)
        xm.append(dv(np.packbits(x.reshape(-1, Cin) != 0, axis=1, bitorder="little").reshape(-1), dev))
        keep.append(dv((rng.standard_normal((K, K, Cin, F)) * 0.1).astype(np.float32), dev))
        wfs.append(torch.full((wb // 2,), -1, dtype=torch.int16, device=dev))
        wds.append(torch.full((K * K * Cin * F * 2,), -1, dtype=torch.int16, device=dev))
        bs.append(dv((rng.standard_normal(F) * 0.1).astype(np.float32), dev))
        buf, _ = _to_s(ops, dv(rng.standard_normal((B, OH, OW, F)).astype(np.float32), dev), B * OH * OW, F, st)
        dzs.append(buf)
    ops.convs_prep_weights([t.data_ptr() for t in keep], [t.data_ptr() for t in wfs], [t.data_ptr() for t in wds],
                           False, K, K, Cin, F, st)
    out = {}
    try:
        for mode in (0, 1, 3):
            ops.convs_set_patch_kernels(mode)
            ys = [torch.full((B * OH * OW * F * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
            rm = [torch.full((B * OH * OW * F // 8,), 0xAA, dtype=torch.uint8, device=dev) for _ in range(G)]
            dxs = [torch.full((B * IH * IW * Cin * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
            ops.convs_fwd([t.data_ptr() for t in xs], [t.data_ptr() for t in wfs], [t.data_ptr() for t in bs],
                          [t.data_ptr() for t in ys], False, B, IH, IW, Cin, K, K, S, F, True, st,
                          rmasks=[rm[0].data_ptr(), 0, rm[2].data_ptr()])
            ops.convs_dgrad([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], [t.data_ptr() for t in xm],
                            [t.data_ptr() for t in dxs], B, IH, IW, Cin, K, K, S, F, st, xact_bits=True)
            dxn = [torch.full((B * IH * IW * Cin * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
            ops.convs_dgrad([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], [0] * G,
                            [t.data_ptr() for t in dxn], B, IH, IW, Cin, K, K, S, F, st)          # no ReLU mask
            dws = [torch.full((K, K, Cin, F), float("nan"), device=dev) for _ in range(G)]
            dbs = [torch.full((F,), float("nan"), device=dev) for _ in range(G)]
            wsb = ops.conv2d_wgrad_workspace_bytes(G, B, IH, IW, Cin, K, K, S, F)
            wsp = torch.empty(wsb // 4, device=dev)
            ops.convs_wgrad([t.data_ptr() for t in xs], [t.data_ptr() for t in dzs], [t.data_ptr() for t in dws],
                            [t.data_ptr() for t in dbs], wsp.data_ptr(), wsb, False, B, IH, IW, Cin, K, K, S, F, st)
            torch.cuda.synchronize()
            out[mode] = (ys, rm, dxs, dxn, dws, dbs)
    finally:
        ops.convs_set_patch_kernels(1)
    for mode in (1, 3):
        for g in range(G):
            assert torch.equal(out[0][0][g], out[mode][0][g]), f"forward differs, group {g}, mode {mode}"
            assert torch.equal(out[0][1][g], out[mode][1][g]), f"ReLU byte mask differs, group {g}, mode {mode}"
            assert torch.equal(out[0][2][g], out[mode][2][g]), f"data gradient differs, group {g}, mode {mode}"
            assert torch.equal(out[0][3][g], out[mode][3][g]), f"unmasked data gradient differs, group {g}, mode {mode}"
        assert bool((out[mode][1][1] == 0xAA).all()), "a NULL mask entry must not be written"
    # the patch weight gradient sums the pixels in another order (one slab per persistent block): same values to fp32
    # summation accuracy, and identical when repeated
    for g in range(G):
        for a_, b_ in ((out[0][4][g], out[1][4][g]), (out[0][5][g], out[1][5][g])):
            a_, b_ = a_.cpu().numpy(), b_.cpu().numpy()
            assert np.isfinite(b_).all()
            np.testing.assert_allclose(b_, a_, rtol=0, atol=2e-5 * max(1.0, float(np.abs(a_).max())))


@pytest.mark.parametrize("case", [(2, 3), (6, 3), (44, 2), (260, 4), (256, 1)])
def test_last_layer_forward_matches_im2col(dev, case):
    """pconv4_fwd_kernel (csrc/pconv4.h: the 15 x 15 x 128 -> 7 x 7 x 256 layer, K loop ordered by the parity class of the
    input pixel) against the im2col kernel (mode 8): the same bf16_3x products summed in another order -> equal to 2e-6 of
    the output scale or one step of the 16-bit S-format mantissa, not bit for bit; the ReLU byte masks describe the kernel's own output; a NULL mask entry is not
    written; identical when repeated.  One item per block, several items per block (260 images in 4 groups: 3 items), the
    XCD-grouped block order (groups dividing 8) and the plain one (3 groups)."""
    ops = _ops()
    B, G = case
    IH = IW = 15; Cin, F, K, S = 128, 256, 3, 2
    OH = OW = 7
    rng = np.random.default_rng(31)
    st = ops.current_stream()
    wb = ops.convs_weight_bytes(False, K, K, Cin, F)
    assert wb == 2 * 4 * 9 * Cin * F          # S-format image + the fragment-ordered image
    xs, wfs, bs, keep = [], [], [], []
    for g in range(G):
        x = np.maximum(rng.standard_normal((B, IH, IW, Cin)).astype(np.float32), 0)
        buf, _ = _to_s(ops, dv(x, dev), B * IH * IW, Cin, st)
        xs.append(buf)
        keep.append(dv((rng.standard_normal((K, K, Cin, F)) * 0.05).astype(np.float32), dev))
        wfs.append(torch.full((wb // 2,), -1, dtype=torch.int16, device=dev))
        bs.append(dv((rng.standard_normal(F) * 0.1).astype(np.float32), dev))
    ops.convs_prep_weights([t.data_ptr() for t in keep], [t.data_ptr() for t in wfs], None, False, K, K, Cin, F, st)
    out = {}
    try:
        for mode in (8, 1, "again"):
            ops.convs_set_patch_kernels(1 if mode == "again" else mode)
            ys = [torch.full((B * OH * OW * F * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
            rm = [torch.full((B * OH * OW * F // 8,), 0xAA, dtype=torch.uint8, device=dev) for _ in range(G)]
            ops.convs_fwd([t.data_ptr() for t in xs], [t.data_ptr() for t in wfs], [t.data_ptr() for t in bs],
                          [t.data_ptr() for t in ys], False, B, IH, IW, Cin, K, K, S, F, True, st,
                          rmasks=[t.data_ptr() if g != 1 else 0 for g, t in enumerate(rm)])
            torch.cuda.synchronize()
            out[mode] = (ys, rm)
    finally:
        ops.convs_set_patch_kernels(1)
    for g in range(G):
        ref = _from_s(ops, out[8][0][g], B * OH * OW, F, st)
        got = _from_s(ops, out[1][0][g], B * OH * OW, F, st)
        assert np.isfinite(got).all()
        # (the S format keeps 16 mantissa bits: a sum that differs in its last fp32 bits may round to the neighbouring value)
        np.testing.assert_allclose(got, ref, rtol=2.0 ** -14, atol=2e-6 * float(np.abs(ref).max()))
        assert torch.equal(out[1][0][g], out["again"][0][g]) and torch.equal(out[1][1][g], out["again"][1][g])
        if g == 1:
            assert bool((out[1][1][g] == 0xAA).all()), "a NULL mask entry must not be written"
        else:
            hi = out[1][0][g].view(B * OH * OW, 2, F)[:, 0, :].cpu().numpy().view(np.uint16)
            want = np.packbits((hi & 0x7fff) != 0, axis=1, bitorder="little").reshape(-1)
            assert np.array_equal(out[1][1][g].cpu().numpy(), want), "ReLU byte mask does not describe the output"
    # and against float64 on the first images
    nb = min(B, 2)
    x0 = _from_s(ops, xs[0], B * IH * IW, Cin, st).reshape(B, IH, IW, Cin)[:nb].astype(np.float64)
    w0 = keep[0].cpu().numpy().astype(np.float64)
    ref = np.zeros((nb, OH, OW, F))
    for ky in range(K):
        for kx in range(K):
            ref += np.einsum("bhwc,cf->bhwf", x0[:, ky:ky + 2 * OH - 1:2, kx:kx + 2 * OW - 1:2, :], w0[ky, kx])
    ref = np.maximum(ref + bs[0].cpu().numpy().astype(np.float64), 0)
    got = _from_s(ops, out[1][0][0], B * OH * OW, F, st).reshape(B, OH, OW, F)[:nb]
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * float(np.abs(ref).max()))


@pytest.mark.parametrize("masked", [True, False])
@pytest.mark.parametrize("case", [(3, 128, 128), (2, 100, 128), (1, 7, 16), (2, 34, 32), (40, 128, 128), (70, 36, 48)])
def test_dgrad_wgrad1_fused_matches_two_launches(dev, case, masked):
    """serl_convs_dgrad_wgrad1 (conv2's data gradient and conv1's weight + bias gradient in one kernel, dz1 kept on chip)
    against serl_convs_dgrad followed by serl_convs_wgrad(in_u8): the same dz1 bits enter the products, the fp32 sums run
    in another order (one slab per persistent block) -> 2e-5 of the largest entry, and bit-identical when repeated.  Full
    and ragged bands, an image smaller than one band, widths whose last column pair is half empty, batches large enough
    that a block walks several items, with and without the ReLU byte mask."""
    ops = _ops()
    B, IH1, IW1 = case
    C1, F1, F2, K, S, G = 9, 32, 64, 3, 2, 3
    assert ops.convs_dgrad_wgrad1_supported(IH1, IW1, C1, F1, F2)
    assert not ops.convs_dgrad_wgrad1_supported(IH1, IW1, 3, F1, F2) and not ops.convs_dgrad_wgrad1_supported(IH1, IW1 + 1, C1, F1, F2)
    rng = np.random.default_rng(23)
    st = ops.current_stream()
    IH, IW = (IH1 - K) // S + 1, (IW1 - K) // S + 1
    OH, OW = (IH - K) // S + 1, (IW - K) // S + 1
    frames, wds, keep, dzs, rms = [], [], [], [], []
    for g in range(G):
        frames.append(dv(rng.integers(0, 256, (B, IH1, IW1, C1), dtype=np.uint8), dev))
        keep.append(dv((rng.standard_normal((K, K, F1, F2)) * 0.1).astype(np.float32), dev))
        wds.append(torch.full((K * K * F1 * F2 * 2,), -1, dtype=torch.int16, device=dev))
        buf, _ = _to_s(ops, dv(rng.standard_normal((B, OH, OW, F2)).astype(np.float32), dev), B * OH * OW, F2, st)
        dzs.append(buf)
        act = rng.random((B * IH * IW, F1)) < 0.6
        rms.append(dv(np.packbits(act, axis=1, bitorder="little").reshape(-1), dev))
    wfs = [torch.empty(ops.convs_weight_bytes(False, K, K, F1, F2) // 2, dtype=torch.int16, device=dev) for _ in range(G)]
    ops.convs_prep_weights([t.data_ptr() for t in keep], [t.data_ptr() for t in wfs], [t.data_ptr() for t in wds],
                           False, K, K, F1, F2, st)
    masks = [t.data_ptr() for t in rms] if masked else None
    # reference: two launches, dz1 through memory
    dz1 = [torch.full((B * IH * IW * F1 * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
    ops.convs_dgrad([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], masks if masked else [0] * G,
                    [t.data_ptr() for t in dz1], B, IH, IW, F1, K, K, S, F2, st, xact_bits=masked)
    dw_ref = [torch.full((K, K, C1, F1), float("nan"), device=dev) for _ in range(G)]
    db_ref = [torch.full((F1,), float("nan"), device=dev) for _ in range(G)]
    wsb = ops.conv2d_wgrad_workspace_bytes(G, B, IH1, IW1, C1, K, K, S, F1)
    wsp = torch.empty(wsb // 4, device=dev)
    ops.convs_wgrad([t.data_ptr() for t in frames], [t.data_ptr() for t in dz1], [t.data_ptr() for t in dw_ref],
                    [t.data_ptr() for t in db_ref], wsp.data_ptr(), wsb, True, B, IH1, IW1, C1, K, K, S, F1, st)
    outs = []
    wsb2 = ops.convs_dgrad_wgrad1_workspace_bytes(G)
    for rep in range(2):
        dw = [torch.full((K, K, C1, F1), float("nan"), device=dev) for _ in range(G)]
        db = [torch.full((F1,), float("nan"), device=dev) for _ in range(G)]
        wsp2 = torch.full((wsb2 // 4,), float("nan"), device=dev)
        ops.convs_dgrad_wgrad1([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], masks, [t.data_ptr() for t in frames],
                               [t.data_ptr() for t in dw], [t.data_ptr() for t in db], wsp2.data_ptr(), wsb2, B, IH1, IW1,
                               C1, F1, F2, st)
        torch.cuda.synchronize()
        outs.append((dw, db))
    for g in range(G):
        assert torch.equal(outs[0][0][g], outs[1][0][g]) and torch.equal(outs[0][1][g], outs[1][1][g]), "not repeatable"
        for a_, b_ in ((dw_ref[g], outs[0][0][g]), (db_ref[g], outs[0][1][g])):
            a_, b_ = a_.cpu().numpy(), b_.cpu().numpy()
            assert np.isfinite(b_).all()
            np.testing.assert_allclose(b_, a_, rtol=0, atol=2e-5 * max(1.0, float(np.abs(a_).max())))
    # too small a workspace is refused, not overrun
    with pytest.raises(ValueError):
        ops.convs_dgrad_wgrad1([t.data_ptr() for t in dzs], [t.data_ptr() for t in wds], masks, [t.data_ptr() for t in frames],
                               [t.data_ptr() for t in outs[0][0]], [t.data_ptr() for t in outs[0][1]], wsp2.data_ptr(), 1024,
                               B, IH1, IW1, C1, F1, F2, st)


@pytest.mark.parametrize("mode", [1, 4])        # two wave sets half a period apart (default) / one set
@pytest.mark.parametrize("case", [(3, 128, 128, 9), (2, 100, 128, 9), (2, 128, 128, 3), (2, 31, 32, 8), (1, 7, 16, 3),
                                  (24, 128, 128, 9), (45, 31, 32, 8), (70, 20, 48, 3)])
def test_convs_fwd12_bit_identical_to_two_launches(dev, case, mode):
    """serl_convs_fwd12 (layers 1 + 2 of a no-gradient pass in one kernel, layer 1's output kept in LDS) against
    serl_convs_fwd(uint8) followed by serl_convs_fwd: same bits, same ReLU byte mask; full and ragged bands, 1..9 input
    channels, an image smaller than one band, and batches large enough that a persistent block walks several items
    (halo-row reuse, both wave sets, odd and even item counts)."""
    ops = _ops()
    ops.convs_set_patch_kernels(mode)
    try:
        _fwd12_case(ops, dev, case)
    finally:
        ops.convs_set_patch_kernels(1)


def _fwd12_case(ops, dev, case):
    B, IH1, IW1, C1 = case
    F1, F2, K, S, G = 32, 64, 3, 2, 3
    assert ops.convs_fwd12_supported(IH1, IW1, C1, F1, F2)
    assert not ops.convs_fwd12_supported(IH1, IW1, C1, 16, F2) and not ops.convs_fwd12_supported(IH1, IW1 + 1, C1, F1, F2)
    assert not ops.convs_fwd12_supported(IH1, 16, 2, F1, F2)       # uint8 runs (3 * C1 bytes) shorter than 8 are refused
    rng = np.random.default_rng(11)
    st = ops.current_stream()
    IH, IW = (IH1 - K) // S + 1, (IW1 - K) // S + 1
    OH, OW = (IH - K) // S + 1, (IW - K) // S + 1
    wb1, wb2 = ops.convs_weight_bytes(True, K, K, C1, F1), ops.convs_weight_bytes(False, K, K, F1, F2)
    xs, w1, w2, b1, b2, wf1, wf2 = [], [], [], [], [], [], []
    for g in range(G):
        xs.append(dv(rng.integers(0, 256, size=(B, IH1, IW1, C1), dtype=np.uint8), dev))
        w1.append(dv((rng.standard_normal((K, K, C1, F1)) * 0.3).astype(np.float32), dev))
        w2.append(dv((rng.standard_normal((K, K, F1, F2)) * 0.1).astype(np.float32), dev))
        b1.append(dv((rng.standard_normal(F1) * 0.1).astype(np.float32), dev))
        b2.append(dv((rng.standard_normal(F2) * 0.1).astype(np.float32), dev))
        wf1.append(torch.full((wb1 // 2,), -1, dtype=torch.int16, device=dev))
        wf2.append(torch.full((wb2 // 2,), -1, dtype=torch.int16, device=dev))
    p = lambda ts: [t.data_ptr() for t in ts]
    ops.convs_prep_weights(p(w1), p(wf1), None, True, K, K, C1, F1, st)
    ops.convs_prep_weights(p(w2), p(wf2), None, False, K, K, F1, F2, st)
    y1 = [torch.full((B * IH * IW * F1 * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
    ya = [torch.full((B * OH * OW * F2 * 2,), -1, dtype=torch.int16, device=dev) for _ in range(G)]
    yb = [torch.full_like(t, -2) for t in ya]
    ma = [torch.full((B * OH * OW * F2 // 8,), 0xAA, dtype=torch.uint8, device=dev) for _ in range(G)]
    mb = [torch.full_like(t, 0x55) for t in ma]
    ops.convs_fwd(p(xs), p(wf1), p(b1), p(y1), True, B, IH1, IW1, C1, K, K, S, F1, True, st)
    ops.convs_fwd(p(y1), p(wf2), p(b2), p(ya), False, B, IH, IW, F1, K, K, S, F2, True, st,
                  rmasks=[ma[0].data_ptr(), 0, ma[2].data_ptr()])
    ops.convs_fwd12(p(xs), p(wf1), p(b1), p(wf2), p(b2), p(yb), B, IH1, IW1, C1, F1, F2, True, st,
                    rmasks=[mb[0].data_ptr(), 0, mb[2].data_ptr()])
    torch.cuda.synchronize()
    for g in range(G):
        assert torch.equal(ya[g], yb[g]), f"fused layer-2 output differs, group {g}"
    assert torch.equal(ma[0], mb[0]) and torch.equal(ma[2], mb[2]), "ReLU byte mask differs"
    assert bool((mb[1] == 0x55).all()), "a NULL mask entry must not be written"
    got = _from_s(ops, yb[0], B * OH * OW, F2, st)
    assert np.isfinite(got).all() and (got > 0).mean() > 0.05          # not trivially dead
    # ---- the variant that keeps layer 1's output for the backward (two-set kernel): groups 0 and 2 get it, group 0 its
    # ReLU byte mask too; same bits as serl_convs_fwd(in_u8 = 1) on the rows layer 2 reads, nothing else touched
    if not ops.convs_fwd12_keep_supported(IH1, IW1, C1, F1, F2):
        return
    m1ref = torch.full((B * IH * IW * F1 // 8,), 0xAA, dtype=torch.uint8, device=dev)
    y1r = [torch.full_like(t, -1) for t in y1]
    ops.convs_fwd(p(xs), p(wf1), p(b1), p(y1r), True, B, IH1, IW1, C1, K, K, S, F1, True, st, rmasks=[m1ref.data_ptr(), 0, 0])
    yc = [torch.full_like(t, -2) for t in ya]
    mc = [torch.full_like(t, 0x55) for t in ma]
    y1k = [torch.full_like(t, -3) for t in y1]
    m1k = torch.full_like(m1ref, 0x33)
    ops.convs_fwd12_keep(p(xs), p(wf1), p(b1), p(wf2), p(b2), p(yc), B, IH1, IW1, C1, F1, F2, True, st,
                         rmasks=[mc[0].data_ptr(), 0, mc[2].data_ptr()], y1s=[y1k[0].data_ptr(), 0, y1k[2].data_ptr()],
                         rmask1s=[m1k.data_ptr(), 0, 0])
    torch.cuda.synchronize()
    for g in range(G):
        assert torch.equal(ya[g], yc[g]), f"keep variant: layer-2 output differs, group {g}"
    assert torch.equal(ma[0], mc[0]) and torch.equal(ma[2], mc[2])
    nr = 2 * OH + 1                                                     # layer-1 rows that layer 2 reads
    for g in (0, 2):
        a_, b_ = y1r[g].view(B, IH, -1), y1k[g].view(B, IH, -1)
        assert torch.equal(a_[:, :nr], b_[:, :nr]), f"kept layer-1 output differs, group {g}"
        assert bool((b_[:, nr:] == -3).all()), "rows layer 2 does not read must stay untouched"
    assert bool((y1k[1] == -3).all()), "a NULL layer-1 entry must not be written"
    a_, b_ = m1ref.view(B, IH, -1), m1k.view(B, IH, -1)
    assert torch.equal(a_[:, :nr], b_[:, :nr]), "kept layer-1 ReLU byte mask differs"
    assert bool((b_[:, nr:] == 0x33).all())


# ---------------------------------------------------------------------------------- igemm: dense
@pytest.mark.parametrize("dims", [(3, 70, 83, 45), (2, 64, 96, 128)])   # ragged (scalar loads) / aligned (float4 loads)
def test_gemm_variants(dev, prec, dims):
    ops = _ops()
    tol = 2e-5 if prec == "highest" else 1e-4
    rng = np.random.default_rng(2)
    st = ops.current_stream()
    E, B, K, N = dims
    x = rng.standard_normal((B, K)).astype(np.float32)
    w = rng.standard_normal((E, K, N)).astype(np.float32)
    b = rng.standard_normal((E, N)).astype(np.float32)
    X, W, Bi = dv(x, dev), dv(w, dev), dv(b, dev)
    Y = torch.full((E, B, N), float("nan"), device=dev)
    # fwd, shared input, ensemble weights, bias + relu
    ops.gemm(X.data_ptr(), W.data_ptr(), Y.data_ptr(), Bi.data_ptr(), E, B, N, K, a_zs=0, a_ms=K, a_ks=1,
             b_zs=K * N, b_ks=N, b_ns=1, c_zs=B * N, c_ms=N, bias_zs=N, relu=True, stream=st)
    ref = np.maximum(np.einsum("bk,ekn->ebn", x.astype(np.float64), w.astype(np.float64)) + b[:, None, :], 0)
    torch.cuda.synchronize()
    close(Y.cpu().numpy(), ref, rtol=tol, what="gemm fwd")
    # dX = sum_e dY[e] @ W[e]^T  (segmented K over (e, n)); A k-contig, B k-contig
    dy = rng.standard_normal((E, B, N)).astype(np.float32)
    DY = dv(dy, dev)
    DX = torch.full((B, K), float("nan"), device=dev)
    ops.gemm(DY.data_ptr(), W.data_ptr(), DX.data_ptr(), None, 1, B, K, E * N, seglen=N, a_ss=B * N, a_ms=N, a_ks=1,
             b_ss=K * N, b_ks=1, b_ns=N, c_ms=K, a_kcontig=True, b_ncontig=False, stream=st)
    torch.cuda.synchronize()
    close(DX.cpu().numpy(), np.einsum("ebn,ekn->bk", dy.astype(np.float64), w.astype(np.float64)), rtol=tol, what="gemm dX")
    # dW[e] = X^T @ dY[e]; A m-contig
    DW = torch.full((E, K, N), float("nan"), device=dev)
    ops.gemm(X.data_ptr(), DY.data_ptr(), DW.data_ptr(), None, E, K, N, B, a_zs=0, a_ms=1, a_ks=K,
             b_zs=B * N, b_ks=N, b_ns=1, c_zs=K * N, c_ms=N, a_kcontig=False, b_ncontig=True, stream=st)
    torch.cuda.synchronize()
    close(DW.cpu().numpy(), np.einsum("bk,ebn->ekn", x.astype(np.float64), dy.astype(np.float64)), rtol=tol, what="gemm dW")
    # split-K slabs
    nsplit = 3
    SL = torch.full((nsplit, B, N), float("nan"), device=dev)
    ops.gemm(X.data_ptr(), W.data_ptr(), SL.data_ptr(), None, 1, B, N, K, a_ms=K, a_ks=1, b_ks=N, b_ns=1, c_ms=N,
             nsplit=nsplit, c_split_stride=B * N, stream=st)
    torch.cuda.synchronize()
    close(SL.cpu().numpy().sum(0), x.astype(np.float64) @ w[0].astype(np.float64), rtol=tol, what="gemm split-K")
    # big-tile path (M, N >= 512) with accumulate
    M2, N2, K2 = 520, 530, 40
    a2 = rng.standard_normal((M2, K2)).astype(np.float32); b2 = rng.standard_normal((K2, N2)).astype(np.float32)
    c0 = rng.standard_normal((M2, N2)).astype(np.float32)
    A2, B2, C2 = dv(a2, dev), dv(b2, dev), dv(c0, dev)
    ops.gemm(A2.data_ptr(), B2.data_ptr(), C2.data_ptr(), None, 1, M2, N2, K2, a_ms=K2, a_ks=1, b_ks=N2, b_ns=1,
             c_ms=N2, accumulate=True, stream=st)
    torch.cuda.synchronize()
    close(C2.cpu().numpy(), c0 + a2.astype(np.float64) @ b2.astype(np.float64), rtol=tol, what="gemm 128 tile")


# ---------------------------------------------------------------------------------- LN + activation
@pytest.mark.parametrize("use_ln,act,N", [(True, 1, 256), (True, 2, 512), (True, 2, 64), (False, 1, 256), (True, 1, 50)])
def test_ln_act_fwd_bwd(dev, use_ln, act, N):
    ops = _ops()
    rng = np.random.default_rng(3)
    st = ops.current_stream()
    E, B, nslab = 3, 37, 2
    R = E * B
    pre = rng.standard_normal((nslab, R, N)).astype(np.float32)
    bias = rng.standard_normal((E, N)).astype(np.float32)
    gam = (1 + 0.1 * rng.standard_normal((E, N))).astype(np.float32)
    bet = (0.1 * rng.standard_normal((E, N))).astype(np.float32)
    P, Bi, Ga, Be = dv(pre, dev), dv(bias, dev), dv(gam, dev), dv(bet, dev)
    ldy = N + 7
    Y = torch.zeros((R, ldy), device=dev)
    XH = torch.zeros((R, N), device=dev); RS = torch.zeros((R,), device=dev)
    ops.ln_act_fwd(P.data_ptr(), nslab, R * N, N, Bi.data_ptr(), Ga.data_ptr() if use_ln else None,
                   Be.data_ptr() if use_ln else None, N, Y.data_ptr(), ldy, XH.data_ptr() if use_ln else None,
                   RS.data_ptr() if use_ln else None, R, N, B, use_ln, act, st)
    zt = torch.tensor(pre.astype(np.float64).sum(0).reshape(E, B, N) + bias[:, None, :], requires_grad=True)
    h = zt
    if use_ln:
        mean = h.mean(-1, keepdim=True)
        var = (h * h).mean(-1, keepdim=True) - mean * mean
        h = (h - mean) * torch.rsqrt(var + 1e-6) * torch.tensor(gam, dtype=torch.float64)[:, None, :] \
            + torch.tensor(bet, dtype=torch.float64)[:, None, :]
    yref = torch.relu(h) if act == 1 else torch.tanh(h)
    torch.cuda.synchronize()
    close(Y.cpu().numpy()[:, :N], yref.detach().numpy().reshape(R, N), what="ln fwd")
    dy = rng.standard_normal((R, N)).astype(np.float32)
    DY = dv(dy, dev)
    DZ = torch.zeros((R, N), device=dev); G = torch.zeros((R, N), device=dev)
    ops.ln_act_bwd(DY.data_ptr(), N, None, None, 0, Y.data_ptr(), ldy, XH.data_ptr() if use_ln else None,
                   RS.data_ptr() if use_ln else None, Ga.data_ptr() if use_ln else None, N, DZ.data_ptr(),
                   G.data_ptr() if use_ln else None, R, N, B, use_ln, act, st)
    (yref * torch.tensor(dy.astype(np.float64)).reshape(E, B, N)).sum().backward()
    torch.cuda.synchronize()
    close(DZ.cpu().numpy(), zt.grad.numpy().reshape(R, N), rtol=1e-4, what="ln bwd dz")
    if use_ln:
        # column sums: dbeta = sum g, dgamma = sum g*xhat, dbias = sum dz
        dbe = torch.zeros((E, N), device=dev); dga = torch.zeros((E, N), device=dev); dbi = torch.zeros((E, N), device=dev)
        ops.colsum(G.data_ptr(), dbe.data_ptr(), E, B, N, N, N, st, x1=DZ.data_ptr(), out1=dbi.data_ptr(),
                   x2=G.data_ptr(), mul2=XH.data_ptr(), out2=dga.data_ptr())
        torch.cuda.synchronize()
        g = G.cpu().numpy().astype(np.float64).reshape(E, B, N)
        xh = XH.cpu().numpy().astype(np.float64).reshape(E, B, N)
        close(dbe.cpu().numpy(), g.sum(1), what="dbeta")
        close(dga.cpu().numpy(), (g * xh).sum(1), what="dgamma")
        close(dbi.cpu().numpy(), zt.grad.numpy().sum(1), rtol=1e-4, what="dbias")


def test_rowdot_and_rank1_bwd(dev):
    ops = _ops()
    rng = np.random.default_rng(4)
    st = ops.current_stream()
    E, B, N = 4, 33, 256
    h = np.maximum(rng.standard_normal((E * B, N)), 0).astype(np.float32)
    w = rng.standard_normal((E, N)).astype(np.float32); b = rng.standard_normal(E).astype(np.float32)
    H, W, Bi = dv(h, dev), dv(w, dev), dv(b, dev)
    Q = torch.zeros(E * B, device=dev)
    ops.rowdot(H.data_ptr(), W.data_ptr(), N, Bi.data_ptr(), 1, Q.data_ptr(), E * B, N, B, st)
    torch.cuda.synchronize()
    ref = np.einsum("ebn,en->eb", h.reshape(E, B, N).astype(np.float64), w.astype(np.float64)) + b[:, None]
    close(Q.cpu().numpy().reshape(E, B), ref, what="rowdot")
    # rank-1 backward form without LN: dz = dq[r] * w[e][n] * relu'(h)
    dq = rng.standard_normal(E * B).astype(np.float32)
    DQ = dv(dq, dev); DZ = torch.zeros((E * B, N), device=dev)
    ops.ln_act_bwd(None, 0, DQ.data_ptr(), W.data_ptr(), N, H.data_ptr(), N, None, None, None, 0, DZ.data_ptr(), None,
                   E * B, N, B, False, 1, st)
    torch.cuda.synchronize()
    ref = dq.reshape(E, B, 1) * w[:, None, :] * (h.reshape(E, B, N) > 0)
    close(DZ.cpu().numpy().reshape(E, B, N), ref, what="rank1 bwd")
    # weighted column sum: dw3[e][n] = sum_b dq[e,b] h[e,b,n]
    out = torch.zeros((E, N), device=dev)
    ops.colsum(H.data_ptr(), out.data_ptr(), E, B, N, N, N, st, w=DQ.data_ptr())
    torch.cuda.synchronize()
    close(out.cpu().numpy(), np.einsum("eb,ebn->en", dq.reshape(E, B).astype(np.float64),
                                       h.reshape(E, B, N).astype(np.float64)), what="weighted colsum")


# ---------------------------------------------------------------------------------- SLE
@pytest.mark.parametrize("B,HW,C", [(5, 49, 256), (3, 30, 32), (2, 7, 70), (67, 49, 40)])   # B >= 64: sliced kernels
def test_sle(dev, B, HW, C):
    _sle_case(dev, B, HW, C, False)


@pytest.mark.parametrize("B,HW,C", [(5, 49, 256), (3, 30, 32), (70, 49, 72), (66, 17, 32)])   # B >= 64: sliced kernel
def test_sle_split_features(dev, B, HW, C):
    _sle_case(dev, B, HW, C, True)


def _sle_case(dev, B, HW, C, split):
    ops = _ops()
    rng = np.random.default_rng(5)
    st = ops.current_stream()
    feat = np.maximum(rng.standard_normal((B, HW, C)), 0).astype(np.float32)
    ker = rng.standard_normal((HW, C, 8)).astype(np.float32)
    Fe, Ke = dv(feat, dev), dv(ker, dev)
    if split:       # S-format features: the reference sees the values they hold; dfeat comes back in S format
        Fe, featv = _to_s(ops, Fe, B * HW, C, st)
        feat = featv.reshape(B, HW, C)
    ldo = 2 * C * 8
    out = torch.zeros((B, ldo), device=dev)
    ops.sle_fwd(Fe.data_ptr(), Ke.data_ptr(), out.data_ptr() + C * 8 * 4, ldo, B, HW, C, 8, st, feat_split=split)
    torch.cuda.synchronize()
    ref = np.einsum("bhc,hcf->bcf", feat.astype(np.float64), ker.astype(np.float64)).reshape(B, -1)
    close(out.cpu().numpy()[:, C * 8:], ref, what="sle fwd")
    assert float(out[:, :C * 8].abs().max()) == 0.0
    dout = rng.standard_normal((B, ldo)).astype(np.float32)
    DO = dv(dout, dev)
    dfeat = torch.zeros((B, HW, C), device=dev); dker = torch.zeros((HW, C, 8), device=dev)
    ops.sle_bwd(DO.data_ptr() + C * 8 * 4, ldo, Ke.data_ptr(), Fe.data_ptr(), dfeat.data_ptr(), dker.data_ptr(), B, HW,
                C, 8, st, feat_split=split)
    torch.cuda.synchronize()
    d = dout[:, C * 8:].reshape(B, C, 8).astype(np.float64)
    got = _from_s(ops, dfeat, B * HW, C, st).reshape(B, HW, C) if split else dfeat.cpu().numpy()
    close(got, np.einsum("bcf,hcf->bhc", d, ker.astype(np.float64)) * (feat > 0), what="sle dfeat")
    close(dker.cpu().numpy(), np.einsum("bhc,bcf->hcf", feat.astype(np.float64), d), what="sle dker")


# ---------------------------------------------------------------------------------- RNG, crop, gather
def test_crop_offsets_bit_exact(dev):
    from oracle import learner as ol, prng
    ops = _ops()
    st = ops.current_stream()
    key = prng.split(prng.prng_key(42))[1]
    n = 777
    out = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    ops.crop_offsets(key, n, 0, n, 4, out.data_ptr(), st)
    torch.cuda.synchronize()
    ref = ol.crop_offsets(key, n)           # scalar restatement
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(ol.crop_offsets_fast(key, n), ref)
    # a shard of a larger global batch reproduces the global stream
    sh = torch.zeros((100, 2), dtype=torch.int32, device=dev)
    ops.crop_offsets(key, 100, 300, n, 4, sh.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(sh.cpu().numpy(), ref[300:400])
    assert ref.min() == 0 and ref.max() == 8


@pytest.mark.parametrize("S", [3, 1])
def test_gather_crop_bit_exact(dev, S):
    from oracle import learner as ol
    ops = _ops()
    st = ops.current_stream()
    rng = np.random.default_rng(6)
    cap, H, W, n = 64, 128, 128, 24
    frames = rng.integers(0, 256, size=(cap, H, W, 3), dtype=np.uint8)
    indx = rng.integers(S, cap, size=n).astype(np.int64)
    indx[0] = S            # lowest valid
    indx[1] = cap - 1
    indx[2] = 0            # demo-style index below S: wraps like numpy negative indexing (quirk C.7)
    offs = rng.integers(0, 9, size=(2, n, 2)).astype(np.int32)
    offs[0, 3] = (0, 0); offs[1, 3] = (8, 8); offs[0, 4] = (0, 8); offs[1, 4] = (8, 0); offs[0, 5] = (4, 4)
    win = np.lib.stride_tricks.sliding_window_view(frames, S + 1, axis=0)[indx - S]     # (n,H,W,3,S+1)
    ref_obs = ol.random_crop_batch(win[..., :-1], offs[0])
    ref_nxt = ol.random_crop_batch(win[..., 1:], offs[1])
    Fr, Ix, Of = dv(frames, dev), dv(indx, dev), dv(offs, dev)
    o1 = torch.zeros((n, H, W, 3 * S), dtype=torch.uint8, device=dev); o2 = torch.zeros_like(o1)
    ops.gather_crop_frames(Fr.data_ptr(), cap, Ix.data_ptr(), S + 1, 0, Of[0].data_ptr(), o1.data_ptr(), n, H, W, S, 4, st)
    ops.gather_crop_frames(Fr.data_ptr(), cap, Ix.data_ptr(), S + 1, 1, Of[1].data_ptr(), o2.data_ptr(), n, H, W, S, 4, st)
    torch.cuda.synchronize()
    assert np.array_equal(o1.cpu().numpy().reshape(n, H, W, 3, S), ref_obs)
    assert np.array_equal(o2.cpu().numpy().reshape(n, H, W, 3, S), ref_nxt)
    # packed source (S+1 interleaved frames)
    Pk = dv(np.ascontiguousarray(win), dev)
    o1.zero_(); o2.zero_()
    ops.gather_crop_packed(Pk.data_ptr(), S + 1, 0, Of[0].data_ptr(), o1.data_ptr(), n, H, W, S, 4, st)
    ops.gather_crop_packed(Pk.data_ptr(), S + 1, 1, Of[1].data_ptr(), o2.data_ptr(), n, H, W, S, 4, st)
    torch.cuda.synchronize()
    assert np.array_equal(o1.cpu().numpy().reshape(n, H, W, 3, S), ref_obs)
    assert np.array_equal(o2.cpu().numpy().reshape(n, H, W, 3, S), ref_nxt)
    # unpacked source (already split obs / next_obs, T = S)
    Un = dv(np.ascontiguousarray(win[..., 1:]), dev)
    o2.zero_()
    ops.gather_crop_packed(Un.data_ptr(), S, 0, Of[1].data_ptr(), o2.data_ptr(), n, H, W, S, 4, st)
    torch.cuda.synchronize()
    assert np.array_equal(o2.cpu().numpy().reshape(n, H, W, 3, S), ref_nxt)
    # plain window gather
    wout = torch.zeros((n, H, W, 3, S + 1), dtype=torch.uint8, device=dev)
    ops.gather_window(Fr.data_ptr(), cap, Ix.data_ptr(), wout.data_ptr(), n, H * W * 3, S, st)
    torch.cuda.synchronize()
    assert np.array_equal(wout.cpu().numpy(), win)


@pytest.mark.parametrize("hw", [(84, 84), (100, 100), (20, 37), (128, 96)])
def test_gather_crop_any_image_size(dev, hw):
    """The reference's random_crop accepts any H x W (serl/utils/augmentations.py:5-11): frame rows whose byte length
    is not a multiple of 16 (84x84, 100x100: the usual DrQ sizes), heights that are not a multiple of the 8-row tile."""
    from oracle import learner as ol
    ops = _ops()
    st = ops.current_stream()
    H, W = hw
    S, cap, n = 3, 20, 9
    rng = np.random.default_rng(H * 1000 + W)
    frames = rng.integers(0, 256, size=(cap, H, W, 3), dtype=np.uint8)
    indx = rng.integers(S, cap, size=n).astype(np.int64)
    offs = rng.integers(0, 9, size=(n, 2)).astype(np.int32)
    offs[0] = (0, 0); offs[1] = (8, 8)
    win = np.lib.stride_tricks.sliding_window_view(frames, S + 1, axis=0)[indx - S]
    ref = ol.random_crop_batch(win[..., 1:], offs)
    Fr, Ix, Of = dv(frames, dev), dv(indx, dev), dv(offs, dev)
    o = torch.zeros((n, H, W, 3 * S), dtype=torch.uint8, device=dev)
    ops.gather_crop_frames(Fr.data_ptr(), cap, Ix.data_ptr(), S + 1, 1, Of.data_ptr(), o.data_ptr(), n, H, W, S, 4, st)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy().reshape(n, H, W, 3, S), ref)
    Pk = dv(np.ascontiguousarray(win), dev)
    o.zero_()
    ops.gather_crop_packed(Pk.data_ptr(), S + 1, 1, Of.data_ptr(), o.data_ptr(), n, H, W, S, 4, st)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy().reshape(n, H, W, 3, S), ref)


def test_augmentations_module_bit_exact(dev):
    """serl_amd.utils.augmentations.{random_crop, batched_random_crop} == serl/utils/augmentations.py:5-11,44-48 as the
    oracle restates them (threefry randint offsets, edge pad + slice), uint8 bit-exact."""
    from oracle import learner as ol, prng
    from serl_amd.frozen_dict import FrozenDict
    from serl_amd.utils.augmentations import batched_random_crop, random_crop
    rng = np.random.default_rng(2)
    n, H, W, S = 7, 84, 84, 2
    imgs = rng.integers(0, 256, size=(n, H, W, 3, S), dtype=np.uint8)
    key = prng.prng_key(11)
    obs = FrozenDict({"pixels": imgs, "state": np.zeros((n, 3), np.float32)})
    out = batched_random_crop(key, obs, "pixels", padding=4)
    want = ol.random_crop_batch(imgs, ol.crop_offsets_fast(key, n))
    assert np.array_equal(out["pixels"].cpu().numpy(), want) and out["state"] is obs["state"]
    one = random_crop(key, imgs[0], 4)
    off = prng.randint(key, (2,), 0, 9)
    assert np.array_equal(one.cpu().numpy(), ol.random_crop_batch(imgs[:1], off[None])[0])


def test_normal_stream_matches_jax_documentation(dev):
    """The eps output of serl_tanh_gaussian_fwd (threefry -> uniform -> erf_inv on the GPU; the stream behind
    serl/distributions/tanh_normal.py:53 as sampled at sac_learner.py:126,169) on the values the public JAX documentation
    prints for jax.random.normal (tests/golden/prng_kat.json: jax_documentation).  1 ulp-level tolerance: the device
    log1p / sqrt are not the host's."""
    import json
    import os
    from serl_amd import prng as nprng
    ops = _ops()
    st = ops.current_stream()
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prng_kat.json")))["jax_documentation"]
    k42 = nprng.PRNGKey(42)
    for key, B, A, want in ((nprng.PRNGKey(0), 2, 5, d["normal_prngkey0_10"]),
                            (nprng.PRNGKey(0), 10, 1, d["normal_prngkey0_10"]),
                            (k42, 1, 1, [d["normal_prngkey42_scalar"]]),
                            (nprng.split(k42)[1], 1, 1, [d["normal_subkey_of_split_prngkey42_scalar"]])):
        ms = torch.zeros((B, 2 * A), device=dev)
        act = torch.zeros((B, A), device=dev); ep = torch.full((B, A), float("nan"), device=dev)
        ops.tanh_gaussian_fwd(ms.data_ptr(), ms.data_ptr() + 4 * A, 2 * A, act.data_ptr(), A, None, ep.data_ptr(), None,
                              B, A, key, 0, B, False, st)
        torch.cuda.synchronize()
        np.testing.assert_allclose(ep.cpu().numpy().reshape(-1), np.asarray(want, np.float32), rtol=3e-7, atol=0)
        # mean 0, log-std 0: the action is tanh(eps)
        np.testing.assert_allclose(act.cpu().numpy().reshape(-1), np.tanh(np.asarray(want, np.float64)), rtol=2e-6)


def test_tanh_gaussian(dev):
    from oracle import learner as ol, prng
    ops = _ops()
    st = ops.current_stream()
    rng = np.random.default_rng(7)
    B, A = 256, 7
    ms = rng.standard_normal((B, 2 * A)).astype(np.float32)
    ms[:, A:] *= 3
    ms[0, A] = 5.0; ms[1, A] = -25.0      # clipped log-stds
    MS = dv(ms, dev)
    act = torch.zeros((B, A + 3), device=dev); xp = torch.zeros((B, A), device=dev)
    ep = torch.zeros((B, A), device=dev); lp = torch.zeros((B,), device=dev)
    key = prng.split(prng.prng_key(5))[0]
    ops.tanh_gaussian_fwd(MS.data_ptr(), MS.data_ptr() + 4 * A, 2 * A, act.data_ptr(), A + 3, xp.data_ptr(),
                          ep.data_ptr(), lp.data_ptr(), B, A, key, 0, B, False, st)
    torch.cuda.synchronize()
    eps_ref = prng.normal(key, (B, A))
    close(ep.cpu().numpy(), eps_ref, rtol=5e-7, what="normal stream")
    mean = torch.tensor(ms[:, :A], dtype=torch.float64, requires_grad=True)
    raw = torch.tensor(ms[:, A:], dtype=torch.float64, requires_grad=True)
    lstd = torch.clamp(raw, -20.0, 2.0)
    x = mean + torch.exp(lstd) * torch.tensor(eps_ref, dtype=torch.float64)
    a = torch.tanh(x)
    logp = ol.tanh_normal_log_prob(mean, lstd, x)
    close(act.cpu().numpy()[:, :A], a.detach().numpy(), what="actions")
    # log-prob: the reference evaluates (x - mean)/sigma in float32 (cancels to 0 when sigma = e^-20),
    # so compare with the float32 oracle, not the float64 one
    m32 = torch.tensor(ms[:, :A]); l32 = torch.clamp(torch.tensor(ms[:, A:]), -20.0, 2.0)
    x32 = m32 + torch.exp(l32) * torch.tensor(eps_ref)
    close(lp.cpu().numpy(), ol.tanh_normal_log_prob(m32, l32, x32).numpy(), rtol=1e-5, what="log prob")
    # shard == slice of the global stream
    sh = torch.zeros((64, A), device=dev); a2 = torch.zeros((64, A), device=dev)
    ops.tanh_gaussian_fwd(MS.data_ptr() + 64 * 2 * A * 4, MS.data_ptr() + 64 * 2 * A * 4 + 4 * A, 2 * A, a2.data_ptr(),
                          A, None, sh.data_ptr(), None, 64, A, key, 64, B, False, st)
    torch.cuda.synchronize()
    assert np.array_equal(sh.cpu().numpy(), ep.cpu().numpy()[64:128])
    # backward: L = mean(alpha*logp) + sum(a * c)  with da = c
    alpha = 0.37
    c = rng.standard_normal((B, A)).astype(np.float32)
    loss = (alpha * logp).mean() + (a * torch.tensor(c, dtype=torch.float64)).sum()
    loss.backward()
    LT = dv(np.array([math.log(alpha)], dtype=np.float32), dev)
    DA = dv(c, dev)
    dms = torch.zeros((B, 2 * A), device=dev)
    ops.tanh_gaussian_bwd(act.data_ptr(), A + 3, ep.data_ptr(), MS.data_ptr() + 4 * A, 2 * A, DA.data_ptr(), A,
                          LT.data_ptr(), 1.0 / B, dms.data_ptr(), dms.data_ptr() + 4 * A, 2 * A, B, A, st)
    torch.cuda.synchronize()
    close(dms.cpu().numpy()[:, :A], mean.grad.numpy(), rtol=1e-4, what="dmean")
    close(dms.cpu().numpy()[:, A:], raw.grad.numpy(), rtol=1e-4, what="dlogstd")
    # deterministic mode = tanh(mean)
    ops.tanh_gaussian_fwd(MS.data_ptr(), MS.data_ptr() + 4 * A, 2 * A, act.data_ptr(), A + 3, None, None, None, B, A,
                          key, 0, B, True, st)
    torch.cuda.synchronize()
    close(act.cpu().numpy()[:, :A], np.tanh(ms[:, :A].astype(np.float64)), what="mode")


def test_td_loss_actor_loss_adam_temperature(dev):
    from oracle import learner as ol
    ops = _ops()
    st = ops.current_stream()
    rng = np.random.default_rng(8)
    E, Em, B = 10, 2, 256
    nq = rng.standard_normal((Em, B)).astype(np.float32); q = rng.standard_normal((E, B)).astype(np.float32)
    rew = rng.standard_normal(B).astype(np.float32); msk = (rng.random(B) < 0.9).astype(np.float32)
    nlp = rng.standard_normal(B).astype(np.float32)
    lt = np.array([math.log(0.2)], dtype=np.float32)
    NQ, Q, R, M, NLP, LT = (dv(a, dev) for a in (nq, q, rew, msk, nlp, lt))
    tgt = torch.zeros(B, device=dev); dq = torch.zeros((E, B), device=dev); info = torch.zeros(4, device=dev)
    for backup in (False, True):
        ops.td_loss(NQ.data_ptr(), Em, R.data_ptr(), M.data_ptr(), NLP.data_ptr() if backup else None, LT.data_ptr(),
                    0.99, Q.data_ptr(), E, B, 1.0 / (E * B), tgt.data_ptr(), dq.data_ptr(), info.data_ptr(), st)
        torch.cuda.synchronize()
        t = rew + 0.99 * msk * nq.min(0)
        if backup:
            t = t - 0.99 * msk * 0.2 * nlp
        close(tgt.cpu().numpy(), t, what="target")
        close(dq.cpu().numpy(), 2 * (q - t) / (E * B), what="dq")
        close(info.cpu().numpy()[0] / (E * B), ((q - t) ** 2).mean(), what="loss")
        close(info.cpu().numpy()[1] / (E * B), q.mean(), rtol=1e-4, what="q mean")
    ops.actor_loss(Q.data_ptr(), E, B, NLP.data_ptr(), LT.data_ptr(), info.data_ptr(), st)
    torch.cuda.synchronize()
    close(info.cpu().numpy()[0] / B, (nlp * 0.2 - q.mean(0)).mean(), rtol=1e-4, what="actor loss")
    close(info.cpu().numpy()[1] / B, nlp.mean(), rtol=1e-4, what="sum logp")
    # Adam + Polyak against the oracle's optax restatement, 3 steps, odd length (tail path)
    n = 1027
    p0 = rng.standard_normal(n).astype(np.float32); t0 = rng.standard_normal(n).astype(np.float32)
    ts = dict(params=dict(w=p0.copy()), mu=dict(w=np.zeros(n, np.float32)), nu=dict(w=np.zeros(n, np.float32)), count=0)
    tgt_ref = dict(w=t0.copy())
    P, Mo, V, T = dv(p0, dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev), dv(t0, dev)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        ts = ol.adam_step(ts, dict(w=g), 3e-4)
        tgt_ref = ol.polyak(ts["params"], tgt_ref, 0.005)
        bc1 = np.float32(1) - np.float32(0.9) ** np.float32(step)
        bc2 = np.float32(1) - np.float32(0.999) ** np.float32(step)
        ops.adam_polyak(P.data_ptr(), Mo.data_ptr(), V.data_ptr(), dv(g, dev).data_ptr(), T.data_ptr(), n, 3e-4, 0.9,
                        0.999, 1e-8, bc1, bc2, 0.005, 1.0, st)
        torch.cuda.synchronize()
    close(P.cpu().numpy(), ts["params"]["w"], rtol=1e-6, what="adam params")
    close(T.cpu().numpy(), tgt_ref["w"], rtol=1e-6, what="polyak")
    # temperature
    state = dv(np.array([math.log(0.1), 0, 0], dtype=np.float32), dev)
    sl = dv(np.array([-3.0 * B], dtype=np.float32), dev)     # entropy = 3.0
    out = torch.zeros(3, device=dev)
    ops.temperature_update(state.data_ptr(), sl.data_ptr(), 1.0 / B, -3.5, 3e-4, 0.9, 0.999, 1e-8, 0.1, 0.001, out.data_ptr(), st)
    torch.cuda.synchronize()
    st0 = dict(rng=None, temp=dict(params=dict(log_temp=np.float32(math.log(0.1))), mu=dict(log_temp=np.float32(0)),
                                   nu=dict(log_temp=np.float32(0)), count=0))
    new, tinfo = ol.update_temperature(st0, dict(target_entropy=-3.5, temp_lr=3e-4), 3.0)
    close(out.cpu().numpy()[:2], [tinfo["temperature"], tinfo["temperature_loss"]], rtol=1e-5, what="temp info")
    close(state.cpu().numpy()[0], new["temp"]["params"]["log_temp"], rtol=1e-6, what="log_temp")


def test_error_channel(dev):
    ops = _ops()
    with pytest.raises(ValueError):
        ops.fill(0, 1.0, 10, 0)
    with pytest.raises(ValueError):
        ops.conv2d_fwd([1] * 9, [1] * 9, [1] * 9, [1] * 9, False, 1, 8, 8, 3, 3, 3, 1, 4, True, 0)


def test_prep_weights_multi_equals_single(dev):
    """serl_convs_prep_weights_multi (all layers x groups of a pass in one launch) writes the same wf / wd bits as
    one serl_convs_prep_weights call per layer."""
    ops = _ops()
    st = ops.current_stream()
    rng = np.random.default_rng(3)
    layers = [(True, 3, 9, 32), (False, 3, 32, 64), (False, 3, 64, 40)]          # (in_u8, K, C, F)
    jobs, singles = [], []
    for u8, K, C, F in layers:
        for g in range(2):
            w = dv((rng.standard_normal((K, K, C, F)) * 0.1).astype(np.float32), dev)
            nwf = ops.convs_weight_bytes(u8, K, K, C, F) // 2
            wf = [torch.full((nwf,), -1, dtype=torch.int16, device=dev) for _ in range(2)]
            wd = [torch.full((K * K * C * F * 2,), -1, dtype=torch.int16, device=dev) for _ in range(2)]
            jobs.append((w.data_ptr(), wf[0].data_ptr(), 0 if u8 else wd[0].data_ptr(), u8, K, K, C, F))
            ops.convs_prep_weights([w.data_ptr()], [wf[1].data_ptr()], None if u8 else [wd[1].data_ptr()], u8, K, K, C, F, st)
            singles.append((w, wf, wd, u8))
    ops.convs_prep_weights_multi(jobs, st)
    torch.cuda.synchronize()
    for w, wf, wd, u8 in singles:
        assert torch.equal(wf[0], wf[1])
        if not u8:
            assert torch.equal(wd[0], wd[1])


@pytest.mark.gpu
@pytest.mark.parametrize("n,lo,hi", [(1, 0, 7), (2, 0, 1), (9, 0, 12), (256, 3, 200000), (5121, 0, 1000000),
                                     (4096, -5, 2**31 - 1)])
def test_prng_randint_device_bit_exact(n, lo, hi):
    """serl_prng_randint == oracle.prng.randint (jax.random.randint restatement), odd and even n, and the max."""
    from oracle import prng as op
    from serl_amd import ops
    key = op.split(op.prng_key(n + 11))[1]
    out = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ops.prng_randint(key, n, lo, hi, out.data_ptr(), out.data_ptr() + 4 * n, torch.cuda.current_stream().cuda_stream)
    ref = op.randint(key, (n,), lo, hi)
    got = out.cpu().numpy()
    assert np.array_equal(got[:n], ref)
    assert got[n] == ref.max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((37,), np.float32), ((7,), np.uint8), ((3, 5), np.float16), ((), np.float32),
                                          ((), np.bool_), ((16, 16, 3), np.uint8), ((1024,), np.float32)])
def test_take_rows_bit_exact(shape, dtype):
    from serl_amd import ops
    rng = np.random.default_rng(0)
    nrows, n = 101, 300
    src = rng.integers(0, 255, size=(nrows, *shape)).astype(dtype)
    indx = rng.integers(0, nrows, size=n).astype(np.int32)
    indx[:3] = [0, nrows - 1, 0]
    t = torch.from_numpy(src).cuda()
    ix = torch.from_numpy(indx).cuda()
    out = torch.empty((n, *shape), dtype=t.dtype, device="cuda")
    ops.take_rows(t.data_ptr(), nrows, t[0].numel() * t.element_size(), ix.data_ptr(), out.data_ptr(), n,
                  torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(out.cpu().numpy(), src[indx])


@pytest.mark.gpu
def test_gpu_replay_buffer_matches_oracle():
    """GPUReplayBuffer.upload / insert / get_iterator (serl/data/replay_buffer.py:118-152, dataset.py:106-129)
    against oracle.replay.ring_upload + ring_sample_jax: indices, max and every leaf bit-exact over 3 draws."""
    from helpers import state_spaces
    from oracle import prng as op
    from oracle import replay as orp
    from serl_amd.data import GPUReplayBuffer
    obs_space, act_space = state_spaces(6, 3)
    cap = 50
    rb = GPUReplayBuffer(obs_space, act_space, cap)
    ob = orp.new_ring(((6,), np.float32), ((3,), np.float32), cap)
    rng = np.random.default_rng(1)

    def blk(n):
        return dict(observations=rng.standard_normal((n, 6)).astype(np.float32),
                    next_observations=rng.standard_normal((n, 6)).astype(np.float32),
                    actions=rng.standard_normal((n, 3)).astype(np.float32),
                    rewards=rng.standard_normal(n).astype(np.float32), masks=np.ones(n, np.float32),
                    dones=rng.integers(0, 2, n).astype(bool))
    for n in (20, 11):
        b = blk(n)
        rb.upload(b)
        orp.ring_upload(ob, b)
    one = {k: v[0] for k, v in blk(1).items()}
    rb.insert(one)
    orp.ring_insert(ob, one)
    assert len(rb) == ob["size"] == 32 and rb._insert_index == ob["head"]
    with pytest.raises(ValueError):
        rb.upload(blk(19))
    with pytest.raises(NotImplementedError):
        rb.sample(4)
    rb.seed(7)
    key = op.prng_key(7)
    it = rb.get_iterator(sample_args=dict(batch_size=33))
    for _ in range(3):
        mx, batch = next(it)
        key, omx, ref, indx = orp.ring_sample_jax(ob, key, 33)
        assert int(mx) == omx
        for k in ("observations", "next_observations", "actions", "rewards", "masks", "dones"):
            assert np.array_equal(batch[k].cpu().numpy(), ref[k]), k
    # default rng (no seed): PRNGKey(42), dataset.py:108
    rb2 = GPUReplayBuffer(obs_space, act_space, cap)
    rb2.upload(blk(10))
    mx, batch = rb2.sample_jax(5)
    assert int(mx) == op.randint(op.split(op.prng_key(42))[0], (5,), 0, 10).max()


def test_tanh_normal_distribution_object(dev):
    """serl_amd.distributions.TanhNormal -- the object the reference's actor apply_fn returns
    (serl/distributions/tanh_normal.py:17-53, tanh_transformed.py:17-30): sample(seed) on the jax.random.normal stream,
    log_prob of the sample (cached pre-tanh value) and of arbitrary actions (float32 atanh of the clipped action), mode()."""
    from oracle import learner as ol, prng
    from serl_amd.distributions import TanhNormal
    rng = np.random.default_rng(9)
    B, A = 37, 7
    mean = rng.standard_normal((B, A)).astype(np.float32)
    raw = (rng.standard_normal((B, A)) * 2).astype(np.float32)
    raw[0, 0], raw[1, 1] = 4.0, -30.0                      # clipped log-stds
    d = TanhNormal(dv(mean, dev), dv(raw, dev))
    assert d.batch_shape == (B,) and d.event_shape == (A,)
    key = prng.split(prng.prng_key(11))[1]
    a = d.sample(seed=key)
    m32, l32 = torch.tensor(mean), torch.clamp(torch.tensor(raw), -20.0, 2.0)
    oa, ox = ol.tanh_normal_sample(m32, l32, key)
    close(a.cpu().numpy(), oa.numpy(), rtol=2e-6, what="sample")
    close(d.log_prob(a).cpu().numpy(), ol.tanh_normal_log_prob(m32, l32, ox).numpy(), rtol=1e-5, what="log_prob of the sample")
    close(d.mode().cpu().numpy(), np.tanh(mean.astype(np.float64)), rtol=2e-6, what="mode")
    # arbitrary actions, incl. +-1 (the bijector clips before atanh)
    acts = rng.uniform(-0.999, 0.999, (B, A)).astype(np.float32)
    acts[2, 3], acts[3, 0] = 1.0, -1.0
    clipped = np.clip(acts.astype(np.float64), -float(np.float32(0.99999994)), float(np.float32(0.99999994)))
    x = torch.tensor(np.arctanh(clipped))
    want = ol.tanh_normal_log_prob(m32.double(), l32.double(), x).numpy()
    got = d.log_prob(dv(acts, dev)).cpu().numpy()
    # rows holding a clipped +-1 action sit on atanh's pole in float32: compare those loosely
    edge = (np.abs(acts) == 1).any(1)
    close(got[~edge], want[~edge], rtol=2e-5, what="log_prob of given actions")
    assert np.all(np.isfinite(got)) and np.abs(got[edge] - want[edge]).max() < 1.0
    a2, lp2 = d.sample_and_log_prob(key)
    assert torch.equal(a2, a) and torch.equal(lp2, d.log_prob(a2))
