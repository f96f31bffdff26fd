"""Thin Python front of the C ABI: every function takes raw device pointers (ints, e.g.
`tensor.data_ptr()`), sizes and a HIP stream handle, and enqueues hand-written HIP kernels.
No computation happens in Python/torch here."""
from __future__ import annotations

from . import _lib
from ._lib import call, ptr_table

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def conv_cost(kind, in_u8, n, B, IH, IW, C, KH, KW, stride, F):
    """Algorithmic (flops, HBM bytes) of one grouped conv launch: every tensor is read or
    written once; activations and gradients are 4 bytes per element in either format, uint8 pixels 1 byte."""
    OH, OW = (IH - KH) // stride + 1, (IW - KW) // stride + 1
    flops = 2.0 * n * B * OH * OW * F * KH * KW * C
    x = n * B * IH * IW * C * (1 if in_u8 else 4)
    y = n * B * OH * OW * F * 4
    w = n * KH * KW * C * F * 4
    if kind == "fwd":
        byts = x + w + y + n * F * 4
    elif kind == "dgrad":       # reads dz, w and the ReLU mask (2-byte hi plane of x), writes dx
        byts = y + w + x // 2 + x
    else:                       # wgrad: reads x and dz, writes dw + dbias (split-K slabs stay in L2 / MALL)
        byts = x + y + w + n * F * 4
    return flops, float(byts)


def conv2d_fwd(xs, ws, bs, ys, in_u8, B, IH, IW, C, KH, KW, stride, F, relu, stream):
    n = len(xs)
    px, kx = ptr_table(xs); pw, kw = ptr_table(ws); pb, kb = ptr_table(bs); py, ky = ptr_table(ys)
    call("serl_conv2d_fwd", n, px, pw, pb, py, int(in_u8), B, IH, IW, C, KH, KW, stride, F, int(relu), stream)


def conv2d_dgrad(dzs, ws, xacts, dxs, B, IH, IW, C, KH, KW, stride, F, stream):
    n = len(dzs)
    pz, kz = ptr_table(dzs); pw, kw = ptr_table(ws); pa, ka = ptr_table(xacts); pd, kd = ptr_table(dxs)
    call("serl_conv2d_dgrad", n, pz, pw, pa, pd, B, IH, IW, C, KH, KW, stride, F, stream)


def conv2d_wgrad_workspace_bytes(ngroups, B, IH, IW, C, KH, KW, stride, F) -> int:
    r = _lib.load().serl_conv2d_wgrad_workspace_bytes(ngroups, B, IH, IW, C, KH, KW, stride, F)
    if r < 0:
        raise ValueError("conv2d_wgrad_workspace_bytes: " + _lib.load().serl_last_error().decode())
    return int(r)


def conv2d_wgrad(xs, dzs, dws, dbs, workspace, workspace_bytes, in_u8, B, IH, IW, C, KH, KW, stride, F, stream):
    n = len(xs)
    px, kx = ptr_table(xs); pz, kz = ptr_table(dzs); pw, kw = ptr_table(dws); pb, kb = ptr_table(dbs)
    call("serl_conv2d_wgrad", n, px, pz, pw, pb, workspace, workspace_bytes, int(in_u8), B, IH, IW, C, KH, KW,
         stride, F, stream)


# ---- S-format conv stack (precision "bf16_3x"; include/serl_hip.h)
def convs_weight_bytes(in_u8, KH, KW, C, F) -> int:
    return int(_lib.load().serl_convs_weight_bytes(int(in_u8), KH, KW, C, F))


def split_f32(src, dst, rows, cols, stream):
    call("serl_split_f32", src, dst, rows, cols, stream)


def merge_split(src, dst, rows, cols, stream):
    call("serl_merge_split", src, dst, rows, cols, stream)


def convs_prep_weights(ws, wfs, wds, in_u8, KH, KW, C, F, stream):
    n = len(ws)
    pw, kw = ptr_table(ws); pf, kf = ptr_table(wfs)
    if wds is None:
        pd = None
    else:
        pd, kd = ptr_table(wds)
    call("serl_convs_prep_weights", n, pw, pf, pd, int(in_u8), KH, KW, C, F, stream)


def convs_prep_weights_multi(jobs, stream):
    """jobs: list of (w, wf, wd or 0, in_u8, KH, KW, C, F) -- every (layer, group) of an encoder pass in one launch."""
    import ctypes
    n = len(jobs)
    pw, kw_ = ptr_table([j[0] for j in jobs]); pf, kf = ptr_table([j[1] for j in jobs]); pd, kd = ptr_table([j[2] for j in jobs])
    cols = [(ctypes.c_int * n)(*[int(j[c]) for j in jobs]) for c in range(3, 8)]
    call("serl_convs_prep_weights_multi", n, pw, pf, pd, *cols, stream)


def convs_fwd(xs, wfs, bs, ys, in_u8, B, IH, IW, C, KH, KW, stride, F, relu, stream, rmasks=None):
    n = len(xs)
    px, kx = ptr_table(xs); pw, kw = ptr_table(wfs); pb, kb = ptr_table(bs); py, ky = ptr_table(ys)
    pm, km = ptr_table(rmasks) if rmasks is not None else (None, None)
    call("serl_convs_fwd", n, px, pw, pb, py, int(in_u8), B, IH, IW, C, KH, KW, stride, F, int(relu), pm, stream)


def convs_fwd12(xs, wf1s, b1s, wf2s, b2s, ys, B, IH1, IW1, C1, F1, F2, relu, stream, rmasks=None):
    """Fused layers 1 + 2 of a no-gradient encoder pass (uint8 frames -> S-format layer-2 output)."""
    n = len(xs)
    px, kx = ptr_table(xs); p1, k1 = ptr_table(wf1s); pb1, kb1 = ptr_table(b1s); p2, k2 = ptr_table(wf2s)
    pb2, kb2 = ptr_table(b2s); py, ky = ptr_table(ys)
    pm, km = ptr_table(rmasks) if rmasks is not None else (None, None)
    call("serl_convs_fwd12", n, px, p1, pb1, p2, pb2, py, B, IH1, IW1, C1, F1, F2, int(relu), pm, stream)


def convs_fwd12_keep(xs, wf1s, b1s, wf2s, b2s, ys, B, IH1, IW1, C1, F1, F2, relu, stream, rmasks=None, y1s=None, rmask1s=None):
    """convs_fwd12 that also writes layer 1's output (and its ReLU byte mask) for the groups with a y1 pointer."""
    n = len(xs)
    px, kx = ptr_table(xs); p1, k1 = ptr_table(wf1s); pb1, kb1 = ptr_table(b1s); p2, k2 = ptr_table(wf2s)
    pb2, kb2 = ptr_table(b2s); py, ky = ptr_table(ys)
    pm, km = ptr_table(rmasks) if rmasks is not None else (None, None)
    py1, ky1 = ptr_table(y1s) if y1s is not None else (None, None)
    pm1, km1 = ptr_table(rmask1s) if rmask1s is not None else (None, None)
    call("serl_convs_fwd12_keep", n, px, p1, pb1, p2, pb2, py, B, IH1, IW1, C1, F1, F2, int(relu), pm, py1, pm1, stream)


def convs_fwd12_keep_supported(IH1, IW1, C1, F1, F2):
    from ._lib import load
    return bool(load().serl_convs_fwd12_keep_supported(IH1, IW1, C1, F1, F2))


def convs_fwd12_supported(IH1, IW1, C1, F1, F2):
    from ._lib import load
    return bool(load().serl_convs_fwd12_supported(IH1, IW1, C1, F1, F2))


def convs_set_patch_kernels(enable):
    """Process-wide: 1 (default) = LDS-patch kernels where they fit, 0 = im2col kernels only, 2 = without the wide
    (4-wave) patch forward, 3 = plus the LDS-DMA ring GEMM for the deep layers (A/B measurements)."""
    call("serl_convs_set_patch_kernels", int(enable))


def convs_dgrad(dzs, wds, xacts, dxs, B, IH, IW, C, KH, KW, stride, F, stream, xact_bits=False):
    n = len(dzs)
    pz, kz = ptr_table(dzs); pw, kw = ptr_table(wds); pa, ka = ptr_table(xacts); pd, kd = ptr_table(dxs)
    call("serl_convs_dgrad", n, pz, pw, pa, pd, B, IH, IW, C, KH, KW, stride, F, int(xact_bits), stream)


def convs_dgrad_wgrad1_supported(IH1, IW1, C1, F1, F2):
    from ._lib import load
    return bool(load().serl_convs_dgrad_wgrad1_supported(IH1, IW1, C1, F1, F2))


def convs_dgrad_wgrad1_workspace_bytes(ngroups):
    from ._lib import load
    return int(load().serl_convs_dgrad_wgrad1_workspace_bytes(ngroups))


def convs_dgrad_wgrad1(dz2s, wd2s, rmask1s, frames, dw1s, db1s, workspace, workspace_bytes, B, IH1, IW1, C1, F1, F2, stream):
    """conv2's data gradient + conv1's weight / bias gradient in one launch: dz1 never reaches HBM."""
    n = len(dz2s)
    pz, kz = ptr_table(dz2s); pw, kw = ptr_table(wd2s)
    pm, km = ptr_table(rmask1s) if rmask1s is not None else (None, None)
    pf, kf = ptr_table(frames); pdw, kdw = ptr_table(dw1s); pdb, kdb = ptr_table(db1s)
    call("serl_convs_dgrad_wgrad1", n, pz, pw, pm, pf, pdw, pdb, workspace, workspace_bytes, B, IH1, IW1, C1, F1, F2, stream)


def convs_wgrad(xs, dzs, dws, dbs, workspace, workspace_bytes, in_u8, B, IH, IW, C, KH, KW, stride, F, stream):
    n = len(xs)
    px, kx = ptr_table(xs); pz, kz = ptr_table(dzs); pw, kw = ptr_table(dws); pb, kb = ptr_table(dbs)
    call("serl_convs_wgrad", n, px, pz, pw, pb, workspace, workspace_bytes, int(in_u8), B, IH, IW, C, KH, KW,
         stride, F, stream)


def convs_wgrad_noalloc(xs, dzs, dws, dbs, in_u8, B, IH, IW, C, KH, KW, stride, F, stream):
    """serl_convs_wgrad with the split-K slabs in the library's static device array: no workspace, no allocation."""
    n = len(xs)
    px, kx = ptr_table(xs); pz, kz = ptr_table(dzs); pw, kw = ptr_table(dws); pb, kb = ptr_table(dbs)
    call("serl_convs_wgrad_noalloc", n, px, pz, pw, pb, int(in_u8), B, IH, IW, C, KH, KW, stride, F, stream)


def gemm(a, b, c, bias, nbatch, M, N, K, *, a_zs=0, a_ms, a_ks, b_zs=0, b_ks, b_ns, c_zs=0, c_ms,
         bias_zs=0, seglen=None, a_ss=0, b_ss=0, nsplit=1, c_split_stride=0, relu=False, accumulate=False,
         a_kcontig=True, b_ncontig=True, stream=0):
    call("serl_gemm", a, b, c, bias, nbatch, M, N, K, K if seglen is None else seglen,
         a_zs, a_ss, a_ms, a_ks, b_zs, b_ss, b_ks, b_ns, c_zs, c_ms, bias_zs,
         nsplit, c_split_stride, int(relu), int(accumulate), int(a_kcontig), int(b_ncontig), stream)


def ln_act_fwd(pre, nslab, pre_ss, ldpre, bias, gamma, beta, pgs, y, ldy, xhat, rstd, R, N, rows_per_group,
               use_ln, act, stream, head=None):
    """head = (w, w_group_stride, b, b_group_stride, q): fuse the Dense(1) head q[row] = y[row,:].w[e,:] + b[e]."""
    if head is not None:
        call("serl_ln_act_head_fwd", pre, nslab, pre_ss, ldpre, bias, gamma, beta, pgs, y, ldy, xhat, rstd, R, N,
             rows_per_group, int(use_ln), act, *head, stream)
        return
    call("serl_ln_act_fwd", pre, nslab, pre_ss, ldpre, bias, gamma, beta, pgs, y, ldy, xhat, rstd, R, N,
         rows_per_group, int(use_ln), act, stream)


def ln_act_bwd(dy, lddy, dq, w_head, w_gs, y, ldy, xhat, rstd, gamma, pgs, dz, g, R, N, rows_per_group,
               use_ln, act, stream):
    call("serl_ln_act_bwd", dy, lddy, dq, w_head, w_gs, y, ldy, xhat, rstd, gamma, pgs, dz, g, R, N,
         rows_per_group, int(use_ln), act, stream)


def trunk_ln_fwd(slab, nslab, slab_ss, N0, bias0, gamma0, beta0, y0, xhat0, rstd0, x1, ldx1, K1, W1, N1, bias1, gamma1,
                 beta1, y1, xhat1, rstd1, ldy, R, act, stream):
    call("serl_trunk_ln_fwd", slab, nslab, slab_ss, N0, bias0, gamma0, beta0, y0, xhat0, rstd0, x1, ldx1, K1, W1, N1,
         bias1, gamma1, beta1, y1, xhat1, rstd1, ldy, R, act, stream)


def ln_act_bwd_seg(segs, dy_nslab, dy_ss, R, act, stream):
    """segs: list of (dy, lddy, y, ldy, xhat, rstd, gamma, dz, g, N) sharing the rows; dy is summed over dy_nslab slabs."""
    import ctypes
    n = len(segs)
    tabs = {c: ptr_table([sg[c] for sg in segs]) for c in (0, 2, 4, 5, 6, 7, 8)}
    ints = {c: (ctypes.c_int * n)(*[int(sg[c]) for sg in segs]) for c in (1, 3, 9)}
    call("serl_ln_act_bwd_seg", n, tabs[0][0], ints[1], int(dy_nslab), int(dy_ss), tabs[2][0], ints[3], tabs[4][0],
         tabs[5][0], tabs[6][0], tabs[7][0], tabs[8][0], R, ints[9], act, stream)


def colsum(x0, out0, ngroups, rows_per_group, N, ldx, out_gs, stream, w=None, mul=None, x1=None, out1=None,
           x2=None, mul2=None, out2=None):
    call("serl_colsum", x0, w, mul, out0, x1, out1, x2, mul2, out2, ngroups, rows_per_group, N, ldx, out_gs, stream)


def rowdot(h, w, w_gs, b, b_gs, q, R, N, rows_per_group, stream):
    call("serl_rowdot", h, w, w_gs, b, b_gs, q, R, N, rows_per_group, stream)


def copy2d(dst, ldd, src, lds, rows, cols, stream):
    call("serl_copy2d", dst, ldd, src, lds, rows, cols, stream)


def fill(p, v, n, stream):
    call("serl_fill", p, float(v), n, stream)


def tanh_gaussian_fwd(mean, lstd, ld_ms, act, ld_act, xpre, eps, logp, B, A, key, row_offset, total_rows,
                      deterministic, stream):
    call("serl_tanh_gaussian_fwd", mean, lstd, ld_ms, act, ld_act, xpre, eps, logp, B, A, int(key[0]), int(key[1]),
         row_offset, total_rows, int(deterministic), stream)


def tanh_gaussian_bwd(act, ld_act, eps, lstd, ld_ms, da, ld_da, log_temp, inv_batch, dmean, dlstd, ld_out, B, A,
                      stream):
    call("serl_tanh_gaussian_bwd", act, ld_act, eps, lstd, ld_ms, da, ld_da, log_temp, float(inv_batch), dmean,
         dlstd, ld_out, B, A, stream)


def td_loss(next_q, emin, rew, mask, next_logp, log_temp, discount, q, E, B, inv_count, target, dq, info, stream):
    call("serl_td_loss", next_q, emin, rew, mask, next_logp, log_temp, float(discount), q, E, B, float(inv_count),
         target, dq, info, stream)


def actor_loss(q, E, B, logp, log_temp, info, stream):
    call("serl_actor_loss", q, E, B, logp, log_temp, info, stream)


def adam_polyak(p, m, v, g, target, n, lr, b1, b2, eps, bc1, bc2, tau, grad_scale, stream):
    call("serl_adam_polyak", p, m, v, g, target, n, float(lr), float(b1), float(b2), float(eps), float(bc1),
         float(bc2), float(tau), float(grad_scale), stream)


def temperature_update(state, sum_logp, inv_batch, target_entropy, lr, b1, b2, eps, bc1, bc2, info_out, stream):
    call("serl_temperature_update", state, sum_logp, float(inv_batch), float(target_entropy), float(lr), float(b1),
         float(b2), float(eps), float(bc1), float(bc2), info_out, stream)


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def sle_fwd(feats, kers, outs, ldo, B, HW, C, F, stream, feat_split=False):
    """Grouped: feats / kers / outs are equally long lists of pointers (a single pointer = one group)."""
    feats, kers, outs = _as_list(feats), _as_list(kers), _as_list(outs)
    pf, kf = ptr_table(feats); pk, kk = ptr_table(kers); po, ko = ptr_table(outs)
    call("serl_sle_fwd", len(feats), pf, pk, po, ldo, B, HW, C, F, int(feat_split), stream)


def sle_bwd(douts, ldo, kers, feats, dfeats, dkers, B, HW, C, F, stream, feat_split=False):
    douts, kers, feats = _as_list(douts), _as_list(kers), _as_list(feats)
    pd, kd = ptr_table(douts); pk, kk = ptr_table(kers); pf, kf = ptr_table(feats)
    pdf = pdk = None
    if dfeats is not None:
        pdf, kdf = ptr_table(_as_list(dfeats))
    if dkers is not None:
        pdk, kdk = ptr_table(_as_list(dkers))
    call("serl_sle_bwd", len(douts), pd, ldo, pk, pf, pdf, pdk, B, HW, C, F, int(feat_split), stream)   # bit 1: no ReLU mask


def gather_crop_frames(frames, cap, indx, T, t0, off, out, n, H, W, S, pad, stream):
    call("serl_gather_crop_frames", frames, cap, indx, T, t0, off, out, n, H, W, S, pad, stream)


def gather_crop_packed(src, T, t0, off, out, n, H, W, S, pad, stream):
    call("serl_gather_crop_packed", src, T, t0, off, out, n, H, W, S, pad, stream)


def gather_window(frames, cap, indx, out, n, HW3, S, stream):
    call("serl_gather_window", frames, cap, indx, out, n, HW3, S, stream)


def prng_randint(key, n, minval, maxval, out, out_max, stream):
    call("serl_prng_randint", int(key[0]), int(key[1]), n, minval, maxval, out, out_max, stream)


def take_rows(src, nrows, row_bytes, indx, out, n, stream):
    call("serl_take_rows", src, nrows, row_bytes, indx, out, n, stream)


def crop_offsets(key, n, row_offset, n_total, pad, out, stream):
    call("serl_crop_offsets", int(key[0]), int(key[1]), n, row_offset, n_total, pad, out, stream)


def dropout(x, ld, rows, cols, key, keep_prob, row_offset, total_rows, stream):
    call("serl_dropout", x, ld, rows, cols, int(key[0]), int(key[1]), float(keep_prob), row_offset, total_rows, stream)


def vice_reward(logits, sigmoids, mean_out, rewards, n, stream):
    """logits / sigmoids: per camera device pointers ((n,) floats; sigmoids may be None)."""
    pl, kl = ptr_table(logits)
    ps = None
    if sigmoids is not None:
        ps, ks = ptr_table(sigmoids)
    call("serl_vice_reward", len(logits), pl, ps, mean_out, rewards, n, stream)


def row_combine(x, src_rows, ia, ib, wa, wb, out, rows, cols, stream):
    call("serl_row_combine", x, src_rows, ia, ib, wa, wb, out, rows, cols, stream)


def bce_mixup(y, ya, yb, lam, n, dy, loss, stream):
    call("serl_bce_mixup", y, ya, yb, float(lam), n, dy, loss, stream)


def gp_norm(G, rows, D, scale, alpha, norms, stream):
    call("serl_gp_norm", G, rows, D, float(scale), alpha, norms, stream)


def ln_tangent_fwd(a, xhat, rstd, gamma, y, out, w, R, N, use_ln, act, stream):
    call("serl_ln_tangent_fwd", a, xhat, rstd, gamma, y, out, w, R, N, int(use_ln), act, stream)


def ln_tangent_bwd(outbar, tbar_ext, y, w, a, xhat, rstd, gamma, abar, ubar, ggamma, nbar, R, N, use_ln, act, stream):
    call("serl_ln_tangent_bwd", outbar, tbar_ext, y, w, a, xhat, rstd, gamma, abar, ubar, ggamma, nbar, R, N, int(use_ln),
         act, stream)


def axpy(y, x, a, n, stream):
    call("serl_axpy", y, x, float(a), n, stream)
