"""VICELearner.update_classifier (serl/agents/vice/vice_learner.py:316-474): one training step of every camera's VICE
classifier -- batch = [goal, aug(goal), obs, aug(obs)] with smoothed labels 0.9 / 0.1, frozen-backbone encoding, mixup,
mixup binary cross-entropy + 10 x gradient penalty on interpolated embeddings, Adam.

Everything runs in HIP kernels; this module sequences them.  The gradient penalty's parameter gradient (the reference:
jax.grad through vmap(jax.grad)) is the hand-derived "reverse mode through the tangent pass" of csrc/vice.hip.
Random streams: jax's split / randint / permutation / uniform as the reference draws them, except
  * lam = jax.random.beta(key_0, 1, 1) is drawn as uniform(key_0): Beta(1,1) is the uniform distribution, the reference's
    gamma-rejection bit stream is not restated;
  * Dropout masks: bernoulli(fold_in(key, layer), keep) -- flax's path-folded rng is not restatable.
Python closes over the NAME `key`: both loss closures of the reference see the second `key, rng = split(rng, 2)` (:438,451),
and under vmap(grad(.)) with that closed-over rng every gradient-penalty sample gets the same (1, F) mask."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from ... import ops, prng
from ...data.device_batch import FrameWindows
from ...networks.heads import ACT_LEAKY, VICE_HIDDEN
from ...networks.params import ParamStore
from ...networks.spec import SLE_FEATURES, TRUNK_DIM
from ..train_state import ADAM_B1, ADAM_B2, ADAM_EPS, bias_corrections

PAD = 4
PRE = ("OneDimOutput_0",)


def _uniform(key, n):
    """jax.random.uniform(key, (n,)) on the host (float32 in [0, 1))."""
    bits = prng.bits(key, max(n, 1))[:n] if n else prng.bits(key, 1)
    return ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def _obs_pixels(leaf, in_next, dev, n, H, W):
    """The observation images of a batch leaf as a device (n,H,W,3) uint8 tensor (`_unpack`: [..., :-1] of a packed leaf)."""
    if isinstance(leaf, FrameWindows):
        leaf = leaf.packed if leaf.packed is not None else leaf.materialize()
        in_next = False
    t = leaf if isinstance(leaf, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(leaf)))
    t = t.to(dev)
    if not in_next:
        t = t[..., :-1]
    assert tuple(t.shape) == (n, H, W, 3, 1) and t.dtype == torch.uint8, f"VICE needs (n,H,W,3,1) uint8 pixels, got {tuple(t.shape)}"
    return t.reshape(n, H, W, 3).contiguous()


class _Net:
    """Launch sequences of one VICE classifier (serl_amd/networks/heads.py layout) in training mode."""

    def __init__(self, agent, store, ws, st):
        self.store, self.ws, self.st = store, ws, st
        self.fh, self.fw, self.fc = agent._shared["feat_shape"]
        self.F8 = self.fc * SLE_FEATURES
        self.ln = agent._cfg.critic_layer_norm
        self.keep = 1.0 - float(agent.vice_dropout_rate or 0.0)

    def p(self, *path):
        return self.store.ptr(*path)

    def drop(self, t, cols, key, which, rows, bcast):
        if self.keep >= 1.0:
            return
        k = prng.fold_in(key, which)
        ops.dropout(t.data_ptr(), cols, rows, cols, k, self.keep, -1 if bcast else 0, 1 if bcast else rows, self.st)

    def forward(self, x_ptr, rows, key, bcast, tag):
        """Training-mode forward on (rows, fh, fw, fc) features; keeps what the backward passes need under `tag`."""
        ws, st, F8, T, Hd = self.ws, self.st, self.F8, TRUNK_DIM, VICE_HIDDEN
        g = lambda name, shape: ws.get(f"{tag}_{name}", shape)
        s = g("s", (rows, F8))
        ops.sle_fwd([x_ptr], [self.p("SpatialLearnedEmbeddings_0", "kernel")], [s.data_ptr()], F8, rows, self.fh * self.fw,
                    self.fc, SLE_FEATURES, st)
        self.drop(s, F8, key, 0, rows, bcast)                                    # encoded_encoder.py:34
        pre1 = g("pre1", (rows, T))
        ops.gemm(s.data_ptr(), self.p("Dense_0", "kernel"), pre1.data_ptr(), None, 1, rows, T, F8, a_ms=F8, a_ks=1, b_ks=T,
                 b_ns=1, c_ms=T, stream=st)
        t, xh1, rs1 = g("t", (rows, T)), g("xh1", (rows, T)), g("rs1", (rows,))
        ops.ln_act_fwd(pre1.data_ptr(), 1, 0, T, self.p("Dense_0", "bias"), self.p("LayerNorm_0", "scale"),
                       self.p("LayerNorm_0", "bias"), 0, t.data_ptr(), T, xh1.data_ptr(), rs1.data_ptr(), rows, T, rows,
                       True, ops.ACT_TANH, st)
        v = g("v", (rows, Hd))
        ops.gemm(t.data_ptr(), self.p(*PRE, "MLP_0", "Dense_0", "kernel"), v.data_ptr(), self.p(*PRE, "MLP_0", "Dense_0", "bias"),
                 1, rows, Hd, T, a_ms=T, a_ks=1, b_ks=Hd, b_ns=1, c_ms=Hd, stream=st)
        self.drop(v, Hd, key, 1, rows, bcast)                                    # mlp.py:26-29 (after Dense, before LayerNorm)
        r, xh2, rs2 = g("r", (rows, Hd)), g("xh2", (rows, Hd)), g("rs2", (rows,))
        ops.ln_act_fwd(v.data_ptr(), 1, 0, Hd, None, self.p(*PRE, "MLP_0", "LayerNorm_0", "scale") if self.ln else None,
                       self.p(*PRE, "MLP_0", "LayerNorm_0", "bias") if self.ln else None, 0, r.data_ptr(), Hd,
                       xh2.data_ptr() if self.ln else None, rs2.data_ptr() if self.ln else None, rows, Hd, rows, self.ln,
                       ACT_LEAKY, st)
        y = g("y", (rows,))
        ops.rowdot(r.data_ptr(), self.p(*PRE, "Dense_0", "kernel"), 0, self.p(*PRE, "Dense_0", "bias"), 0, y.data_ptr(), rows,
                   Hd, rows, st)
        return dict(s=s, t=t, xh1=xh1, rs1=rs1, r=r, xh2=xh2, rs2=rs2, y=y, rows=rows, tag=tag)

    def backward(self, f, dy, x_ptr, key, bcast, grads: ParamStore | None, want_dx):
        """Reverse pass from dy (rows,): parameter gradients into `grads` (overwritten) and / or the input gradient."""
        ws, st, F8, T, Hd, rows, tag = self.ws, self.st, self.F8, TRUNK_DIM, VICE_HIDDEN, f["rows"], f["tag"]
        g = lambda name, shape: ws.get(f"{tag}_b_{name}", shape)
        gp = (lambda *path: grads.ptr(*path)) if grads is not None else None
        w3 = self.p(*PRE, "Dense_0", "kernel")
        if grads is not None:
            ops.colsum(f["r"].data_ptr(), gp(*PRE, "Dense_0", "kernel"), 1, rows, Hd, Hd, 0, st, w=dy.data_ptr())
            ops.colsum(dy.data_ptr(), gp(*PRE, "Dense_0", "bias"), 1, rows, 1, 1, 0, st)
        dz2, g2 = g("dz2", (rows, Hd)), g("g2", (rows, Hd))
        ops.ln_act_bwd(None, 0, dy.data_ptr(), w3, 0, f["r"].data_ptr(), Hd, f["xh2"].data_ptr() if self.ln else None,
                       f["rs2"].data_ptr() if self.ln else None, self.p(*PRE, "MLP_0", "LayerNorm_0", "scale") if self.ln else None,
                       0, dz2.data_ptr(), g2.data_ptr() if self.ln else None, rows, Hd, rows, self.ln, ACT_LEAKY, st)
        if grads is not None and self.ln:
            ops.colsum(g2.data_ptr(), gp(*PRE, "MLP_0", "LayerNorm_0", "bias"), 1, rows, Hd, Hd, 0, st,
                       x2=g2.data_ptr(), mul2=f["xh2"].data_ptr(), out2=gp(*PRE, "MLP_0", "LayerNorm_0", "scale"))
        self.drop(dz2, Hd, key, 1, rows, bcast)                                  # dv = dv1 * mask
        if grads is not None:
            ops.colsum(dz2.data_ptr(), gp(*PRE, "MLP_0", "Dense_0", "bias"), 1, rows, Hd, Hd, 0, st)
            ops.gemm(f["t"].data_ptr(), dz2.data_ptr(), gp(*PRE, "MLP_0", "Dense_0", "kernel"), None, 1, T, Hd, rows, a_ms=1,
                     a_ks=T, b_ks=Hd, b_ns=1, c_ms=Hd, a_kcontig=False, stream=st)
        dt = g("dt", (rows, T))
        ops.gemm(dz2.data_ptr(), self.p(*PRE, "MLP_0", "Dense_0", "kernel"), dt.data_ptr(), None, 1, rows, T, Hd, a_ms=Hd,
                 a_ks=1, b_ks=1, b_ns=Hd, c_ms=T, b_ncontig=False, stream=st)
        dz1, g1 = g("dz1", (rows, T)), g("g1", (rows, T))
        ops.ln_act_bwd(dt.data_ptr(), T, None, None, 0, f["t"].data_ptr(), T, f["xh1"].data_ptr(), f["rs1"].data_ptr(),
                       self.p("LayerNorm_0", "scale"), 0, dz1.data_ptr(), g1.data_ptr(), rows, T, rows, True, ops.ACT_TANH, st)
        if grads is not None:
            ops.colsum(g1.data_ptr(), gp("LayerNorm_0", "bias"), 1, rows, T, T, 0, st, x1=dz1.data_ptr(),
                       out1=gp("Dense_0", "bias"), x2=g1.data_ptr(), mul2=f["xh1"].data_ptr(), out2=gp("LayerNorm_0", "scale"))
            ops.gemm(f["s"].data_ptr(), dz1.data_ptr(), gp("Dense_0", "kernel"), None, 1, F8, T, rows, a_ms=1, a_ks=F8,
                     b_ks=T, b_ns=1, c_ms=T, a_kcontig=False, stream=st)
        ds = g("ds", (rows, F8))
        ops.gemm(dz1.data_ptr(), self.p("Dense_0", "kernel"), ds.data_ptr(), None, 1, rows, F8, T, a_ms=T, a_ks=1, b_ks=1,
                 b_ns=T, c_ms=F8, b_ncontig=False, stream=st)
        self.drop(ds, F8, key, 0, rows, bcast)
        dx = g("dx", (rows, self.fh, self.fw, self.fc)) if want_dx else None
        ops.sle_bwd([ds.data_ptr()], F8, [self.p("SpatialLearnedEmbeddings_0", "kernel")], [x_ptr],
                    [dx.data_ptr()] if want_dx else None,
                    [gp("SpatialLearnedEmbeddings_0", "kernel")] if grads is not None else None, rows, self.fh * self.fw,
                    self.fc, SLE_FEATURES, st, feat_split=2)      # 2: fp32 features, no ReLU mask on the input gradient
        return dx

    def penalty_gradient(self, f, x_ptr, G, alpha, key, grads: ParamStore, tmp: ParamStore):
        """d/d theta of sum_i alpha_i * (tangent of y_i along G_i): the gradient penalty's parameter gradient, written to
        `grads` (every leaf it touches is overwritten; the caller zeroed the rest); `tmp` holds one extra SLE-kernel term."""
        ws, st, F8, T, Hd, M = self.ws, self.st, self.F8, TRUNK_DIM, VICE_HIDDEN, f["rows"]
        g = lambda name, shape: ws.get(f"gp_t_{name}", shape)
        gp = lambda *path: grads.ptr(*path)
        W1, W2, w3 = self.p("Dense_0", "kernel"), self.p(*PRE, "MLP_0", "Dense_0", "kernel"), self.p(*PRE, "Dense_0", "kernel")
        gam1 = self.p("LayerNorm_0", "scale")
        gam2 = self.p(*PRE, "MLP_0", "LayerNorm_0", "scale") if self.ln else None
        # ---- tangent forward along G
        sd = g("sd", (M, F8))
        ops.sle_fwd([G.data_ptr()], [self.p("SpatialLearnedEmbeddings_0", "kernel")], [sd.data_ptr()], F8, M, self.fh * self.fw,
                    self.fc, SLE_FEATURES, st)
        self.drop(sd, F8, key, 0, M, True)
        ud = g("ud", (M, T))
        ops.gemm(sd.data_ptr(), W1, ud.data_ptr(), None, 1, M, T, F8, a_ms=F8, a_ks=1, b_ks=T, b_ns=1, c_ms=T, stream=st)
        td, w1 = g("td", (M, T)), g("w1", (M, T))
        ops.ln_tangent_fwd(ud.data_ptr(), f["xh1"].data_ptr(), f["rs1"].data_ptr(), gam1, f["t"].data_ptr(), td.data_ptr(),
                           w1.data_ptr(), M, T, True, ops.ACT_TANH, st)
        vd = g("vd", (M, Hd))
        ops.gemm(td.data_ptr(), W2, vd.data_ptr(), None, 1, M, Hd, T, a_ms=T, a_ks=1, b_ks=Hd, b_ns=1, c_ms=Hd, stream=st)
        self.drop(vd, Hd, key, 1, M, True)
        rd, w2 = g("rd", (M, Hd)), g("w2", (M, Hd))
        ops.ln_tangent_fwd(vd.data_ptr(), f["xh2"].data_ptr() if self.ln else None, f["rs2"].data_ptr() if self.ln else None,
                           gam2, f["r"].data_ptr(), rd.data_ptr(), w2.data_ptr(), M, Hd, self.ln, ACT_LEAKY, st)
        # ---- adjoint of Phi = sum alpha * (rd . w3)
        ops.colsum(rd.data_ptr(), gp(*PRE, "Dense_0", "kernel"), 1, M, Hd, Hd, 0, st, w=alpha.data_ptr())
        rbar = g("rbar", (M, Hd))
        ops.gemm(alpha.data_ptr(), w3, rbar.data_ptr(), None, 1, M, Hd, 1, a_ms=1, a_ks=1, b_ks=Hd, b_ns=1, c_ms=Hd, stream=st)
        ab2, ub2, gg2, nb2 = g("ab2", (M, Hd)), g("ub2", (M, Hd)), g("gg2", (M, Hd)), g("nb2", (M, Hd))
        ops.ln_tangent_bwd(rbar.data_ptr(), None, f["r"].data_ptr(), w2.data_ptr(), vd.data_ptr(),
                           f["xh2"].data_ptr() if self.ln else None, f["rs2"].data_ptr() if self.ln else None, gam2,
                           ab2.data_ptr(), ub2.data_ptr(), gg2.data_ptr(), nb2.data_ptr(), M, Hd, self.ln, ACT_LEAKY, st)
        if self.ln:
            ops.colsum(nb2.data_ptr(), gp(*PRE, "MLP_0", "LayerNorm_0", "bias"), 1, M, Hd, Hd, 0, st,
                       x1=gg2.data_ptr(), out1=gp(*PRE, "MLP_0", "LayerNorm_0", "scale"))
        self.drop(ab2, Hd, key, 1, M, True)                                      # adjoint of the tangent of Dense_2's output
        self.drop(ub2, Hd, key, 1, M, True)                                      # adjoint of its primal
        ops.colsum(ub2.data_ptr(), gp(*PRE, "MLP_0", "Dense_0", "bias"), 1, M, Hd, Hd, 0, st)
        ops.gemm(td.data_ptr(), ab2.data_ptr(), gp(*PRE, "MLP_0", "Dense_0", "kernel"), None, 1, T, Hd, M, a_ms=1, a_ks=T,
                 b_ks=Hd, b_ns=1, c_ms=Hd, a_kcontig=False, stream=st)
        ops.gemm(f["t"].data_ptr(), ub2.data_ptr(), gp(*PRE, "MLP_0", "Dense_0", "kernel"), None, 1, T, Hd, M, a_ms=1, a_ks=T,
                 b_ks=Hd, b_ns=1, c_ms=Hd, a_kcontig=False, accumulate=True, stream=st)
        tdbar, tbar = g("tdbar", (M, T)), g("tbar", (M, T))
        for src, dst in ((ab2, tdbar), (ub2, tbar)):
            ops.gemm(src.data_ptr(), W2, dst.data_ptr(), None, 1, M, T, Hd, a_ms=Hd, a_ks=1, b_ks=1, b_ns=Hd, c_ms=T,
                     b_ncontig=False, stream=st)
        ab1, ub1, gg1, nb1 = g("ab1", (M, T)), g("ub1", (M, T)), g("gg1", (M, T)), g("nb1", (M, T))
        ops.ln_tangent_bwd(tdbar.data_ptr(), tbar.data_ptr(), f["t"].data_ptr(), w1.data_ptr(), ud.data_ptr(),
                           f["xh1"].data_ptr(), f["rs1"].data_ptr(), gam1, ab1.data_ptr(), ub1.data_ptr(), gg1.data_ptr(),
                           nb1.data_ptr(), M, T, True, ops.ACT_TANH, st)
        ops.colsum(nb1.data_ptr(), gp("LayerNorm_0", "bias"), 1, M, T, T, 0, st, x1=ub1.data_ptr(), out1=gp("Dense_0", "bias"))
        ops.colsum(gg1.data_ptr(), gp("LayerNorm_0", "scale"), 1, M, T, T, 0, st)
        ops.gemm(sd.data_ptr(), ab1.data_ptr(), gp("Dense_0", "kernel"), None, 1, F8, T, M, a_ms=1, a_ks=F8, b_ks=T, b_ns=1,
                 c_ms=T, a_kcontig=False, stream=st)
        ops.gemm(f["s"].data_ptr(), ub1.data_ptr(), gp("Dense_0", "kernel"), None, 1, F8, T, M, a_ms=1, a_ks=F8, b_ks=T, b_ns=1,
                 c_ms=T, a_kcontig=False, accumulate=True, stream=st)
        sdbar, sbar = g("sdbar", (M, F8)), g("sbar", (M, F8))
        for src, dst in ((ab1, sdbar), (ub1, sbar)):
            ops.gemm(src.data_ptr(), W1, dst.data_ptr(), None, 1, M, F8, T, a_ms=T, a_ks=1, b_ks=1, b_ns=T, c_ms=F8,
                     b_ncontig=False, stream=st)
            self.drop(dst, F8, key, 0, M, True)
        K = ("SpatialLearnedEmbeddings_0", "kernel")
        hw = self.fh * self.fw
        ops.sle_bwd([sdbar.data_ptr()], F8, [self.p(*K)], [G.data_ptr()], None, [gp(*K)], M, hw, self.fc, SLE_FEATURES, st)
        ops.sle_bwd([sbar.data_ptr()], F8, [self.p(*K)], [x_ptr], None, [tmp.ptr(*K)], M, hw, self.fc, SLE_FEATURES, st)
        nk = hw * self.fc * SLE_FEATURES
        ops.axpy(gp(*K), tmp.ptr(*K), 1.0, nk, st)


def update_classifier(agent, batch, pixel_keys):
    cfg = agent._cfg
    dev = agent.actor.store.flat.device
    if agent.vice_goal_pool is None:
        raise ValueError("VICELearner.update_classifier needs vice_goal_pool= at create()")
    H, W = cfg.H, cfg.W
    obs, nxt = batch["observations"], batch["next_observations"]
    n = len(batch["rewards"])
    B, N4, M = n, 4 * n, 2 * n
    if N4 > 4096:
        raise ValueError("update_classifier: batch sizes up to 1024 (host-side permutation of 4 x batch rows)")
    ws = agent._shared["reward_ws"]
    backbone = agent._shared["backbone"]
    fh, fw, fc = agent._shared["feat_shape"]
    D = fh * fw * fc
    info, new_cls = OrderedDict(), OrderedDict()
    u8 = lambda name, shape: agent._shared.setdefault("vice_u8", {}).setdefault(
        (name, shape), torch.empty(shape, dtype=torch.uint8, device=dev))
    i32 = lambda t: torch.from_numpy(np.ascontiguousarray(t, dtype=np.int32)).to(dev)
    f32 = lambda t: torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(dev)
    with torch.cuda.device(dev):
        st = ops.current_stream()
        rng = agent.rng
        key_aug, rng = prng.split(rng)                                            # :368
        off = torch.empty((B, 2), dtype=torch.int32, device=dev)

        def crop(key, src, dst):
            """dst = batched_random_crop of the (B,H,W,3) uint8 images with the key of one data_augmentation_fn call's slot."""
            ops.crop_offsets(key, B, 0, B, PAD, off.data_ptr(), st)
            ops.gather_crop_packed(src.data_ptr(), 1, 0, off.data_ptr(), dst.data_ptr(), B, H, W, 1, PAD, st)

        def slot_key(key, p_key):
            """data_augmentation_fn(key, .): `k, key = split(key)` per pixel key in order; the k of p_key."""
            for pk in cfg.pixel_keys:
                k, key = prng.split(key)
                if pk == p_key:
                    return k
        pools = agent._shared.setdefault("vice_goal_dev", {})
        for p_key in pixel_keys:
            ts = agent.vice_classifiers[p_key]
            pixels = _obs_pixels(obs[p_key], p_key in nxt, dev, B, H, W)
            allpx = u8("all", (N4, H, W, 3))                                      # [goal, aug goal, obs, aug obs] (:396-398)
            goal, aug_goal, px, aug_px = allpx[:B], allpx[B:2 * B], allpx[2 * B:3 * B], allpx[3 * B:]
            px.copy_(pixels)
            crop(slot_key(key_aug, p_key), pixels, aug_px)
            _, rng = prng.split(rng)                                              # :377 (unused key)
            key_s, rng = prng.split(rng)                                          # _sample :335-339
            pool = pools.get(p_key)
            if pool is None:
                gpool = agent.vice_goal_pool[p_key]
                gpool = gpool if isinstance(gpool, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(gpool)))
                pool = pools[p_key] = gpool.to(dev).reshape(gpool.shape[0], H, W, 3).contiguous()
                assert pool.dtype == torch.uint8
            indx = torch.empty(B, dtype=torch.int32, device=dev)
            ops.prng_randint(key_s, B, 0, pool.shape[0], indx.data_ptr(), None, st)
            ops.take_rows(pool.data_ptr(), pool.shape[0], H * W * 3, indx.data_ptr(), goal.data_ptr(), B, st)
            key_g, rng = prng.split(rng)                                          # :387
            crop(slot_key(key_g, p_key), goal, aug_goal)
            enc = ws.get("uc_enc", (N4, fh, fw, fc))
            backbone.forward(allpx, out=enc, stream=st)                           # encode (:410)
            k3 = prng.split(rng, 3)                                               # :413
            key_0, key_1, rng = k3[0], k3[1], k3[2]
            lam = float(_uniform(key_0, 1)[0])                                    # beta(key_0, 1, 1): see module doc
            perm = prng.choice_no_replace(key_1, N4, N4)                          # jax.random.permutation(key_1, 4B)
            y = (np.concatenate([np.ones(M), np.zeros(M)]).astype(np.float32) * np.float32(0.8) + np.float32(0.1))   # :401-407
            mix = ws.get("uc_mix", (N4, fh, fw, fc))
            ar = np.arange(N4)
            # (named tensors: a temporary's block would go back to the allocator -- and to the next temporary -- at once)
            ia, ib, wa, wb = i32(ar), i32(perm), f32(np.full(N4, lam)), f32(np.full(N4, 1.0 - lam))
            ops.row_combine(enc.data_ptr(), N4, ia.data_ptr(), ib.data_ptr(), wa.data_ptr(), wb.data_ptr(), mix.data_ptr(), N4, D, st)
            key_e, rng = prng.split(rng)                                          # :421
            eps = _uniform(key_e, M)
            gpx = ws.get("uc_gpx", (M, fh, fw, fc))
            ga, gb, ea, eb = i32(ar[:M]), i32(ar[:M] + M), f32(eps), f32(np.float32(1.0) - eps)
            ops.row_combine(mix.data_ptr(), N4, ga.data_ptr(), gb.data_ptr(), ea.data_ptr(), eb.data_ptr(), gpx.data_ptr(), M, D, st)
            _, rng = prng.split(rng)                                              # :438 (`key` is rebound before it is used)
            key_d, rng = prng.split(rng)                                          # :451 the dropout key of both applies
            net = _Net(agent, ts.store, ws, st)
            g_bce, g_gp, g_tmp = (agent._shared.setdefault(("vice_grads", i), ts.store.zeros_like()) for i in range(3))
            # ---- mixup BCE (:440-449)
            fm = net.forward(mix.data_ptr(), N4, key_d, False, "mx")
            dy, loss = ws.get("uc_dy", (N4,)), ws.get("uc_loss", (1,))
            ya, yb = f32(y), f32(y[perm])
            ops.bce_mixup(fm["y"].data_ptr(), ya.data_ptr(), yb.data_ptr(), lam, N4, dy.data_ptr(), loss.data_ptr(), st)
            net.backward(fm, dy, mix.data_ptr(), key_d, False, g_bce, want_dx=False)
            # ---- gradient penalty (:451-465): G = d y / d x per sample, norms, then the gradient of 10 mean((n - 1)^2)
            fg = net.forward(gpx.data_ptr(), M, key_d, True, "gp")
            ones = ws.get("uc_ones", (M,))
            ops.fill(ones.data_ptr(), 1.0, M, st)
            G = net.backward(fg, ones, gpx.data_ptr(), key_d, True, None, want_dx=True)
            alpha, norms, nsum = ws.get("uc_alpha", (M,)), ws.get("uc_norms", (M,)), ws.get("uc_nsum", (1,))
            ops.gp_norm(G.data_ptr(), M, D, 10.0, alpha.data_ptr(), norms.data_ptr(), st)
            ops.colsum(norms.data_ptr(), nsum.data_ptr(), 1, M, 1, 1, 0, st)
            ops.fill(g_gp.base, 0.0, g_gp.flat.numel(), st)
            net.penalty_gradient(fg, gpx.data_ptr(), G, alpha, key_d, g_gp, g_tmp)
            ops.axpy(g_bce.base, g_gp.base, 1.0, g_bce.flat.numel(), st)
            agent._shared.setdefault("vice_last_grads", {})[p_key] = ParamStore(g_bce.layout, g_bce.flat.clone())   # tests / tools
            # ---- Adam (:469-470)
            new = ts.clone()
            new.step += 1
            bc1, bc2 = bias_corrections(new.step)
            ops.adam_polyak(new.store.base, new.mu.data_ptr(), new.nu.data_ptr(), g_bce.base, None, new.store.flat.numel(),
                            new.lr, ADAM_B1, ADAM_B2, ADAM_EPS, bc1, bc2, 0.0, 1.0, st)
            new_cls[p_key] = new
            info[f"bce_loss_{p_key}"] = loss.clone()
            info[f"grad_norm_{p_key}"] = nsum.clone()
        host = {k: float(v.cpu().numpy()[0]) / (M if k.startswith("grad_norm") else 1) for k, v in info.items()}
    return agent.replace(vice_classifiers=new_cls, rng=rng), host
