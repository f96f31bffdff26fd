"""Device-side forward / backward of the networks on the learner path, as sequences of HIP
kernel launches (serl_amd.ops) over flat parameter stores and a per-batch-size workspace.

What is mirrored (paths relative to the reference):
  encoder   TwoD4PGEncoder + SpatialLearnedEmbeddings   serl/networks/encoders/two_d4pg_encoder.py, spatial.py
  trunk     PixelMultiplexer Dense(512)/LN/tanh + state Dense(latent)/LN/tanh   serl/networks/pixel_multiplexer.py:25-45
  ensemble  Ensemble(StateActionValue(MLP[, LayerNorm]))  serl/networks/{ensemble,state_action_value,mlp}.py
  policy    TanhNormal(MLP)                                serl/distributions/tanh_normal.py
Backward passes are hand-derived; tests check them against torch autograd on the CPU oracle.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .spec import NetConfig, SLE_FEATURES, TRUNK_DIM, actor_head_prefix, critic_head_prefix

RELU, TANH = ops.ACT_RELU, ops.ACT_TANH


class Workspace:
    """Named, lazily allocated device buffers (activations, gradients of activations, split-K slabs)."""

    def __init__(self, device):
        self.device = device
        self._t = {}

    def tensor(self, name, shape, dtype=torch.float32):
        t = self._t.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            # the zero fill runs on torch's current stream; kernels that use the buffer may run on a side stream
            # (sac_learner._side_streams): wait for that one stream (not the device). Only on first use of a name/shape.
            torch.cuda.current_stream(self.device).synchronize()
            self._t[name] = t
        return t

    def ptr(self, name, shape, dtype=torch.float32) -> int:
        return self.tensor(name, shape, dtype).data_ptr()

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._t.values())


class GradSide:
    """Side stream for parameter-gradient kernels (column sums, dW GEMMs): they are only needed by the optimizer
    step, so they leave the critical path dz -> dX -> dz of the backward chain.  fork() makes the side stream
    wait for everything enqueued on the main stream so far and returns its handle; join() makes main wait."""

    def __init__(self):
        self.main = torch.cuda.current_stream()
        self.side = torch.cuda.Stream()
        self._ev = []
        self._n = 0

    def begin(self):
        self.main = torch.cuda.current_stream()
        self._n = 0

    def fork(self) -> int:
        if self._n == len(self._ev):
            self._ev.append(torch.cuda.Event())
        ev = self._ev[self._n]
        self._n += 1
        ev.record(self.main)
        self.side.wait_event(ev)
        return self.side.cuda_stream

    def join(self):
        if self._n == len(self._ev):
            self._ev.append(torch.cuda.Event())
        ev = self._ev[self._n]
        self._n += 1
        ev.record(self.side)
        self.main.wait_event(ev)


def _ksplit(K):
    """Split-K factor for the skinny (M = batch) trunk GEMM so that >= ~256 blocks are in flight."""
    if K < 1024:
        return 1
    n = 8
    while n > 1 and (K + n - 1) // n < 256:
        n //= 2
    klen = ((K + n - 1) // n + 15) // 16 * 16
    return (K + klen - 1) // klen


# =============================================================================== encoder
def split_conv_stack(cfg: NetConfig) -> bool:
    """True when the conv stack runs on S-format tensors (include/serl_hip.h): precision "bf16_3x" and every
    layer's channel counts fit the format (first layer: uint8 pixels with KW*C <= 32; then C, F multiples of 8).
    The activation buffers have the same byte size either way (4 bytes per element)."""
    from .. import get_matmul_precision
    if get_matmul_precision() != "bf16_3x":
        return False
    for li, L in enumerate(cfg.conv_layers()):
        if L.F % 8 or (L.K * L.C > 32 if li == 0 else L.C % 8) or L.S > 2:
            return False
    return True


def _rmask_ptr(ws, li, tag, ci, B, L):
    return ws.ptr(f"rmask{li}_{tag}_{ci}", ((B * L.OH * L.OW * L.F // 8 + 3) // 4,))


def encoder_fwd(cfg: NetConfig, passes, B, ws: Workspace, st, mask_tag=None, prep=True):
    """passes: list of (tag, ParamStore, [u8 image ptr per camera in cfg.cams order]).
    All passes x cameras run as groups of one launch per conv layer. Writes ws['enc_<tag>'] (B, enc_dim).
    S-format stack: the pass `mask_tag` (the one encoder_bwd will be called for) also writes the ReLU byte masks of
    its hidden activations, which encoder_bwd(use_masks=True) reads instead of the activations' hi planes.
    prep=False: the S-format weight copies wf/wd of these passes are still valid (parameters unchanged since the
    last call with prep=True), skip re-laying them out."""
    layers = cfg.conv_layers()
    nl, ncam = len(layers), len(cfg.cams)
    split = split_conv_stack(cfg)
    groups = [(tag, store, ci, imgs[ci]) for tag, store, imgs in passes for ci in range(ncam)]
    for g0 in range(0, len(groups), 8):
        chunk = groups[g0:g0 + 8]
        xs = [g[3] for g in chunk]
        wf_all = []
        if split:
            jobs = []
            for li, L in enumerate(layers):
                nwf = ops.convs_weight_bytes(li == 0, L.K, L.K, L.C, L.F) // 4
                wfs = [ws.ptr(f"wf{li}_{g[0]}_{g[2]}", (nwf,)) for g in chunk]
                wds = [ws.ptr(f"wd{li}_{g[0]}_{g[2]}", (L.K * L.K * L.C * L.F,)) if li > 0 else 0 for g in chunk]
                wf_all.append(wfs)
                for g, wf, wd in zip(chunk, wfs, wds):
                    jobs.append((g[1].ptr("image_encoder", f"Conv_{g[2] * nl + li}", "kernel"), wf, wd, li == 0,
                                 L.K, L.K, L.C, L.F))
            if prep:      # every (layer, group) in one launch, ahead of the dependent chain of conv kernels
                for j0 in range(0, len(jobs), 32):
                    ops.convs_prep_weights_multi(jobs[j0:j0 + 32], st)
        for li, L in enumerate(layers):
            wsp = [g[1].ptr("image_encoder", f"Conv_{g[2] * nl + li}", "kernel") for g in chunk]
            bsp = [g[1].ptr("image_encoder", f"Conv_{g[2] * nl + li}", "bias") for g in chunk]
            ys = [ws.ptr(f"act{li}_{g[0]}_{g[2]}", (B, L.OH, L.OW, L.F)) for g in chunk]
            if split:
                rm = None
                if mask_tag is not None and li < nl - 1 and any(g[0] == mask_tag for g in chunk):
                    rm = [_rmask_ptr(ws, li, g[0], g[2], B, L) if g[0] == mask_tag else 0 for g in chunk]
                ops.convs_fwd(xs, wf_all[li], bsp, ys, li == 0, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, True, st, rmasks=rm)
            else:
                ops.conv2d_fwd(xs, wsp, bsp, ys, li == 0, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, True, st)
            xs = ys
    last = layers[-1]
    for g0 in range(0, len(groups), 8):
        chunk = groups[g0:g0 + 8]
        ops.sle_fwd([ws.ptr(f"act{nl - 1}_{tag}_{ci}", (B, last.OH, last.OW, last.F)) for tag, _, ci, _ in chunk],
                    [store.ptr("image_encoder", f"SpatialLearnedEmbeddings_{ci}", "kernel") for _, store, ci, _ in chunk],
                    [ws.ptr(f"enc_{tag}", (B, cfg.enc_dim)) + 4 * ci * last.F * SLE_FEATURES for tag, _, ci, _ in chunk],
                    cfg.enc_dim, B, last.OH * last.OW, last.F, SLE_FEATURES, st, feat_split=split)


def encoder_bwd(cfg: NetConfig, tag, store, grads, imgs, B, ws: Workspace, st, gside: "GradSide | None" = None,
                use_masks=False):
    """Backward of one encoder pass from ws['denc_<tag>'] (B, enc_dim) into `grads` (same layout as store).
    With `gside` the weight-gradient kernels run on its side stream next to the data-gradient chain.
    use_masks: the forward of this pass ran with mask_tag=tag (S-format stack only)."""
    layers = cfg.conv_layers()
    nl, ncam = len(layers), len(cfg.cams)
    last = layers[-1]
    split = split_conv_stack(cfg)      # same decision as the forward that filled act* / wd* for this tag
    denc = ws.ptr(f"denc_{tag}", (B, cfg.enc_dim))
    cams = range(ncam)
    sle_args = ([denc + 4 * ci * last.F * SLE_FEATURES for ci in cams], cfg.enc_dim,
                [store.ptr("image_encoder", f"SpatialLearnedEmbeddings_{ci}", "kernel") for ci in cams],
                [ws.ptr(f"act{nl - 1}_{tag}_{ci}", (B, last.OH, last.OW, last.F)) for ci in cams])
    dfeat = [ws.ptr(f"dact{nl - 1}_{ci}", (B, last.OH, last.OW, last.F)) for ci in cams]
    dker = [grads.ptr("image_encoder", f"SpatialLearnedEmbeddings_{ci}", "kernel") for ci in cams]
    if gside is None:
        ops.sle_bwd(*sle_args, dfeat, dker, B, last.OH * last.OW, last.F, SLE_FEATURES, st, feat_split=split)
    else:
        ops.sle_bwd(*sle_args, None, dker, B, last.OH * last.OW, last.F, SLE_FEATURES, gside.fork(), feat_split=split)
        ops.sle_bwd(*sle_args, dfeat, None, B, last.OH * last.OW, last.F, SLE_FEATURES, st, feat_split=split)
    slab_bytes = max(ops.conv2d_wgrad_workspace_bytes(ncam, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F) for L in layers)
    slab = ws.ptr("wgrad_slab", (slab_bytes // 4,))
    for li in range(nl - 1, -1, -1):
        L = layers[li]
        dz = [ws.ptr(f"dact{li}_{ci}", (B, L.OH, L.OW, L.F)) for ci in range(ncam)]
        if li == 0:
            xs = list(imgs)
        else:
            P = layers[li - 1]
            xs = [ws.ptr(f"act{li - 1}_{tag}_{ci}", (B, P.OH, P.OW, P.F)) for ci in range(ncam)]
        (ops.convs_wgrad if split else ops.conv2d_wgrad)(
            xs, dz,
            [grads.ptr("image_encoder", f"Conv_{ci * nl + li}", "kernel") for ci in range(ncam)],
            [grads.ptr("image_encoder", f"Conv_{ci * nl + li}", "bias") for ci in range(ncam)],
            slab, slab_bytes, li == 0, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F,
            gside.fork() if gside is not None else st)      # dz of this layer is ready on the main stream here
        if li > 0:
            P = layers[li - 1]
            dx = [ws.ptr(f"dact{li - 1}_{ci}", (B, P.OH, P.OW, P.F)) for ci in range(ncam)]
            if split:
                wds = [ws.ptr(f"wd{li}_{tag}_{ci}", (L.K * L.K * L.C * L.F,)) for ci in range(ncam)]
                if use_masks:
                    rm = [_rmask_ptr(ws, li - 1, tag, ci, B, P) for ci in range(ncam)]
                    ops.convs_dgrad(dz, wds, rm, dx, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, st, xact_bits=True)
                else:
                    ops.convs_dgrad(dz, wds, xs, dx, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, st)
            else:
                ops.conv2d_dgrad(dz, [store.ptr("image_encoder", f"Conv_{ci * nl + li}", "kernel") for ci in range(ncam)],
                                 xs, dx, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, st)


# =============================================================================== trunk
def trunk_fwd(cfg: NetConfig, tag, store, state_ptr, z_ptr, ldz, B, ws: Workspace, st, save):
    """z[:, :512] = tanh(LN(Dense_0(enc))), z[:, 512:512+latent] = tanh(LN(Dense_1(state)))."""
    K = cfg.enc_dim
    ns = _ksplit(K)
    slab = ws.ptr(f"slab0_{tag}", (ns, B, TRUNK_DIM))
    ops.gemm(ws.ptr(f"enc_{tag}", (B, K)), store.ptr("Dense_0", "kernel"), slab, None, 1, B, TRUNK_DIM, K,
             a_ms=K, a_ks=1, b_ks=TRUNK_DIM, b_ns=1, c_ms=TRUNK_DIM, nsplit=ns, c_split_stride=B * TRUNK_DIM, stream=st)
    # one launch: slab sum + bias + LN + tanh of the image branch, and the whole state branch (Dense_1 has K = state_dim
    # ~ 19 terms per output: computed inline in fp32) -- instead of ln_act + gemm + ln_act on the latency-bound chain
    Ld, Sd = cfg.latent_dim, cfg.state_dim
    ops.trunk_ln_fwd(slab, ns, B * TRUNK_DIM, TRUNK_DIM, store.ptr("Dense_0", "bias"), store.ptr("LayerNorm_0", "scale"),
                     store.ptr("LayerNorm_0", "bias"), z_ptr,
                     ws.ptr(f"xhat0_{tag}", (B, TRUNK_DIM)) if save else None,
                     ws.ptr(f"rstd0_{tag}", (B,)) if save else None,
                     state_ptr, Sd, Sd, store.ptr("Dense_1", "kernel"), Ld, store.ptr("Dense_1", "bias"),
                     store.ptr("LayerNorm_1", "scale"), store.ptr("LayerNorm_1", "bias"), z_ptr + 4 * TRUNK_DIM,
                     ws.ptr(f"xhat1_{tag}", (B, Ld)) if save else None, ws.ptr(f"rstd1_{tag}", (B,)) if save else None,
                     ldz, B, TANH, st)


def trunk_bwd(cfg: NetConfig, tag, store, grads, state_ptr, z_ptr, ldz, dz_ptr, lddz, B, ws: Workspace, st, need_denc,
              gside: "GradSide | None" = None, dz_slabs=(1, 0)):
    """dz_ptr: gradient w.r.t. z (row stride lddz), given as the sum of dz_slabs = (count, stride in floats) split-K
    partials.  Both LayerNorm branches are differentiated in one launch."""
    K, Ld, Sd = cfg.enc_dim, cfg.latent_dim, cfg.state_dim
    stw = st            # stream of the parameter-gradient kernels
    dpre0 = ws.ptr("dpre0", (B, TRUNK_DIM)); g0 = ws.ptr("g0", (B, TRUNK_DIM))
    xh0 = ws.ptr(f"xhat0_{tag}", (B, TRUNK_DIM))
    dpre1 = ws.ptr("dpre1", (B, Ld)); g1 = ws.ptr("g1", (B, Ld))
    xh1 = ws.ptr(f"xhat1_{tag}", (B, Ld))
    ops.ln_act_bwd_seg(
        [(dz_ptr, lddz, z_ptr, ldz, xh0, ws.ptr(f"rstd0_{tag}", (B,)), store.ptr("LayerNorm_0", "scale"), dpre0, g0,
          TRUNK_DIM),
         (dz_ptr + 4 * TRUNK_DIM, lddz, z_ptr + 4 * TRUNK_DIM, ldz, xh1, ws.ptr(f"rstd1_{tag}", (B,)),
          store.ptr("LayerNorm_1", "scale"), dpre1, g1, Ld)],
        dz_slabs[0], dz_slabs[1], B, TANH, st)
    if gside is not None:
        stw = gside.fork()
    ops.colsum(g0, grads.ptr("LayerNorm_0", "bias"), 1, B, TRUNK_DIM, TRUNK_DIM, 0, stw,
               x1=dpre0, out1=grads.ptr("Dense_0", "bias"), x2=g0, mul2=xh0, out2=grads.ptr("LayerNorm_0", "scale"))
    enc = ws.ptr(f"enc_{tag}", (B, K))
    ops.gemm(enc, dpre0, grads.ptr("Dense_0", "kernel"), None, 1, K, TRUNK_DIM, B, a_ms=1, a_ks=K,
             b_ks=TRUNK_DIM, b_ns=1, c_ms=TRUNK_DIM, a_kcontig=False, stream=stw)
    if need_denc:
        ops.gemm(dpre0, store.ptr("Dense_0", "kernel"), ws.ptr(f"denc_{tag}", (B, K)), None, 1, B, K, TRUNK_DIM,
                 a_ms=TRUNK_DIM, a_ks=1, b_ks=1, b_ns=TRUNK_DIM, c_ms=K, b_ncontig=False, stream=st)
    ops.colsum(g1, grads.ptr("LayerNorm_1", "bias"), 1, B, Ld, Ld, 0, stw,
               x1=dpre1, out1=grads.ptr("Dense_1", "bias"), x2=g1, mul2=xh1, out2=grads.ptr("LayerNorm_1", "scale"))
    ops.gemm(state_ptr, dpre1, grads.ptr("Dense_1", "kernel"), None, 1, Sd, Ld, B, a_ms=1, a_ks=Sd, b_ks=Ld, b_ns=1,
             c_ms=Ld, a_kcontig=False, stream=stw)


# =============================================================================== critic ensemble
def ensemble_fwd(cfg: NetConfig, tag, store, m0, mstride, E, inp_ptr, ld_inp, B, ws: Workspace, st, save):
    """Members m0, m0+mstride, ... (E of them) of the critic ensemble on a shared input (B, ld_inp).
    Returns ptr of q (E, B)."""
    pre = critic_head_prefix(cfg)
    din = cfg.latent + cfg.action_dim
    ln = cfg.critic_layer_norm
    a_ptr, a_zs, a_ms = inp_ptr, 0, ld_inp
    nl = len(cfg.hidden_dims)
    q = ws.ptr(f"q_{tag}", (E, B))
    hl = cfg.hidden_dims[-1]
    head = (store.ptr(*pre, "Dense_0", "kernel") + 4 * m0 * hl, mstride * hl,
            store.ptr(*pre, "Dense_0", "bias") + 4 * m0, mstride, q)
    for i, h in enumerate(cfg.hidden_dims):
        W = store.ptr(*pre, "MLP_0", f"Dense_{i}", "kernel") + 4 * m0 * din * h
        bias = store.ptr(*pre, "MLP_0", f"Dense_{i}", "bias") + 4 * m0 * h
        hbuf = ws.ptr(f"h{i}_{tag}", (E, B, h))
        if ln:
            pbuf = ws.ptr(f"epre{i}_{tag}", (E, B, h))   # per tag: the shape differs between passes (E)
            ops.gemm(a_ptr, W, pbuf, None, E, B, h, din, a_zs=a_zs, a_ms=a_ms, a_ks=1, b_zs=mstride * din * h,
                     b_ks=h, b_ns=1, c_zs=B * h, c_ms=h, stream=st)
            # the last hidden layer's launch also applies the Dense(1) Q head (state_action_value.py:17-19)
            ops.ln_act_fwd(pbuf, 1, 0, h, bias, store.ptr(*pre, "MLP_0", f"LayerNorm_{i}", "scale") + 4 * m0 * h,
                           store.ptr(*pre, "MLP_0", f"LayerNorm_{i}", "bias") + 4 * m0 * h, mstride * h, hbuf, h,
                           ws.ptr(f"exh{i}_{tag}", (E, B, h)) if save else None,
                           ws.ptr(f"ers{i}_{tag}", (E * B,)) if save else None, E * B, h, B, True, RELU, st,
                           head=head if i == nl - 1 else None)
        else:
            ops.gemm(a_ptr, W, hbuf, bias, E, B, h, din, a_zs=a_zs, a_ms=a_ms, a_ks=1, b_zs=mstride * din * h,
                     b_ks=h, b_ns=1, c_zs=B * h, c_ms=h, bias_zs=mstride * h, relu=True, stream=st)
        a_ptr, a_zs, a_ms, din = hbuf, B * h, h, h
    if not ln:
        ops.rowdot(a_ptr, head[0], head[1], head[2], head[3], q, E * B, din, B, st)
    return q


def ensemble_bwd(cfg: NetConfig, tag, store, grads, dq_ptr, inp_ptr, ld_inp, B, ws: Workspace, st,
                 gside: "GradSide | None" = None, dinp="sum"):
    """Backward of the full ensemble from dq (E, B). Writes parameter grads when `grads` is given and
    returns ptr of dinp (B, latent+action) = sum over members.  dinp="slabs": return (ptr, nslab, slab stride) of
    the split-K partials instead (the consumer sums them on read: one launch less); dinp=None: the input gradient
    is not needed (state-based critic step), skip its GEMM."""
    pre = critic_head_prefix(cfg)
    E, ln = cfg.num_qs, cfg.critic_layer_norm
    dims = [cfg.latent + cfg.action_dim] + list(cfg.hidden_dims)
    nl = len(cfg.hidden_dims)
    hl = dims[-1]
    if grads is not None:
        stw = gside.fork() if gside is not None else st
        ops.colsum(ws.ptr(f"h{nl - 1}_{tag}", (E, B, hl)), grads.ptr(*pre, "Dense_0", "kernel"), E, B, hl, hl, hl, stw,
                   w=dq_ptr)
        ops.colsum(dq_ptr, grads.ptr(*pre, "Dense_0", "bias"), E, B, 1, 1, 1, stw)
    dy_ptr = None
    for i in range(nl - 1, -1, -1):
        h, hp = dims[i + 1], dims[i]
        hbuf = ws.ptr(f"h{i}_{tag}", (E, B, h))
        dz = ws.ptr(f"edz{i}", (E, B, h))
        g = ws.ptr(f"eg{i}", (E, B, h)) if ln else None
        xh = ws.ptr(f"exh{i}_{tag}", (E, B, h)) if ln else None
        rs = ws.ptr(f"ers{i}_{tag}", (E * B,)) if ln else None
        gam = store.ptr(*pre, "MLP_0", f"LayerNorm_{i}", "scale") if ln else None
        if i == nl - 1:
            ops.ln_act_bwd(None, 0, dq_ptr, store.ptr(*pre, "Dense_0", "kernel"), h, hbuf, h, xh, rs, gam, h, dz, g,
                           E * B, h, B, ln, RELU, st)
        else:
            ops.ln_act_bwd(dy_ptr, h, None, None, 0, hbuf, h, xh, rs, gam, h, dz, g, E * B, h, B, ln, RELU, st)
        W = store.ptr(*pre, "MLP_0", f"Dense_{i}", "kernel")
        if grads is not None:
            gb = grads.ptr(*pre, "MLP_0", f"Dense_{i}", "bias")
            stw = gside.fork() if gside is not None else st       # after this layer's ln_act_bwd
            if ln:
                ops.colsum(g, grads.ptr(*pre, "MLP_0", f"LayerNorm_{i}", "bias"), E, B, h, h, h, stw, x1=dz, out1=gb,
                           x2=g, mul2=xh, out2=grads.ptr(*pre, "MLP_0", f"LayerNorm_{i}", "scale"))
            else:
                ops.colsum(dz, gb, E, B, h, h, h, stw)
            if i == 0:
                a_ptr, a_zs, a_ks = inp_ptr, 0, ld_inp
            else:
                a_ptr, a_zs, a_ks = ws.ptr(f"h{i - 1}_{tag}", (E, B, hp)), B * hp, hp
            ops.gemm(a_ptr, dz, grads.ptr(*pre, "MLP_0", f"Dense_{i}", "kernel"), None, E, hp, h, B, a_zs=a_zs, a_ms=1,
                     a_ks=a_ks, b_zs=B * h, b_ks=h, b_ns=1, c_zs=hp * h, c_ms=h, a_kcontig=False, stream=stw)
        if i > 0:
            dy_ptr = ws.ptr(f"edh{i - 1}", (E, B, hp))
            ops.gemm(dz, W, dy_ptr, None, E, B, hp, h, a_zs=B * h, a_ms=h, a_ks=1, b_zs=hp * h, b_ks=1, b_ns=h,
                     c_zs=B * hp, c_ms=hp, b_ncontig=False, stream=st)
        elif dinp is None:
            return None
        else:
            want = dinp
            dinp = ws.ptr("dinp", (B, hp))
            ns = _ksplit(E * h)          # few (M, N) tiles and a long K = (member, feature) loop: split it
            if ns > 1 and want == "slabs":
                slab = ws.ptr("dinp_slab", (ns, B, hp))
                ops.gemm(dz, W, slab, None, 1, B, hp, E * h, seglen=h, a_ss=B * h, a_ms=h, a_ks=1, b_ss=hp * h, b_ks=1,
                         b_ns=h, c_ms=hp, b_ncontig=False, nsplit=ns, c_split_stride=B * hp, stream=st)
                return slab, ns, B * hp
            if ns > 1:
                slab = ws.ptr("dinp_slab", (ns, B, hp))
                ops.gemm(dz, W, slab, None, 1, B, hp, E * h, seglen=h, a_ss=B * h, a_ms=h, a_ks=1, b_ss=hp * h, b_ks=1,
                         b_ns=h, c_ms=hp, b_ncontig=False, nsplit=ns, c_split_stride=B * hp, stream=st)
                ops.ln_act_fwd(slab, ns, B * hp, hp, None, None, None, 0, dinp, hp, None, None, B, hp, B, False,
                               ops.ACT_NONE, st)
            else:
                ops.gemm(dz, W, dinp, None, 1, B, hp, E * h, seglen=h, a_ss=B * h, a_ms=h, a_ks=1, b_ss=hp * h, b_ks=1,
                         b_ns=h, c_ms=hp, b_ncontig=False, stream=st)
            return (dinp, 1, 0) if want == "slabs" else dinp


# =============================================================================== policy head
def policy_fwd(cfg: NetConfig, tag, store, z_ptr, ldz, B, ws: Workspace, st, key, row_offset, total_rows,
               act_ptr, ld_act, deterministic=False, save=False, want_logp=True, logp_out=None):
    """TanhNormal(MLP) on z (B, latent): writes actions to act_ptr (row stride ld_act); returns ptr of logp (B,)
    (`logp_out` when given, else a workspace buffer of the pass)."""
    pre = actor_head_prefix(cfg)
    A = cfg.action_dim
    a_ptr, a_ms, din = z_ptr, ldz, cfg.latent
    for i, h in enumerate(cfg.hidden_dims):
        hbuf = ws.ptr(f"ph{i}_{tag}", (B, h))
        ops.gemm(a_ptr, store.ptr(*pre, "MLP_0", f"Dense_{i}", "kernel"), hbuf,
                 store.ptr(*pre, "MLP_0", f"Dense_{i}", "bias"), 1, B, h, din, a_ms=a_ms, a_ks=1, b_ks=h, b_ns=1,
                 c_ms=h, relu=True, stream=st)
        a_ptr, a_ms, din = hbuf, h, h
    ms = ws.ptr(f"ms_{tag}", (B, 2 * A))
    wm, wl = store.ptr(*pre, "OutputDenseMean", "kernel"), store.ptr(*pre, "OutputDenseLogStd", "kernel")
    bm, bl = store.ptr(*pre, "OutputDenseMean", "bias"), store.ptr(*pre, "OutputDenseLogStd", "bias")
    ops.gemm(a_ptr, wm, ms, bm, 2, B, A, din, a_zs=0, a_ms=a_ms, a_ks=1, b_zs=(wl - wm) // 4, b_ks=A, b_ns=1,
             c_zs=A, c_ms=2 * A, bias_zs=(bl - bm) // 4, stream=st)
    logp = (logp_out if logp_out is not None else ws.ptr(f"logp_{tag}", (B,))) if want_logp else None
    ops.tanh_gaussian_fwd(ms, ms + 4 * A, 2 * A, act_ptr, ld_act, None,
                          ws.ptr(f"eps_{tag}", (B, A)) if save else None, logp, B, A,
                          key if key is not None else (0, 0), row_offset, total_rows, deterministic, st)
    return logp


def policy_bwd(cfg: NetConfig, tag, store, grads, z_ptr, ldz, B, ws: Workspace, st, need_dz):
    """Backward from ws['dms'] (B, 2A) = [dmean | dlogstd]; returns ptr of dz (B, latent) when need_dz."""
    pre = actor_head_prefix(cfg)
    A = cfg.action_dim
    nl = len(cfg.hidden_dims)
    dims = [cfg.latent] + list(cfg.hidden_dims)
    hl = dims[-1]
    dms = ws.ptr("dms", (B, 2 * A))
    wm, wl = store.ptr(*pre, "OutputDenseMean", "kernel"), store.ptr(*pre, "OutputDenseLogStd", "kernel")
    gwm, gwl = grads.ptr(*pre, "OutputDenseMean", "kernel"), grads.ptr(*pre, "OutputDenseLogStd", "kernel")
    hlast = ws.ptr(f"ph{nl - 1}_{tag}", (B, hl))
    ops.gemm(hlast, dms, gwm, None, 2, hl, A, B, a_zs=0, a_ms=1, a_ks=hl, b_zs=A, b_ks=2 * A, b_ns=1,
             c_zs=(gwl - gwm) // 4, c_ms=A, a_kcontig=False, stream=st)
    ops.colsum(dms, grads.ptr(*pre, "OutputDenseMean", "bias"), 1, B, A, 2 * A, 0, st)
    ops.colsum(dms + 4 * A, grads.ptr(*pre, "OutputDenseLogStd", "bias"), 1, B, A, 2 * A, 0, st)
    dh = ws.ptr(f"pdh{nl - 1}", (B, hl))
    ops.gemm(dms, wm, dh, None, 1, B, hl, 2 * A, seglen=A, a_ss=A, a_ms=2 * A, a_ks=1, b_ss=(wl - wm) // 4, b_ks=1,
             b_ns=A, c_ms=hl, b_ncontig=False, stream=st)
    for i in range(nl - 1, -1, -1):
        h, hp = dims[i + 1], dims[i]
        hbuf = ws.ptr(f"ph{i}_{tag}", (B, h))
        dz = ws.ptr(f"pdz{i}", (B, h))
        ops.ln_act_bwd(dh, h, None, None, 0, hbuf, h, None, None, None, 0, dz, None, B, h, B, False, RELU, st)
        ops.colsum(dz, grads.ptr(*pre, "MLP_0", f"Dense_{i}", "bias"), 1, B, h, h, 0, st)
        if i == 0:
            a_ptr, a_ks = z_ptr, ldz
        else:
            a_ptr, a_ks = ws.ptr(f"ph{i - 1}_{tag}", (B, hp)), hp
        ops.gemm(a_ptr, dz, grads.ptr(*pre, "MLP_0", f"Dense_{i}", "kernel"), None, 1, hp, h, B, a_ms=1, a_ks=a_ks,
                 b_ks=h, b_ns=1, c_ms=h, a_kcontig=False, stream=st)
        if i > 0 or need_dz:
            dh = ws.ptr(f"pdh{i - 1}" if i > 0 else "pdzlat", (B, hp))
            ops.gemm(dz, store.ptr(*pre, "MLP_0", f"Dense_{i}", "kernel"), dh, None, 1, B, hp, h, a_ms=h, a_ks=1,
                     b_ks=1, b_ns=h, c_ms=hp, b_ncontig=False, stream=st)
    return dh if need_dz else None
