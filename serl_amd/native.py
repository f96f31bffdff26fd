"""Python face of the native step driver (serl_amd/csrc/driver.hip; include/serl_hip.h "Native step driver").

One `NativeLearner` per `create()` call, shared by every agent copy derived from it (agents are immutable value
objects over device buffers; the handle owns streams, events and the activation workspace).  `update()` makes ONE
C-ABI call that enqueues the whole update; `act()` one call per action."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import torch

from . import _lib
from .networks.spec import NetConfig, actor_layout, critic_layout

_DP_TRACE = os.environ.get("SERL_AMD_DP_TRACE", "0") != "0"     # a stderr line around every all-reduce of update()

c_fp = C.POINTER(C.c_float)


class LearnerConfig(C.Structure):
    _fields_ = [("action_dim", C.c_int), ("num_hidden", C.c_int), ("hidden_dims", C.c_int * 4), ("num_qs", C.c_int),
                ("num_min_qs", C.c_int), ("critic_layer_norm", C.c_int), ("backup_entropy", C.c_int),
                ("has_temperature", C.c_int), ("pixel", C.c_int), ("obs_dim", C.c_int), ("num_cameras", C.c_int),
                ("height", C.c_int), ("width", C.c_int), ("stack", C.c_int), ("state_dim", C.c_int),
                ("latent_dim", C.c_int), ("num_conv", C.c_int), ("cnn_features", C.c_int * 8),
                ("cnn_filters", C.c_int * 8), ("cnn_strides", C.c_int * 8), ("aug_rank", C.c_int * 4),
                ("discount", C.c_float), ("tau", C.c_float), ("target_entropy", C.c_float), ("actor_lr", C.c_float),
                ("critic_lr", C.c_float), ("temp_lr", C.c_float), ("encoder", C.c_int), ("feat_h", C.c_int),
                ("feat_w", C.c_int), ("feat_c", C.c_int), ("encoder_dropout", C.c_float), ("critic_no_action", C.c_int),
                ("critic_dropout", C.c_float)]


class AgentBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("actor", "actor_mu", "actor_nu", "critic", "critic_mu", "critic_nu", "target",
                                          "temp", "critic_grads", "actor_grads")] + \
               [("actor_step", C.c_int), ("critic_step", C.c_int), ("temp_step", C.c_int)]


class PixelSource(C.Structure):
    _fields_ = [("kind", C.c_int), ("frames", C.c_void_p), ("capacity", C.c_longlong), ("indx", C.c_void_p),
                ("obs", C.c_void_p), ("next_obs", C.c_void_p)]


class UpdateBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("aug_mode", C.c_int), ("actions", C.c_void_p), ("rewards", C.c_void_p),
                ("masks", C.c_void_p), ("obs", C.c_void_p), ("next_obs", C.c_void_p), ("state", C.c_void_p),
                ("next_state", C.c_void_p), ("pixels", PixelSource * 4), ("features_hook", C.c_int),
                ("use_sac_rng", C.c_int), ("sac_rng", C.c_uint * 2)]


class BcBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("actor", "actor_mu", "actor_nu", "actor_grads", "disc", "disc_mu", "disc_nu",
                                          "disc_grads")] + [("actor_step", C.c_int), ("disc_step", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)

KIND_FRAMES, KIND_PACKED, KIND_UNPACKED = 0, 1, 2


def _offsets(layout):
    offs = [off for off, _ in layout.entries.values()] + [layout.size]
    return (C.c_longlong * len(offs))(*offs), len(offs)


class NativeLearner:
    def __init__(self, cfg: NetConfig, *, num_min_qs, backup_entropy, has_temperature, discount, tau, target_entropy,
                 actor_lr, critic_lr, temp_lr, critic_layout_fn=critic_layout, critic_no_action=False,
                 critic_dropout_rate=None):
        if len(cfg.hidden_dims) > 4:
            raise ValueError("at most 4 hidden layers")
        lc = LearnerConfig()
        lc.action_dim, lc.num_hidden, lc.num_qs = cfg.action_dim, len(cfg.hidden_dims), cfg.num_qs
        for i, h in enumerate(cfg.hidden_dims):
            lc.hidden_dims[i] = int(h)
        lc.num_min_qs = int(num_min_qs or 0)
        lc.critic_layer_norm, lc.backup_entropy = int(cfg.critic_layer_norm), int(bool(backup_entropy))
        lc.has_temperature = int(bool(has_temperature))
        lc.pixel, lc.obs_dim = int(cfg.pixel), int(cfg.obs_dim)
        if cfg.pixel:
            cams = cfg.cams
            if len(cams) > 4 or len(cfg.cnn_features) > 8:
                raise ValueError("at most 4 cameras and 8 conv layers")
            lc.num_cameras, lc.height, lc.width, lc.stack = len(cams), cfg.H, cfg.W, cfg.stack
            lc.state_dim, lc.latent_dim, lc.num_conv = cfg.state_dim, cfg.latent_dim, len(cfg.cnn_features)
            for i, (f, k, s) in enumerate(zip(cfg.cnn_features, cfg.cnn_filters, cfg.cnn_strides)):
                lc.cnn_features[i], lc.cnn_filters[i], lc.cnn_strides[i] = int(f), int(k), int(s)
            lc.encoder = {"d4pg": 0, "mobilenet": 1}[cfg.encoder]
            lc.feat_h, lc.feat_w, lc.feat_c, lc.encoder_dropout = cfg.feat_h, cfg.feat_w, cfg.feat_c, cfg.encoder_dropout
            for ci, cam in enumerate(cams):           # augmentation keys are split in the caller's key order
                lc.aug_rank[ci] = cfg.pixel_keys.index(cam)
        lc.discount, lc.tau, lc.target_entropy = float(discount), float(tau), float(target_entropy)
        lc.actor_lr, lc.critic_lr, lc.temp_lr = float(actor_lr or 0.0), float(critic_lr or 0.0), float(temp_lr or 0.0)
        lc.critic_no_action = int(bool(critic_no_action))      # hybrid BC: the "critic" slot is the discrete actor
        lc.critic_dropout = float(critic_dropout_rate or 0.0)
        co, nco = _offsets(critic_layout_fn(cfg))
        ao, nao = _offsets(actor_layout(cfg))
        h = C.c_void_p()
        _lib.call("serl_learner_create", C.byref(lc), co, nco, ao, nao, C.byref(h))
        self._h = h
        self.cfg = cfg
        self._cb = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().serl_learner_destroy(h)
            except Exception:       # noqa: BLE001  interpreter shutdown
                pass

    def _comm_stream(self):
        """torch view of the handle's communication stream (serl_learner_comm_stream)."""
        if getattr(self, "_comm", None) is None:
            p = C.c_void_p()
            _lib.call("serl_learner_comm_stream", self._h, C.byref(p))
            self._comm = torch.cuda.ExternalStream(p.value)
        return self._comm

    def set_backbone(self, backbone):
        """The frozen MobileNetV3 feature extractor (serl_amd.networks.mobilenet.Backbone) of a mobilenet-encoder learner."""
        self._backbone = backbone          # keeps the device buffer alive
        _lib.call("serl_learner_set_backbone", self._h, backbone.flat.data_ptr())

    # ------------------------------------------------------------------ update
    def update(self, buffers: AgentBuffers, batch: UpdateBatch, utd_ratio, rng, info_out, stream, dist=None,
               reduce_tensors=None, features_hook=None):
        """-> new rng (uint32[2]).  reduce_tensors: (critic_grads, actor_grads, info[:4], None, critic_grads suffix behind the
        image_encoder leaves, image_encoder prefix) torch tensors for `dist` (serl_allreduce_fn's `which`).
        features_hook(): called once the augmented images / backbone features of all rows are enqueued (VICE rewards)."""
        rin = (C.c_uint * 2)(int(rng[0]), int(rng[1]))
        rout = (C.c_uint * 2)()
        cb, rank, world = C.cast(None, ALLREDUCE_FN), 0, 1
        err = []
        multi = dist is not None and dist.world_size > 1
        if multi:
            rank, world = dist.rank, dist.world_size
        batch.features_hook = int(features_hook is not None)
        if multi or features_hook is not None:
            def hook(_ctx, which):
                try:
                    if which == 3:
                        features_hook()
                    elif which == 4:        # trunk + ensemble bucket: on the driver's comm stream, next to the encoder backward
                        with torch.cuda.stream(self._comm_stream()):
                            dist.all_reduce_sum(reduce_tensors[4])
                    else:
                        if _DP_TRACE:
                            print(f"[dp rank {rank}] all-reduce {which} ({reduce_tensors[which].numel()} elements) ...", file=sys.stderr, flush=True)
                        dist.all_reduce_sum(reduce_tensors[which])
                        if _DP_TRACE:
                            print(f"[dp rank {rank}] all-reduce {which} done", file=sys.stderr, flush=True)
                    return 0
                except Exception as e:      # noqa: BLE001  surfaced below; a C caller only sees the status
                    err.append(e)
                    return 2
            cb = ALLREDUCE_FN(hook)
        status = _lib.load().serl_learner_update(self._h, C.byref(buffers), C.byref(batch), int(utd_ratio), rin, rout,
                                                 info_out, stream, cb, None, rank, world)
        if err:
            raise err[0]
        _lib.check(status, "serl_learner_update")
        return np.array([rout[0], rout[1]], dtype=np.uint32)

    # ------------------------------------------------------------------ behaviour cloning
    def bc_update(self, buffers: "BcBuffers", batch: UpdateBatch, ld_actions, key_aug, key1, key2, info_out, stream):
        ks = [(C.c_uint * 2)(int(k[0]), int(k[1])) for k in (key_aug, key1, key2)]
        _lib.call("serl_learner_bc_update", self._h, C.byref(buffers), C.byref(batch), int(ld_actions), ks[0], ks[1], ks[2],
                  info_out, stream)

    def bc_logit(self, disc_params_ptr, images, state_ptr, out_ptr, stream):
        imgs = (C.c_void_p * len(images))(*images)
        _lib.call("serl_learner_bc_logit", self._h, disc_params_ptr, imgs, state_ptr, out_ptr, stream)

    # ------------------------------------------------------------------ acting
    def act(self, actor_params_ptr, images, state_or_obs_ptr, key, deterministic, act_out_ptr, eps_out_ptr, stream):
        imgs = None
        if images is not None:
            imgs = (C.c_void_p * len(images))(*images)
        k = None if key is None else (C.c_uint * 2)(int(key[0]), int(key[1]))
        _lib.call("serl_learner_act", self._h, actor_params_ptr, imgs, state_or_obs_ptr, k, int(bool(deterministic)),
                  act_out_ptr, eps_out_ptr, stream)

    # ------------------------------------------------------------------ tooling
    def read_buffer(self, B, name, shape, dtype=torch.float32, device=None):
        out = torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=device or "cuda")
        _lib.call("serl_learner_read_buffer", self._h, int(B), name.encode(), out.data_ptr(),
                  out.numel() * out.element_size(), torch.cuda.current_stream().cuda_stream)
        return out

    def buffer_ptr(self, B, name):
        """(device address, bytes) of a workspace buffer of the update at minibatch size B."""
        p, n = C.c_void_p(), C.c_longlong()
        _lib.call("serl_learner_buffer_ptr", self._h, int(B), name.encode(), C.byref(p), C.byref(n))
        return p.value, n.value

    def workspace_bytes(self) -> int:
        return int(_lib.load().serl_learner_workspace_bytes(self._h))

    def set_timing(self, conv_events=False, phase_events=False):
        _lib.call("serl_learner_set_timing", self._h, int(conv_events), int(phase_events))

    def timing_reset(self):
        _lib.call("serl_learner_timing_reset", self._h)

    def conv_timings(self):
        """[(kind, in_u8, groups, B, IH, C, F), ms, flops, bytes] of every conv launch since the last reset (after a sync)."""
        lib = _lib.load()
        kinds = ("convs_fwd", "convs_dgrad", "convs_wgrad", "convs_fwd12", "convs_dgw1", "convs_fwd12keep")
        out = []
        desc, val = (C.c_int * 7)(), (C.c_double * 3)()
        for i in range(lib.serl_learner_timing_count(self._h)):
            _lib.call("serl_learner_timing_get", self._h, i, desc, val)
            out.append(((kinds[desc[0]], bool(desc[1]), desc[2], desc[3], desc[4], desc[5], desc[6]), val[0], val[1], val[2]))
        return out

    def phase_timings(self):
        """[(mark name, ms to the next mark)] of the updates since the last reset (after a sync)."""
        lib = _lib.load()
        out = []
        name, ms = C.c_char_p(), C.c_double()
        for i in range(lib.serl_learner_phase_count(self._h)):
            _lib.call("serl_learner_phase_get", self._h, i, C.byref(name), C.byref(ms))
            out.append((name.value.decode(), ms.value))
        return out
