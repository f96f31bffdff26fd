"""Mirrors serl/utils/augmentations.py:5-11,44-48 (`random_crop`, `batched_random_crop`) on the device: the offsets
are jax.random.randint(key, (2,), 0, 2*padding+1) (threefry, bit-exact) and the pad-by-edge + dynamic_slice is the
uint8 gather kernel `DrQLearner.update` uses (serl_gather_crop_packed).  `random_crop_jitter` (:14-42) is dead code in
the reference (it needs the un-imported dm_pix) and is not provided."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def _to_device_u8(imgs):
    if isinstance(imgs, torch.Tensor):
        t = imgs
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(imgs)))
    if t.dtype != torch.uint8:
        raise TypeError(f"random_crop: uint8 images expected, got {t.dtype}")
    if not torch.cuda.is_available():
        raise RuntimeError("serl_amd needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    return t.to("cuda").contiguous()


def _crop(imgs, off, padding):
    """imgs (n,H,W,3,S) uint8 device, off (n,2) int32 device -> cropped copy (same shape)."""
    n, H, W, C, S = imgs.shape
    if C != 3:
        raise ValueError(f"random_crop: images must be (H, W, 3, stack), got {tuple(imgs.shape[1:])}")
    out = torch.empty_like(imgs)
    # (H,W,3,S) row-major == (H,W,3*S) with channel index c*S+s: the kernel's output layout; T = S frames, first one 0
    ops.gather_crop_packed(imgs.data_ptr(), S, 0, off.data_ptr(), out.data_ptr(), n, H, W, S, int(padding),
                           ops.current_stream())
    return out


def random_crop(key, img, padding):
    """augmentations.py:5-11: one (H,W,3,S) uint8 image shifted by randint(key, (2,), 0, 2*padding+1) - padding."""
    x = _to_device_u8(img)[None]
    off = torch.empty((1, 2), dtype=torch.int32, device=x.device)
    ops.prng_randint(key, 2, 0, 2 * int(padding) + 1, off.data_ptr(), None, ops.current_stream())
    return _crop(x, off, padding)[0]


def batched_random_crop(key, obs, pixel_key, padding=4):
    """augmentations.py:44-48: keys = split(key, n); every image its own shift; returns obs with the key replaced."""
    x = _to_device_u8(obs[pixel_key])
    n = x.shape[0]
    off = torch.empty((n, 2), dtype=torch.int32, device=x.device)
    ops.crop_offsets(key, n, 0, n, int(padding), off.data_ptr(), ops.current_stream())
    imgs = _crop(x, off, padding)
    if hasattr(obs, "copy") and not isinstance(obs, dict):
        return obs.copy(add_or_replace={pixel_key: imgs})
    out = dict(obs)
    out[pixel_key] = imgs
    return out
