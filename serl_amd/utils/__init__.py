"""serl_amd.utils: the reference's `serl.utils` package layout (commons, augmentations) + checkpoints."""
from . import augmentations, commons
from .augmentations import batched_random_crop, random_crop
from .checkpoints import restore_checkpoint, restore_checkpoint_, save_checkpoint
from .commons import _reset_weights, _share_encoder, _unpack, ema, get_data

__all__ = ["commons", "augmentations", "_unpack", "_share_encoder", "get_data", "_reset_weights", "ema",
           "save_checkpoint", "restore_checkpoint", "restore_checkpoint_", "random_crop", "batched_random_crop"]
