"""Data-parallel glue: one process per GPU, gradients summed with one all-reduce per critic step over
RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for CPU tests).

The reference has no multi-device path; this follows its commented-out batch split
(serl/data/replay_buffer.py:92-98): every rank holds the full parameters, owns rows
[rank*B, (rank+1)*B) of each UTD minibatch of the global batch, and all random draws are keyed by
the *global* row index so the result does not depend on the number of ranks."""
from __future__ import annotations


class DataParallel:
    def __init__(self, group=None):
        import torch.distributed as td
        if not td.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._td = td
        self.group = group
        self.rank = td.get_rank(group)
        self.world_size = td.get_world_size(group)

    def all_reduce_sum(self, tensor):
        if self.world_size > 1:
            if tensor.is_cuda and self._td.get_backend(self.group) == "gloo":
                # gloo has no device path of its own: torch stages a device tensor through pinned memory on pool streams.
                # Two ranks on one GPU (tools/rehearse_dp.sh) stalled in that staging for minutes now and then -- with the
                # update's kernels queued behind it, and at full size even with the device idle -- so the host copy is made
                # here, on the caller's stream, and gloo only ever sees host memory.  RCCL ("nccl") is not affected.
                host = tensor.cpu()
                self._td.all_reduce(host, op=self._td.ReduceOp.SUM, group=self.group)
                tensor.copy_(host)
                return tensor
            self._td.all_reduce(tensor, op=self._td.ReduceOp.SUM, group=self.group)
        return tensor


def shard_rows(n_global_rows: int, utd_ratio: int, rank: int, world_size: int):
    """Row indices (into the global B_g*U batch) that `rank` owns: inside every contiguous UTD slice of
    B_g rows (serl/agents/sac/sac_learner.py:227-232) the rank's contiguous block of B_g/world rows."""
    import numpy as np
    assert n_global_rows % (utd_ratio * world_size) == 0
    bg = n_global_rows // utd_ratio
    b = bg // world_size
    return np.concatenate([np.arange(i * bg + rank * b, i * bg + (rank + 1) * b) for i in range(utd_ratio)])
