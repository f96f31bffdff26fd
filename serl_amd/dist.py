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
        # bench.py instrumentation: with `timing` set, every device all-reduce is bracketed by events on the caller's
        # stream (RCCL) or timed on the host (gloo); `allreduce_seconds` sums them (synchronise first)
        self.timing = False
        self._pairs = []
        self._host_s = 0.0

    @property
    def allreduce_seconds(self) -> float:
        return self._host_s + sum(a.elapsed_time(b) for a, b in self._pairs) * 1e-3

    def timing_reset(self):
        self._pairs, self._host_s = [], 0.0

    def all_reduce_sum(self, tensor):
        if self.world_size > 1:
            if tensor.is_cuda and self._td.get_backend(self.group) == "gloo":
                # gloo has no device path of its own: torch stages a device tensor through pinned memory on pool streams.
                # Two ranks on one GPU (tools/rehearse_dp.sh) stalled in that staging for minutes now and then -- with the
                # update's kernels queued behind it, and at full size even with the device idle -- so the host copy is made
                # here, on the caller's stream, and gloo only ever sees host memory.  RCCL ("nccl") is not affected.
                import time
                t0 = time.perf_counter()
                host = tensor.cpu()
                self._td.all_reduce(host, op=self._td.ReduceOp.SUM, group=self.group)
                tensor.copy_(host)
                if self.timing:
                    self._host_s += time.perf_counter() - t0
                return tensor
            if self.timing and tensor.is_cuda:
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self._td.all_reduce(tensor, op=self._td.ReduceOp.SUM, group=self.group)
                b.record()
                self._pairs.append((a, b))
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
