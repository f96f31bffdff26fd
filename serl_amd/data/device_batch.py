"""Device-side batch leaves produced by the replay iterators."""
from __future__ import annotations


class FrameWindows:
    """Lazy pixel leaf of a batch sampled from a device-resident frame store: sample j is the window of
    store rows indx[j]-S .. indx[j] (memory_efficient_replay_buffer.py:155-160).  DrQLearner.update
    fuses the gather with `_unpack` and the random-shift crop in one kernel; `materialize()` returns
    the packed (n,H,W,3,S+1) uint8 tensor the reference would have produced."""

    def __init__(self, frames, capacity, indx, num_stack):
        self.frames, self.capacity, self.indx, self.num_stack = frames, int(capacity), indx, int(num_stack)

    @property
    def shape(self):
        n = self.indx.shape[0]
        _, H, W, C = self.frames.shape
        return (n, H, W, C, self.num_stack + 1)

    def __len__(self):
        return self.indx.shape[0]

    def materialize(self):
        import torch
        from .. import ops
        n, H, W, C, T = self.shape
        out = torch.empty((n, H, W, C, T), dtype=torch.uint8, device=self.frames.device)
        ops.gather_window(self.frames.data_ptr(), self.capacity, self.indx.data_ptr(), out.data_ptr(), n, H * W * C,
                          self.num_stack, ops.current_stream())
        return out
