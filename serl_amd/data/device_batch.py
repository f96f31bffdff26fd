"""Device-side batch leaves produced by the replay iterators."""
from __future__ import annotations


class FrameWindows:
    """Lazy pixel leaf of a batch sampled from a device-resident frame store: sample j is the window of
    store rows indx[j]-S .. indx[j] (memory_efficient_replay_buffer.py:155-160).  DrQLearner.update
    fuses the gather with `_unpack` and the random-shift crop in one kernel; `materialize()` returns
    the packed (n,H,W,3,S+1) uint8 tensor the reference would have produced."""

    def __init__(self, frames, capacity, indx, num_stack, first_rows=None):
        self.frames, self.capacity, self.indx, self.num_stack = frames, int(capacity), indx, int(num_stack)
        # The reference gathers pixels at sample time, so its batches cannot change afterwards.  A lazy window batch
        # can: an insert() that lands on one of its rows before the learner has enqueued the gather would change it.
        # The owning buffer therefore keeps un-consumed batches registered and, before overwriting a row that one of
        # them references (`first_rows`: host int64, first store row of every window), calls pin(): the packed
        # tensor is gathered then, and DrQLearner.update reads that instead of the store.
        self.first_rows = first_rows
        self.packed = None
        self.consumed = False

    def touches(self, row) -> bool:
        if self.first_rows is None:
            return False
        d = int(row) - self.first_rows
        return bool(((d >= 0) & (d <= self.num_stack)).any())

    def pin(self):
        if self.packed is None and not self.consumed:
            self.packed = self.materialize()
        return self.packed

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
