"""Device-side batch leaves produced by the replay iterators."""
from __future__ import annotations


class FrameWindows:
    """Lazy pixel leaf of a batch sampled from a device-resident frame store: sample j is the window of
    store rows indx[j]-S .. indx[j] (memory_efficient_replay_buffer.py:155-160).  DrQLearner.update
    fuses the gather with `_unpack` and the random-shift crop in one kernel; `materialize()` returns
    the packed (n,H,W,3,S+1) uint8 tensor the reference would have produced."""

    def __init__(self, frames, capacity, indx, num_stack, first_rows=None, lock=None):
        self.frames, self.capacity, self.indx, self.num_stack = frames, int(capacity), indx, int(num_stack)
        # the owning buffer's lock: insert() holds it while it pins batches and overwrites a row; the learner holds it
        # from the moment it reads `packed` until the gather is enqueued and `consumed` is set (hold_windows below)
        self.lock = lock
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


def hold_windows(leaves):
    """Context manager: the locks of the device replay buffers that own the lazy FrameWindows among `leaves`, held
    together.  An update() enters it before it looks at `packed` and leaves it after the native call has enqueued the
    gather: an insert() from another thread (the reference's replay thread + learner thread pattern) then either pins
    the batch before the update reads it, or overwrites the row after the gather is in the stream.  Ordering on the
    DEVICE is by stream: insert() copies on torch's current stream and update() launches on torch's current stream, so
    both threads must use the same stream (the default one unless the caller changes it)."""
    import contextlib
    stack = contextlib.ExitStack()
    seen = []
    for leaf in leaves:
        lock = getattr(leaf, "lock", None) if isinstance(leaf, FrameWindows) else None
        if lock is not None and all(lock is not l for l in seen):
            seen.append(lock)
    for lock in sorted(seen, key=id):           # a fixed order: two learners on two buffers cannot deadlock
        stack.enter_context(lock)
    return stack
