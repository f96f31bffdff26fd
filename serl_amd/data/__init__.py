from .dataset import Dataset
from .device_batch import FrameWindows
from .memory_efficient_replay_buffer import MemoryEfficientReplayBuffer
from .replay_buffer import GPUReplayBuffer, ReplayBuffer

__all__ = ["Dataset", "ReplayBuffer", "GPUReplayBuffer", "MemoryEfficientReplayBuffer", "FrameWindows"]
