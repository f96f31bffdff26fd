"""Per-launch floor of a chain of tiny dependent kernels on one stream: eager ctypes launches vs. a captured hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from serl_amd import ops

x = torch.zeros(1024, device="cuda")
N = 200
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = ops.current_stream()
    def chain():
        for i in range(N):
            ops.fill(x.data_ptr(), float(i), 1024, st)
    chain(); torch.cuda.synchronize()
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record(); chain(); b.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
        print(f"eager: host {1e6 * (t1 - t0) / N:.2f} us/launch, gpu {1e3 * a.elapsed_time(b) / N:.2f} us/launch")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = ops.current_stream()
        chain()
    g.replay(); torch.cuda.synchronize()
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record(); g.replay(); b.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
        print(f"graph: host {1e6 * (t1 - t0) / N:.2f} us/launch, gpu {1e3 * a.elapsed_time(b) / N:.2f} us/launch")
