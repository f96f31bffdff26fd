"""2-rank gloo all_reduce of CUDA tensors of growing size on ONE GPU (is the rehearsal path itself sound?)."""
import os, sys, time, faulthandler
import torch, torch.distributed as td
faulthandler.enable()
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
td.init_process_group("gloo")
for n in (1 << 10, 1 << 16, 1 << 20, 5_250_000):
    x = torch.full((n,), float(rank + 1), device="cuda")
    t0 = time.time()
    td.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank} n={n} sum={x[0].item()} {time.time() - t0:.3f}s", flush=True)
td.barrier()
print("done", flush=True)
