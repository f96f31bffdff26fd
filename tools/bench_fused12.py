"""HIP-event timing of the fused layers 1 + 2 of a no-gradient encoder pass (serl_convs_fwd12) against the two
serl_convs_fwd launches it replaces, at the bench shapes (B = 256, 128 x 128 x 9 uint8 frames, BENCH_G groups); BENCH_KEEP=1 times serl_convs_fwd12_keep as the
update launches it (the last two groups keep layer 1's output and both ReLU byte masks)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serl_amd import ops  # noqa: E402

B = int(os.environ.get("BENCH_B", "256"))
G = int(os.environ.get("BENCH_G", "2"))
dev = torch.device("cuda")
st = ops.current_stream()
IH1, C1, F1, F2 = 128, 9, 32, 64
IH, OH = 63, 31
p = lambda ts: [t.data_ptr() for t in ts]
xs = [torch.randint(0, 256, (B, IH1, IH1, C1), dtype=torch.uint8, device=dev) for _ in range(G)]
w1 = [torch.randn((3, 3, C1, F1), device=dev) * 0.3 for _ in range(G)]
w2 = [torch.randn((3, 3, F1, F2), device=dev) * 0.1 for _ in range(G)]
b1 = [torch.zeros(F1, device=dev) for _ in range(G)]
b2 = [torch.zeros(F2, device=dev) for _ in range(G)]
wf1 = [torch.empty(ops.convs_weight_bytes(True, 3, 3, C1, F1) // 2, dtype=torch.int16, device=dev) for _ in range(G)]
wf2 = [torch.empty(ops.convs_weight_bytes(False, 3, 3, F1, F2) // 2, dtype=torch.int16, device=dev) for _ in range(G)]
ops.convs_prep_weights(p(w1), p(wf1), None, True, 3, 3, C1, F1, st)
ops.convs_prep_weights(p(w2), p(wf2), None, False, 3, 3, F1, F2, st)
y1 = [torch.empty(B * IH * IH * F1 * 2, dtype=torch.int16, device=dev) for _ in range(G)]
y2 = [torch.empty(B * OH * OH * F2 * 2, dtype=torch.int16, device=dev) for _ in range(G)]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def two():
    ops.convs_fwd(p(xs), p(wf1), p(b1), p(y1), True, B, IH1, IH1, C1, 3, 3, 2, F1, True, st)
    ops.convs_fwd(p(y1), p(wf2), p(b2), p(y2), False, B, IH, IH, F1, 3, 3, 2, F2, True, st)


if os.environ.get("BENCH_KEEP"):      # as the update's 4-group launch: the last two groups (the critic pass) keep layer 1's output + masks
    rm1 = [torch.empty(B * IH * IH * F1 // 8, dtype=torch.uint8, device=dev) for _ in range(2)]
    rm2 = [torch.empty(B * OH * OH * F2 // 8, dtype=torch.uint8, device=dev) for _ in range(2)]
    ky1 = [0] * (G - 2) + p(y1[:2]); km1 = [0] * (G - 2) + p(rm1); km2 = [0] * (G - 2) + p(rm2)
    print(f"G={G} keep: fused {timeit(lambda: ops.convs_fwd12_keep(p(xs), p(wf1), p(b1), p(wf2), p(b2), p(y2), B, IH1, IH1, C1, F1, F2, True, st, rmasks=km2, y1s=ky1, rmask1s=km1)):7.1f} us")
    sys.exit(0)
print(f"G={G}: two launches {timeit(two):7.1f} us   fused {timeit(lambda: ops.convs_fwd12(p(xs), p(wf1), p(b1), p(wf2), p(b2), p(y2), B, IH1, IH1, C1, F1, F2, True, st)):7.1f} us")
