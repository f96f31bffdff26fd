"""Phase timestamps of the two-set fused layers-1+2 kernel (library built with -DPFPP_TRACE, tools/build_variant.sh):
block 0, slots 8..11, one row per wave: cycles since the slot's earliest start at the trace points
0 slot start | L1 role: 1 halo row copied, 2 tiles done | L2 role: 1 DMA issued, 2 K loop done, 3 stores issued, 4 DMA landed | 5 past the barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serl_amd import ops, _lib  # noqa: E402

B, G = 256, 2
dev = torch.device("cuda")
st = ops.current_stream()
IH1, C1, F1, F2 = 128, 9, 32, 64
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
y2 = [torch.empty(B * 31 * 31 * F2 * 2, dtype=torch.int16, device=dev) for _ in range(G)]
for _ in range(3):
    ops.convs_fwd12(p(xs), p(wf1), p(b1), p(wf2), p(b2), p(y2), B, IH1, IH1, C1, F1, F2, True, st)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["SERL_AMD_LIB"])
out = (ctypes.c_uint32 * 256)()
assert lib.serl_debug_pf_trace(out) == 0
t = np.array(out, dtype=np.int64).reshape(4, 8, 8)
for k in range(4):
    t0 = t[k, :, 0].min()
    print(f"slot {8 + k}")
    for w in range(8):
        role = "L1" if ((8 + k) & 1) == (w >> 2) else "L2"
        print(f"  wave {w} {role}: " + " ".join(f"{int(v - t0):7d}" for v in t[k, w, :6]))
