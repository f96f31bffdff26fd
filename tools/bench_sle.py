"""HIP-event timing of sle_fwd at the bench shapes (4 groups, B=256, 7x7x256 S-format features)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from serl_amd import ops
G, B, HW, C = int(os.environ.get("SLE_G", "4")), 256, 49, 256
dev = "cuda"
feat = [torch.randint(0, 30000, (B * HW * 2 * C,), dtype=torch.int16, device=dev) for _ in range(G)]
ker = [torch.randn(HW * C * 8, device=dev) for _ in range(G)]
out = [torch.zeros(B, 4096, device=dev) for _ in range(G)]
st = ops.current_stream()
fn = lambda: ops.sle_fwd([t.data_ptr() for t in feat], [t.data_ptr() for t in ker], [t.data_ptr() for t in out], 4096, B, HW, C, 8, st, feat_split=True)
for _ in range(5): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): fn()
b.record(); torch.cuda.synchronize()
print(f"sle_fwd G={G}: {a.elapsed_time(b) / 50 * 1e3:.1f} us")

# data gradient (critic pass: 2 groups, ReLU mask from the features) and kernel gradient
G2 = 2
dout = [torch.randn(B, 4096, device=dev) for _ in range(G2)]
dfeat = [torch.empty(B * HW * 2 * C, dtype=torch.int16, device=dev) for _ in range(G2)]
dker = [torch.empty(HW * C * 8, device=dev) for _ in range(G2)]
p = lambda ts: [t.data_ptr() for t in ts]
for name, fn2 in (("sle_bwd data", lambda: ops.sle_bwd(p(dout), 4096, p(ker[:G2]), p(feat[:G2]), p(dfeat), None, B, HW, C, 8, st, feat_split=True)),
                  ("sle_bwd kernel", lambda: ops.sle_bwd(p(dout), 4096, p(ker[:G2]), p(feat[:G2]), None, p(dker), B, HW, C, 8, st, feat_split=True))):
    for _ in range(5): fn2()
    torch.cuda.synchronize()
    a.record()
    for _ in range(50): fn2()
    b.record(); torch.cuda.synchronize()
    print(f"{name} G={G2}: {a.elapsed_time(b) / 50 * 1e3:.1f} us")
