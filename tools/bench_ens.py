"""HIP-event micro-benchmark of the fused critic-ensemble kernels (csrc/ensemble.hip) stand-alone: forward, backward and
the column sums at the headline head shapes (E members, batch B, input width din).  BENCH_E / BENCH_B / BENCH_DIN."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from serl_amd import _lib, ops

E, B, din = int(os.environ.get("BENCH_E", 10)), int(os.environ.get("BENCH_B", 256)), int(os.environ.get("BENCH_DIN", 23))
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
inp = r(B, din)
w0, b0, g0s, be0 = r(E, din, 256) * 0.1, r(E, 256) * 0.1, 1 + 0.1 * r(E, 256), 0.1 * r(E, 256)
w1, b1, g1s, be1 = r(E, 256, 256) * 0.06, r(E, 256) * 0.1, 1 + 0.1 * r(E, 256), 0.1 * r(E, 256)
hw, hb = r(E, 256) * 0.06, r(E) * 0.1
h0, h1, xh0, xh1 = (torch.empty(E, B, 256, device=dev) for _ in range(4))
rs0, rs1, q = (torch.empty(E, B, device=dev) for _ in range(3))
dq = r(E, B) / (E * B)
dz1, g1, dz0, g0 = (torch.empty(E, B, 256, device=dev) for _ in range(4))
outs = [torch.empty(E, 256, device=dev) for _ in range(7)] + [torch.empty(E, device=dev)]
st = ops.current_stream()
p = lambda t: t.data_ptr()


def fwd():
    _lib.call("serl_ensemble_fwd_fused", p(inp), din, din, B, E, p(w0), p(b0), p(g0s), p(be0), p(w1), p(b1), p(g1s), p(be1),
              p(hw), p(hb), din * 256, 256 * 256, 256, 1, 1, p(h0), p(h1), p(xh0), p(xh1), p(rs0), p(rs1), p(q), st)


def bwd():
    _lib.call("serl_ensemble_bwd_fused", p(dq), B, E, p(hw), p(w1), p(g0s), p(g1s), 1, p(h0), p(h1), p(xh0), p(xh1), p(rs0),
              p(rs1), p(dz1), p(g1), p(dz0), p(g0), st)


def colsum():
    _lib.call("serl_ensemble_colsum", p(dq), p(h1), p(g1), p(xh1), p(dz1), p(g0), p(xh0), p(dz0), B, E, 1, p(outs[0]),
              p(outs[7]), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), p(outs[6]), None, 0, 0, None, st)


for name, fn in (("fwd", fwd), ("bwd", bwd), ("colsum", colsum)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"ensemble {name:7s} E={E} B={B} din={din}: {e0.elapsed_time(e1) / n * 1e3:7.2f} us per launch (back-to-back)")
