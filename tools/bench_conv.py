"""Micro-benchmark of the igemm conv launches at the headline shapes (B=256, E1 encoder): TFLOP/s per
launch, measured with HIP events.  python tools/bench_conv.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serl_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = os.environ.get("BENCH_ONLY", "")        # e.g. "conv3 fwd,conv2 dgrad": run only these launches
NITER = int(os.environ.get("BENCH_N", "20"))
FWD_G = int(os.environ.get("BENCH_FWD_G", "4"))   # forward groups per launch: 2 cameras x (target, critic) passes
MODE = os.environ.get("BENCH_MODE", "s")         # "s": S-format entry points (serl_convs_*); "f": serl_conv2d_*
dev = torch.device("cuda")
if os.environ.get("BENCH_PATCH") is not None:   # 0: im2col kernels only, 1 (library default): patch kernels where they fit
    ops.convs_set_patch_kernels(int(os.environ["BENCH_PATCH"]))
LAYERS = [(128, 9, 32), (63, 32, 64), (31, 64, 128), (15, 128, 256)]   # IH, C, F  (3x3, stride 2)
st = ops.current_stream()


def timeit(fn, n=NITER):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = 0.0
for li, (IH, C, F) in enumerate(LAYERS):
    OH = (IH - 3) // 2 + 1
    flop1 = 2.0 * B * OH * OH * F * 9 * C
    for G, what in ((FWD_G, "fwd"), (2, "wgrad"), (2, "dgrad")):
        if what == "dgrad" and li == 0:
            continue
        if ONLY and f"conv{li + 1} {what}" not in ONLY.split(","):
            continue
        if li == 0:
            xs = [torch.randint(0, 256, (B, IH, IH, C), dtype=torch.uint8, device=dev) for _ in range(G)]
        else:
            xs = [torch.rand((B, IH, IH, C), device=dev) for _ in range(G)]
        ws = [torch.randn((3, 3, C, F), device=dev) * 0.05 for _ in range(G)]
        bs = [torch.zeros(F, device=dev) for _ in range(G)]
        ys = [torch.empty((B, OH, OH, F), device=dev) for _ in range(G)]
        dz = [torch.randn((B, OH, OH, F), device=dev) for _ in range(G)]
        p = lambda ts: [t.data_ptr() for t in ts]

        def to_s(t, rows, cols):
            buf = torch.empty(rows * cols * 2, dtype=torch.int16, device=dev)
            ops.split_f32(t.data_ptr(), buf.data_ptr(), rows, cols, st)
            return buf
        if MODE == "s":
            u8 = li == 0
            xS = xs if u8 else [to_s(x, B * IH * IH, C) for x in xs]
            dzS = [to_s(z, B * OH * OH, F) for z in dz]
            wb = ops.convs_weight_bytes(u8, 3, 3, C, F)
            wf = [torch.empty(wb // 2, dtype=torch.int16, device=dev) for _ in range(G)]
            wd = [torch.empty(9 * C * F * 2, dtype=torch.int16, device=dev) for _ in range(G)]
            ops.convs_prep_weights(p(ws), p(wf), p(wd), u8, 3, 3, C, F, st)
            yS = [torch.empty(B * OH * OH * F * 2, dtype=torch.int16, device=dev) for _ in range(G)]
            if what == "fwd":
                rm = None
                if not os.environ.get("BENCH_NO_RMASK"):     # 2 of the groups (the critic pass) write ReLU byte masks
                    rmt = [torch.empty(B * OH * OH * F // 8, dtype=torch.uint8, device=dev) for _ in range(2)]
                    rm = p(rmt) + [0] * (G - 2)
                fn = lambda: ops.convs_fwd(p(xS), p(wf), p(bs), p(yS), u8, B, IH, IH, C, 3, 3, 2, F, True, st, rmasks=rm)
            elif what == "wgrad":
                nb = ops.conv2d_wgrad_workspace_bytes(G, B, IH, IH, C, 3, 3, 2, F)
                wk = torch.empty(nb // 4, device=dev)
                dws = [torch.empty_like(w) for w in ws]
                fn = lambda: ops.convs_wgrad(p(xS), p(dzS), p(dws), p(bs), wk.data_ptr(), nb, u8, B, IH, IH, C, 3, 3,
                                             2, F, st)
            else:
                dxS = [torch.empty(B * IH * IH * C * 2, dtype=torch.int16, device=dev) for _ in range(G)]
                if not (os.environ.get("BENCH_HIMASK") or os.environ.get("BENCH_NOMASK")):     # as the learner: byte masks
                    bm = [torch.randint(0, 256, (B * IH * IH * C // 8,), dtype=torch.uint8, device="cuda") for _ in range(G)]
                    fn = lambda: ops.convs_dgrad(p(dzS), p(wd), p(bm), p(dxS), B, IH, IH, C, 3, 3, 2, F, st, xact_bits=True)
                else:
                    fn = lambda: ops.convs_dgrad(p(dzS), p(wd), [0] * G if os.environ.get("BENCH_NOMASK") else p(xS), p(dxS), B, IH, IH, C, 3, 3, 2, F, st)
        elif what == "fwd":
            fn = lambda: ops.conv2d_fwd(p(xs), p(ws), p(bs), p(ys), li == 0, B, IH, IH, C, 3, 3, 2, F, True, st)
        elif what == "wgrad":
            nb = ops.conv2d_wgrad_workspace_bytes(G, B, IH, IH, C, 3, 3, 2, F)
            wk = torch.empty(nb // 4, device=dev)
            dws = [torch.empty_like(w) for w in ws]
            fn = lambda: ops.conv2d_wgrad(p(xs), p(dz), p(dws), p(bs), wk.data_ptr(), nb, li == 0, B, IH, IH, C, 3, 3,
                                          2, F, st)
        else:
            dxs = [torch.empty_like(x) for x in xs]
            fn = lambda: ops.conv2d_dgrad(p(dz), p(ws), p(xs), p(dxs), B, IH, IH, C, 3, 3, 2, F, st)
        ms = timeit(fn)
        per_step = ms * (1 if what == "fwd" else 1)
        tot += per_step
        print(f"conv{li + 1} {what:5s} G={G}: {ms * 1e3:8.1f} us  {flop1 * G / ms / 1e9:7.1f} TFLOP/s")
if MODE == "s" and (not ONLY or "dgw1" in ONLY.split(",")):
    # conv2's data gradient + conv1's weight gradient in one launch (serl_convs_dgrad_wgrad1) at the headline geometry
    G, IH1, C1, F1, F2 = 2, 128, 9, 32, 64
    IH = (IH1 - 3) // 2 + 1; OH = (IH - 3) // 2 + 1
    p = lambda ts: [t.data_ptr() for t in ts]
    fr = [torch.randint(0, 256, (B, IH1, IH1, C1), dtype=torch.uint8, device=dev) for _ in range(G)]
    dzS = []
    for _ in range(G):
        z = torch.randn((B, OH, OH, F2), device=dev)
        buf = torch.empty(z.numel() * 2, dtype=torch.int16, device=dev)
        ops.split_f32(z.data_ptr(), buf.data_ptr(), B * OH * OH, F2, st)
        dzS.append(buf)
    ws2 = [torch.randn((3, 3, F1, F2), device=dev) * 0.05 for _ in range(G)]
    wf = [torch.empty(ops.convs_weight_bytes(False, 3, 3, F1, F2) // 2, dtype=torch.int16, device=dev) for _ in range(G)]
    wd = [torch.empty(9 * F1 * F2 * 2, dtype=torch.int16, device=dev) for _ in range(G)]
    ops.convs_prep_weights(p(ws2), p(wf), p(wd), False, 3, 3, F1, F2, st)
    bm = [torch.randint(0, 256, (B * IH * IH * F1 // 8,), dtype=torch.uint8, device=dev) for _ in range(G)]
    dws = [torch.empty((3, 3, C1, F1), device=dev) for _ in range(G)]
    dbs = [torch.empty(F1, device=dev) for _ in range(G)]
    nb = ops.convs_dgrad_wgrad1_workspace_bytes(G)
    wk = torch.empty(nb // 4, device=dev)
    ms = timeit(lambda: ops.convs_dgrad_wgrad1(p(dzS), p(wd), p(bm), p(fr), p(dws), p(dbs), wk.data_ptr(), nb, B, IH1, IH1, C1, F1, F2, st))
    flops = 2.0 * G * B * (OH * OH * F2 * 9 * F1 + IH * IH * F1 * 9 * C1)
    print(f"conv2 dgrad + conv1 wgrad fused (dgw1) G={G}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s")
print(f"sum of one fwd({FWD_G} groups)+wgrad+dgrad per layer: {tot:.3f} ms per grad-step (ideal {301e9 / 157.3e12 * 1e3:.2f})")
