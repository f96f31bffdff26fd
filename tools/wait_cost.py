"""Host cost of one cross-stream dependency on this ROCm: hipEventRecord + hipStreamWaitEvent against
hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory (stream memory operations), each between two streams that
keep running small kernels (torch fills) so that the dependency is real.  ctypes on libamdhip64."""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")
torch.cuda.init()
x = torch.zeros(1 << 16, device="cuda")
s = [torch.cuda.Stream(), torch.cuda.Stream()]
sp = [C.c_void_p(t.cuda_stream) for t in s]
N = 400

ev = [C.c_void_p() for _ in range(N)]
for e in ev:
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0          # hipEventDisableTiming
torch.cuda.synchronize()


def work(i):
    with torch.cuda.stream(s[i]):
        x.add_(1.0)


for name in ("event", "value"):
    if name == "value":
        sig = C.c_void_p()
        r = hip.hipExtMallocWithFlags(C.byref(sig), C.c_size_t(8), C.c_uint(2))     # hipMallocSignalMemory
        print("hipExtMallocWithFlags(signal):", r)
        if r != 0:
            break
        hip.hipMemset(sig, 0, C.c_size_t(8))
    t_a = t_b = 0.0
    for i in range(N):
        work(0)
        t0 = time.perf_counter()
        if name == "event":
            ra = hip.hipEventRecord(ev[i], sp[0])
        else:
            ra = hip.hipStreamWriteValue32(sp[0], sig, C.c_uint(i + 1), C.c_uint(0))
        t1 = time.perf_counter()
        if name == "event":
            rb = hip.hipStreamWaitEvent(sp[1], ev[i], C.c_uint(0))
        else:
            rb = hip.hipStreamWaitValue32(sp[1], sig, C.c_uint(i + 1), C.c_uint(0), C.c_uint(0xFFFFFFFF))   # Gte
        t2 = time.perf_counter()
        assert ra == 0 and rb == 0, (name, ra, rb)
        t_a += t1 - t0; t_b += t2 - t1
        work(1)
    torch.cuda.synchronize()
    print(f"{name:6s}: producer call {t_a / N * 1e6:6.2f} us, consumer wait call {t_b / N * 1e6:6.2f} us (host time per call, {N} dependencies)")
print("x[0] =", float(x[0]), "(expected", 2.0 * N * 2 if False else "n/a", ")")
