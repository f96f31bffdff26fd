"""Timeline of ONE critic grad-step inside an update(), from a `rocprofv3 --kernel-trace --output-format csv` of bench.py
(tools/prof.sh): every kernel between two consecutive `td_loss_kernel` starts, with its queue, start offset and duration,
plus per-queue busy time and the intervals in which no queue ran anything.

  python tools/step_timeline.py <dir with *_kernel_trace.csv> [which td_loss occurrence, default the middle one]
"""
import csv, glob, os, sys
from collections import defaultdict

f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"],
                int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
               for r in csv.DictReader(open(f))))
td = [i for i, r in enumerate(rows) if r[3].startswith("td_loss_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(td) // 2
t0, t1 = rows[td[k]][0], rows[td[k + 1]][0]
sel = [r for r in rows if t0 <= r[0] < t1]
print(f"grad-step {k}: {(t1 - t0) / 1e3:.1f} us, {len(sel)} kernels")
busy = defaultdict(float)
for s, e, q, n, gx, gy, gz in sel:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "")[:64]
    print(f"  q{q:<2d} +{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  grid {gx}x{gy}x{gz:<3d} {short}")
    busy[q] += e - s
for q in sorted(busy):
    print(f"queue {q}: busy {busy[q] / 1e3:.1f} us")
ev = sorted([(s, 1) for s, *_ in sel] + [(e, -1) for _, e, *_ in sel])
depth, last, idle, one = 0, t0, 0.0, 0.0
for t, d in ev:
    t = min(t, t1)
    if depth == 0: idle += t - last
    if depth == 1: one += t - last
    depth += d; last = t
print(f"no kernel running: {idle / 1e3:.1f} us; exactly one running: {one / 1e3:.1f} us")
