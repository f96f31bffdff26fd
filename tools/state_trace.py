"""Per-kernel durations and the gaps between consecutive kernels of the state-based configs, from a
`rocprofv3 --kernel-trace` of tools/bench_state.py: python tools/state_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict

f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 2:]            # the second half of the run: the C2 loop
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for i, (s, e, n) in enumerate(rows):
    n = n[:70]
    dur[n] += e - s
    cnt[n] += 1
    if i:
        gap[n] += max(0, s - rows[i - 1][1])
tot = rows[-1][1] - rows[0][0]
busy = sum(dur.values())
print(f"{len(rows)} kernels over {tot / 1e3:.0f} us: busy {busy / tot:.2f}, mean duration {busy / len(rows) / 1e3:.2f} us, "
      f"mean gap {(tot - busy) / len(rows) / 1e3:.2f} us")
for n in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
    print(f"{cnt[n]:6d} x  dur {dur[n] / cnt[n] / 1e3:7.2f} us  gap before {gap[n] / cnt[n] / 1e3:7.2f} us  {n}")
