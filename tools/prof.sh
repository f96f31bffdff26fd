#!/bin/bash
# usage: tools_prof.sh <tag>   (runs on the GPU box via gpurun; prints the per-kernel summary)
TAG=$1
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/prof_$TAG && cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 \$GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-highest --no-kernel-timer --no-extra-configs > \$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/bench.out 2>&1; tail -c 600 \$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/bench.out" 2>&1 | tail -4
python3 - <<PY
import csv,glob
import os; f=max(glob.glob("gpurun_out/prof_$TAG/*/*_kernel_stats.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6, "per grad-step ms", tot/1e6/80, "(3 updates + the end-to-end loop of 2 = 4 x 20 grad-steps... see bench.py)")
for r in rows[:22]:
    n=r['Name'].replace('igemm_kernel','IG').replace('TileCfg','T')[:95]
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls={r['Calls']:>5} avg={float(r['AverageNs'])/1e3:9.1f}us  {n}")
PY
