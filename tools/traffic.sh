#!/bin/bash
# usage: tools/traffic.sh <tag> ["conv3 fwd,..."]  -- HBM-side traffic of the conv launches (FETCH_SIZE / WRITE_SIZE, one pass each)
TAG=$1; ONLY=$2
CMD="mkdir -p gpurun_out/traffic_$TAG && cd /tmp && export TMPDIR=/tmp && export BENCH_ONLY='$ONLY' BENCH_N=2"
for P in FETCH_SIZE WRITE_SIZE; do
  CMD="$CMD && timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG/$P -- python3 \$GRAFT_REPO_ROOT/tools/bench_conv.py > \$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG/$P.out 2>&1"
done
/usr/local/graft/bin/gpurun --timeout 600 -- "$CMD; tail -2 \$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG/WRITE_SIZE.out" 2>&1 | tail -4
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
order=[]
for f in glob.glob("gpurun_out/traffic_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'gemm_kernel' not in n: continue
        key=(n.replace('sgemm_kernel','SG').replace('TileCfg','T')[:58], r['Grid_Size'] if 'Grid_Size' in r else '')
        if key not in order: order.append(key)
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in order:
    v={k:sum(x)/len(x) for k,x in acc[key].items()}
    print(key[0], 'grid', key[1], ' FETCH_SIZE', v.get('FETCH_SIZE'), ' WRITE_SIZE', v.get('WRITE_SIZE'))
PY
