#!/bin/bash
# usage: tools/pmc.sh <tag> "<conv3 fwd,conv2 dgrad>"   -- hardware counters of single conv launches (one rocprofv3 run per pass)
TAG=$1; ONLY=$2
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
CMD="mkdir -p gpurun_out/pmc_$TAG && cd /tmp && export TMPDIR=/tmp && export BENCH_ONLY='$ONLY' BENCH_N=2"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  CMD="$CMD && timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/p$i -- python3 \$GRAFT_REPO_ROOT/tools/bench_conv.py > \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/p$i.out 2>&1"
done
/usr/local/graft/bin/gpurun --timeout 600 -- "$CMD; tail -3 \$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/p3.out" 2>&1 | tail -6
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'gemm_kernel' not in n and 'igemm' not in n: continue
        n=n.replace('igemm_kernel','IG').replace('sgemm_kernel','SG').replace('igemm_bf3_kernel','BF3').replace('TileCfg','T')[:60]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in acc.items():
    print(n)
    v={k:sum(x)/len(x) for k,x in c.items()}
    for k in sorted(v): print(f"   {k:32s} {v[k]:16.0f}")
    try:
        wc=v['SQ_WAVE_CYCLES']; bc=v['SQ_BUSY_CU_CYCLES']
        print(f"   -> waves/CU-cycle occupancy {wc/bc:.2f}; wait_any {v['SQ_WAIT_INST_ANY']/wc:.2f}; wait_lds {v['SQ_WAIT_INST_LDS']/wc:.2f}; valu_active/busy {v['SQ_ACTIVE_INST_VALU']/bc:.2f}; lds_active/busy {v['SQ_ACTIVE_INST_LDS']/bc:.2f}; mfma_busy/busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/bc:.2f}")
        print(f"   -> insts/wave: valu {v['SQ_INSTS_VALU']/v['SQ_WAVES']:.0f} mfma {v['SQ_INSTS_MFMA']/v['SQ_WAVES']:.0f} lds {v['SQ_INSTS_LDS']/v['SQ_WAVES']:.0f}; bank_conflict/lds_active {v['SQ_LDS_BANK_CONFLICT']/max(1,v['SQ_LDS_IDX_ACTIVE']):.2f}; vmem level {v['SQ_INST_LEVEL_VMEM']/max(1,v['SQ_ACTIVE_INST_VMEM']):.1f}")
        print(f"   -> TCC hit rate {v['TCC_HIT_sum']/max(1,v['TCC_HIT_sum']+v['TCC_MISS_sum']):.2f}; TCP->TCC read latency {v['TCP_TCC_READ_REQ_LATENCY_sum']/max(1,v['TCP_TCC_READ_REQ_sum']):.0f} cyc")
    except KeyError as e: print('   missing',e)
PY
