#!/bin/bash
# usage: tools/pmc_r03.sh <out.txt>  -- hardware counters (SQ / TCP / TCC) of the conv kernels of the headline workload, one
# rocprofv3 --pmc run per counter set and benchmark script (the program itself behind `--`); summary -> <out.txt>
# (MFMA-busy share, occupancy, LDS bank conflicts, L2 hit rate per kernel).  Kernels: the fused layers 1+2
# (pconv_fused12pp), conv2 patch kernels, conv3 wide patch kernels (pconv_fwdw / pconv_dgradw), conv4 and the weight
# gradients on the register-staged im2col kernel (sgemm_kernel<...>).
OUT=${1:-profiles/r03_pmc.txt}
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
D=gpurun_out/pmc_r03
CMD="mkdir -p $D && cd /tmp && export TMPDIR=/tmp && export BENCH_N=2 BENCH_ONLY='conv2 fwd,conv2 dgrad,conv2 wgrad,conv3 fwd,conv3 dgrad,conv3 wgrad,conv4 fwd,conv4 dgrad,conv4 wgrad,conv1 wgrad,dgw1'"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  for S in bench_conv bench_fused12; do
    CMD="$CMD && timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d \$GRAFT_REPO_ROOT/$D/${S}_p$i -- python3 \$GRAFT_REPO_ROOT/tools/$S.py > \$GRAFT_REPO_ROOT/$D/${S}_p$i.out 2>&1"
  done
done
if [ -z "$PMC_PARSE_ONLY" ]; then
rm -rf $D
/usr/local/graft/bin/gpurun --timeout 900 -- "$CMD; echo done" 2>&1 | tail -3
fi
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
NAMES = [("pconv_fused12pp_kernel", "layers 1+2 fused, no-gradient passes (pconv_fused12pp_kernel), 2 groups"),
         ("pconv_fwd_kernel", "conv2 forward patch kernel (pconv_fwd_kernel), 4 groups"),
         ("pconv_dgrad_kernel", "conv2 data gradient patch kernel (pconv_dgrad_kernel), 2 groups"),
         ("pconv_wgrad_kernel", "conv2 weight gradient patch kernel (pconv_wgrad_kernel), 2 groups"),
         ("pconv_dgw1_kernel", "conv2 data gradient + conv1 weight gradient in one kernel (pconv_dgw1_kernel), 2 groups"),
         ("pconv_fwdw_kernel", "conv3 forward wide patch kernel (pconv_fwdw_kernel), 4 groups"),
         ("pconv_dgradw_kernel", "conv3 data gradient wide patch kernel (pconv_dgradw_kernel), 2 groups"),
         ("pconv4_fwd_kernel", "conv4 forward, parity-class patch kernel (pconv4_fwd_kernel), 4 groups"),
         ("pconv_wgradw_kernel", "conv3 weight gradient wide patch kernel (pconv_wgradw_kernel), 2 groups"),
         ("sgemm_kernel<ConvFwdS<false>", "conv4 forward, register-staged im2col (sgemm_kernel<ConvFwdS>), 4 groups"),
         ("sgemm_kernel<ConvDgradS", "conv4 data gradient (sgemm_kernel<ConvDgradS>), 2 groups"),
         ("sgemm_kernel<ConvWgradS<false>", "conv4 weight gradient (sgemm_kernel<ConvWgradS<false>>; before round 3's conv3 patch kernel: conv3 + conv4 averaged), 2 groups"),
         ("sgemm_kernel<ConvWgradS<true>", "conv1 weight gradient, uint8 input (sgemm_kernel<ConvWgradS<true>>), 2 groups"),
         ("wgrad_reduce_kernel", "split-K slab reduction (wgrad_reduce_kernel), averaged over the layers")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_r03/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for key, _ in NAMES:
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
with open(sys.argv[1], "w") as out:
    def w(s=""):
        print(s); out.write(s + "\n")
    w("# rocprofv3 --pmc counters of the conv kernels at the headline shapes (B = 256, E1 encoder), stand-alone launches")
    w("# (tools/pmc_r03.sh: tools/bench_conv.py and tools/bench_fused12.py under three counter sets, one run per set;")
    w("#  values are averages per dispatch; derived lines: see the formulas)")
    for key, title in NAMES:
        c = acc.get(key)
        if not c:
            continue
        v = {k: sum(x) / len(x) for k, x in c.items()}
        w(f"\n== {title}")
        for k in sorted(v):
            w(f"   {k:32s} {v[k]:16.0f}")
        try:
            wc, bc = v["SQ_WAVE_CYCLES"], v["SQ_BUSY_CU_CYCLES"]
            w(f"   -> MFMA pipe busy share (MFMA busy cycles / (4 SIMDs x CU busy cycles)) {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * bc):.3f};  occupancy (wave cycles / CU busy cycles) {wc / bc:.2f} waves;  "
              f"VALU active / busy {v['SQ_ACTIVE_INST_VALU'] / bc:.2f};  LDS active / busy {v['SQ_ACTIVE_INST_LDS'] / bc:.2f}")
            w(f"   -> wait-any share of wave cycles {v['SQ_WAIT_INST_ANY'] / wc:.2f};  wait-LDS share {v['SQ_WAIT_INST_LDS'] / wc:.2f}")
            w(f"   -> per wave: {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f} VALU, {v['SQ_INSTS_MFMA'] / v['SQ_WAVES']:.0f} MFMA, {v['SQ_INSTS_LDS'] / v['SQ_WAVES']:.0f} LDS instructions;  "
              f"LDS bank-conflict cycles / LDS active cycles {v['SQ_LDS_BANK_CONFLICT'] / max(1, v['SQ_LDS_IDX_ACTIVE']):.3f}")
            w(f"   -> L2 (TCC) hit rate {v['TCC_HIT_sum'] / max(1, v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f};  "
              f"L1 -> L2 read latency {v['TCP_TCC_READ_REQ_LATENCY_sum'] / max(1, v['TCP_TCC_READ_REQ_sum']):.0f} cycles")
        except KeyError as e:
            w(f"   (missing counter {e})")
PY
