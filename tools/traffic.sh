#!/bin/bash
# usage: [TRAFFIC_OPS="conv4_fwd@4 ..."] tools/traffic.sh <out.json>  -- HBM-side traffic of every conv launch of the bench workload:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (MI355X_MICROARCH.md, HBM section), one
# launch configuration per run (tools/bench_conv.py, BENCH_ONLY), averaged over the dispatches of the run.
OUT=$1
# <layer>_<op>@<groups>: forward launches carry 4 groups (target + critic passes) or 2 (actor run-ahead pass)
# (since round 2 the driver launches: conv1 fwd with 2 groups (the differentiated pass), layers 1+2 of the other passes fused
#  (fwd12, 2 groups per launch), conv2 fwd with 2 groups, conv3/conv4 fwd with 4 and 2 groups)
OPS=${TRAFFIC_OPS:-"conv1_fwd@2 conv1_fwd12@2 conv1_fwd12@4 conv1_fwd12keep@4 conv1_fwd12keep@2 conv1_wgrad@2 conv2_fwd@4 conv2_fwd@2 conv2_wgrad@2 conv2_dgrad@2 conv2_dgw1@2 conv3_fwd@4 conv3_fwd@2 conv3_wgrad@2 conv3_dgrad@2 conv4_fwd@4 conv4_fwd@2 conv4_wgrad@2 conv4_dgrad@2"}
CMD="mkdir -p gpurun_out/traffic && cd /tmp && export TMPDIR=/tmp && export BENCH_N=2"
for OPG in $OPS; do
  OP=${OPG%@*}; G=${OPG#*@}
  SCRIPT=bench_conv.py
  KEEP=
  if [ "$OP" = conv1_fwd12 ]; then SCRIPT=bench_fused12.py; fi
  if [ "$OP" = conv1_fwd12keep ]; then SCRIPT=bench_fused12.py; KEEP="BENCH_KEEP=1"; fi
  ONLY="${OP/_/ }"
  if [ "$OP" = conv2_dgw1 ]; then ONLY=dgw1; fi      # conv2's data gradient + conv1's weight gradient in one launch
  for P in FETCH_SIZE WRITE_SIZE; do
    CMD="$CMD && $KEEP BENCH_G=$G BENCH_FWD_G=$G BENCH_ONLY='$ONLY' timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/traffic/${OP}_g${G}_$P -- python3 \$GRAFT_REPO_ROOT/tools/$SCRIPT > \$GRAFT_REPO_ROOT/gpurun_out/traffic/${OP}_g${G}_$P.out 2>&1"
  done
done
rm -rf gpurun_out/traffic
/usr/local/graft/bin/gpurun --timeout 1100 -- "$CMD; echo done" 2>&1 | tail -3
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = {}
for d in sorted(glob.glob("gpurun_out/traffic/*_FETCH_SIZE")) + sorted(glob.glob("gpurun_out/traffic/*_WRITE_SIZE")):
    op, counter = d.split("/")[-1].rsplit("_", 2)[0], "_".join(d.split("/")[-1].rsplit("_", 2)[1:])   # conv1_fwd_g4
    vals, name = [], None
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            hit = "pconv_fused12" in kn if "fwd12" in op else ("gemm_kernel" in kn or "pconv_" in kn or "pconv4_" in kn)
            if hit and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"])); name = r["Kernel_Name"]
    if vals:
        e = out.setdefault(op.replace("_", " "), {"kernel": name})
        e[counter + "_KB"] = sum(vals) / len(vals)
for op, e in out.items():
    # gfx950: FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) loads at 64 B -> double it; WRITE_SIZE is exact
    e["hbm_side_bytes"] = 2 * 1024 * e.get("FETCH_SIZE_KB", 0) + 1024 * e.get("WRITE_SIZE_KB", 0)
json.dump({"note": "L2-miss (fabric-side) bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md gfx950 correction); "
                   "Infinity-Cache hits are included; B=256, key = '<layer> <op> g<groups per launch>' as launched by bench.py", "launches": out},
          open(sys.argv[1], "w"), indent=1)
for op, e in out.items(): print(op, round(e["hbm_side_bytes"] / 1e6, 1), "MB")
PY
