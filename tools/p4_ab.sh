# stand-alone times of one conv forward launch (default: the last layer's, pconv4.h; BENCH_OP="conv3 fwd" for the wide patch
# kernel) with 4 and 2 groups, for the default library and variant builds (tools/build_variant.sh):  bash tools/p4_ab.sh "<V1> <V2>"
OP=${BENCH_OP:-conv4 fwd}
for v in "" $1; do
  if [ -n "$v" ]; then export SERL_AMD_LIB=$PWD/serl_amd/csrc/libserl_hip_$v.so; else unset SERL_AMD_LIB; fi
  echo "== ${v:-base}"
  BENCH_N=50 BENCH_ONLY="$OP" timeout -k 10 100 python tools/bench_conv.py 2>&1 | grep "$OP" || exit 1
  BENCH_N=50 BENCH_FWD_G=2 BENCH_ONLY="$OP" timeout -k 10 100 python tools/bench_conv.py 2>&1 | grep "$OP" || exit 1
done
