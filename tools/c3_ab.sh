for v in "" old; do
  if [ -n "$v" ]; then export SERL_AMD_LIB=$PWD/serl_amd/csrc/libserl_hip_$v.so; else unset SERL_AMD_LIB; fi
  echo "== ${v:-base}"
  BENCH_N=50 BENCH_ONLY="conv3 fwd" timeout -k 10 100 python tools/bench_conv.py 2>&1 | grep "conv3 fwd" || exit 1
  BENCH_N=50 BENCH_FWD_G=2 BENCH_ONLY="conv3 fwd" timeout -k 10 100 python tools/bench_conv.py 2>&1 | grep "conv3 fwd" || exit 1
done
