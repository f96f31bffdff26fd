# A/B of conv tile variants: libraries libserl_hip_<V>.so built with -DS_FWD32= / -DS_FWD64= / -DS_DG32= / -DS_DG64= /
# -DS_BIG= overrides (serl_amd/csrc/convs.hip), compared inside one gpurun call (box-to-box spread is +-1 %).
# usage (on the GPU box): bash tools/ab_tiles.sh "D E F"
set -e
L=serl_amd/csrc
VARS=${1:-""}
for v in "" $VARS; do
  s=${v:+_$v}
  echo "== lib$s"
  SERL_AMD_LIB=$L/libserl_hip$s.so timeout -k 10 100 python tools/bench_conv.py 2>/dev/null | grep -i "conv"
  if [ -n "$v" ]; then
    SERL_AMD_LIB=$L/libserl_hip$s.so timeout -k 10 200 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "conv" --timeout 120 --timeout-method thread 2>&1 | tail -1
  fi
done
for i in 1 2; do for v in "" $VARS; do
  s=${v:+_$v}
  SERL_AMD_LIB=$L/libserl_hip$s.so timeout -k 10 120 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$s', d['value'])"
done; done
