# whole-update A/B of two builds of the library inside one GPU call: bash tools/ab_lib.sh "<V or ''> ..."  (libserl_hip_<V>.so; "base" = the default library)
for v in $1; do
  if [ "$v" = base ]; then unset SERL_AMD_LIB; else export SERL_AMD_LIB=$PWD/serl_amd/csrc/libserl_hip_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --no-highest --no-extra-configs --steps 8 > gpurun_out/ab_lib_$v.json 2> gpurun_out/ab_lib_$v.err || { tail -5 gpurun_out/ab_lib_$v.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/ab_lib_$v.json').read().strip().splitlines()[-1]); print('$v:', d['value'], d['ms_per_step'])"
done
