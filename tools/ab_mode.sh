# whole-update A/B of serl_convs_set_patch_kernels modes inside one GPU call: interleaved bench.py values (headline only)
# usage: bash tools/ab_mode.sh "8 1 8 1"
for m in $1; do
  SERL_AMD_CONV_PATCH=$m timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --no-highest --no-extra-configs --steps 8 > gpurun_out/ab_mode_$m.json 2> gpurun_out/ab_mode_$m.err || { tail -5 gpurun_out/ab_mode_$m.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/ab_mode_$m.json').read().strip().splitlines()[-1]); print('mode $m:', d['value'], d['ms_per_step'])"
done
