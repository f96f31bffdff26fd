# whole-update A/B of an environment switch inside one GPU call: bash tools/ab_env.sh VAR "0 1 0 1"
V=$1
for m in $2; do
  env $V=$m timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --no-highest --no-extra-configs --steps 8 > gpurun_out/ab_env_$m.json 2> gpurun_out/ab_env_$m.err || { tail -5 gpurun_out/ab_env_$m.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/ab_env_$m.json').read().strip().splitlines()[-1]); print('$V=$m:', d['value'], d['ms_per_step'], d['last_info'])"
done
