#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a one-GPU box: two ranks on cuda:0 over gloo (SERL_AMD_BENCH_REHEARSAL=1).
# gloo moves the 21 MB gradient through the host with both ranks time-slicing one GPU (~1 s per all-reduce at the
# headline sizes), so the default rehearsal is a short one: REH_ARGS="--batch 32 --utd 2".
# Each rank runs under `timeout -s ABRT` with PYTHONFAULTHANDLER=1, so a hang ends with the Python stacks of all threads.
export SERL_AMD_BENCH_REHEARSAL=1 PYTHONFAULTHANDLER=1 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655
mkdir -p gpurun_out/reh
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 ${REH_TIMEOUT:-120} python -u bench.py --gpus 2 --steps ${REH_STEPS:-1} --warmup 1 ${REH_ARGS:---batch 32 --utd 2} \
      > gpurun_out/reh/rank$r.out 2> gpurun_out/reh/rank$r.err &
done
wait
tail -c 400 gpurun_out/reh/rank0.out; echo; tail -n 40 gpurun_out/reh/rank0.err
