#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a one-GPU box: `python bench.py --gpus 2` starts its own two ranks
# (torch.distributed.run), both on cuda:0 over gloo (SERL_AMD_BENCH_REHEARSAL=1).  gloo moves the 21 MB gradient through
# the host; the default rehearsal is a short one (REH_ARGS="--batch 32 --utd 2"), REH_ARGS=" " runs the headline workload.
# torch stages a device tensor through pinned memory on pool streams for gloo; two ranks on one card stalled in that staging
# for minutes now and then, so serl_amd.dist makes the host copy itself when the backend is gloo (2-3 s per rehearsal, small
# and full-size workload, 9 of 9 runs).
# REH_DUMP_AFTER=<s>: every rank dumps its Python stacks and exits after <s> seconds; SERL_AMD_DP_TRACE=1: a line around
# every all-reduce.
# PYTHONFAULTHANDLER + `timeout -s ABRT`: a hang ends with the Python stacks of all threads of every rank on stderr.
export SERL_AMD_BENCH_REHEARSAL=1 PYTHONFAULTHANDLER=1
mkdir -p gpurun_out/reh
timeout -s ABRT -k 10 ${REH_TIMEOUT:-150} python -u bench.py --gpus 2 --steps ${REH_STEPS:-1} --warmup 1 \
    ${REH_ARGS:---batch 32 --utd 2} > gpurun_out/reh/out.json 2> gpurun_out/reh/err.log
rc=$?
cut -c 1-600 gpurun_out/reh/out.json; echo; tail -n 60 gpurun_out/reh/err.log
exit $rc
