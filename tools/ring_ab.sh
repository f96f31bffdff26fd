# A/B of the LDS-DMA ring GEMM (csrc/sgemm_ring.h) against the register-staged im2col kernel, and its ablation builds
# (tools/build_variant.sh): usage on the GPU box: bash tools/ring_ab.sh "A B"
set -e
L=serl_amd/csrc
for v in "" ${1:-}; do
  s=${v:+_$v}
  for m in ${MODES:-2 1 3}; do
    echo "== lib$s BENCH_PATCH=$m (2: im2col kernels for layers 3-4, 1: wide patch kernel for conv3, 3: + ring GEMM for conv4)"
    SERL_AMD_LIB=$L/libserl_hip$s.so BENCH_PATCH=$m BENCH_FWD_G=${G:-4} BENCH_ONLY="conv3 fwd,conv4 fwd" timeout -k 10 100 python tools/bench_conv.py 2>/dev/null | grep conv
  done
done
