# Build serl_amd/csrc/libserl_hip_<V>.so with extra -D flags on one source file (default convs.hip; SRC=ensemble.hip ...)
# (A/B and ablation builds; the other objects are shared with the default library).
# usage: [SRC=file.hip] bash tools/build_variant.sh V "-DFOO -DBAR=1"
set -e
V=$1; shift
SRC=${SRC:-convs.hip}
BASE=${SRC%.hip}
cd "$(dirname "$0")/../serl_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $@ -c $SRC -o ${BASE}_$V.o
OBJS=$(ls *.o | grep -v "^${BASE}" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libserl_hip_$V.so $OBJS ${BASE}_$V.o
rm -f ${BASE}_$V.o
echo built libserl_hip_$V.so
