# Build serl_amd/csrc/libserl_hip_<V>.so with extra -D flags on convs.hip (A/B and ablation builds; the other objects are
# shared with the default library).  usage: bash tools/build_variant.sh V "-DFOO -DBAR=1"
set -e
V=$1; shift
cd "$(dirname "$0")/../serl_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $@ -c convs.hip -o convs_$V.o
OBJS=$(ls *.o | grep -v '^convs' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libserl_hip_$V.so $OBJS convs_$V.o
rm -f convs_$V.o
echo built libserl_hip_$V.so
