#!/bin/bash
# A second libcoot_hip.so that differs from the default build in fused.hip only (measurement macros of the chain kernels):
#   bash tools/build_fused_variant.sh <tag> "<extra hipcc flags>"  ->  coot-videotext_amd/lib/libcoot_hip_<tag>.so   (run build.sh first)
set -e
TAG=$1; EXTRA=$2
cd "$(dirname "$0")/../coot-videotext_amd/csrc"
mkdir -p obj_$TAG
cp obj/*.o obj_$TAG/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result $EXTRA -c fused.hip -o obj_$TAG/fused.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map obj_$TAG/*.o -o ../lib/libcoot_hip_$TAG.so
rm -rf obj_$TAG
echo "built ../lib/libcoot_hip_$TAG.so"
