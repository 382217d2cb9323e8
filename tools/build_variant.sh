#!/bin/bash
# A second build of libcoot_hip.so with extra compile flags, next to the default one (A/B of compile-time kernel variants in one GPU call):
#   bash tools/build_variant.sh <tag> "<extra hipcc flags>"   ->  coot-videotext_amd/lib/libcoot_hip_<tag>.so
#   COOT_HIP_LIB=$PWD/coot-videotext_amd/lib/libcoot_hip_<tag>.so python bench.py ...
set -e
TAG=$1; EXTRA=$2
cd "$(dirname "$0")/../coot-videotext_amd/csrc"
mkdir -p obj_$TAG ../lib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result $EXTRA"
pids=()
for f in gemm rowops attention pool loss loss_fused fused ref_f32 det retrieval host_input api api_loss api_step; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o obj_$TAG/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map obj_$TAG/*.o -o ../lib/libcoot_hip_$TAG.so
rm -rf obj_$TAG
echo "built ../lib/libcoot_hip_$TAG.so"
