#!/bin/bash
# Builds libcoot_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" obj
# -fvisibility=hidden: only what include/coot_hip.h declares (visibility push(default)) is exported
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result $EXTRA_FLAGS"
pids=()
for f in gemm rowops attention pool loss loss_fused loss_f32 fused ref_f32 det retrieval host_input api api_loss api_step; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer obj/$f.o)" ] || [ ../../include/coot_hip.h -nt obj/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map obj/*.o -o "$OUT/libcoot_hip.so"
echo "built $OUT/libcoot_hip.so"
