#!/bin/bash
# Builds libcoot_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" obj
# -fvisibility=hidden: only what include/coot_hip.h declares (visibility push(default)) is exported
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result $EXTRA_FLAGS"
# two builds of the same sources: bfloat16 operands (libcoot_hip.so, what bench.py times) and IEEE half operands
# (libcoot_hip_f16.so, -DCOOT_OPERAND_F16: common.h; forward-only)
SRCS="gemm rowops attention pool loss loss_fused loss_f32 fused ref_f32 det retrieval host_input api api_loss api_step"
build_one() {  # <obj dir> <output .so> <extra flags>
  local OBJ=$1 SO=$2 EXTRA=$3 pids=()
  mkdir -p $OBJ
  for f in $SRCS; do
    if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer $OBJ/$f.o)" ] || [ ../../include/coot_hip.h -nt $OBJ/$f.o ]; then
      /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f.hip -o $OBJ/$f.o &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait $p; done
  local objs=""
  for f in $SRCS; do objs="$objs $OBJ/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $objs -o "$SO"
  echo "built $SO"
}
build_one obj "$OUT/libcoot_hip.so" ""
build_one obj_f16 "$OUT/libcoot_hip_f16.so" "-DCOOT_OPERAND_F16"
