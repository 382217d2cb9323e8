#!/bin/bash
# Alternating A/B/C... of bench.py at the driver's regime on ONE GPU box (box-to-box spread is ~2 %, so arms only compare within a call).
#   bash tools/ab_multi.sh <tag> <rounds> "<name>|<env assignments>|<bench args>" ...
# Every round runs each arm once (--steps 20 --warmup 5 unless the arm's bench args say otherwise); prints ms_per_step per arm and round.
set -u
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for r in $(seq 1 $ROUNDS); do
  for arm in "$@"; do
    NAME=$(echo "$arm" | cut -d'|' -f1); ENVS=$(echo "$arm" | cut -d'|' -f2); ARGS=$(echo "$arm" | cut -d'|' -f3)
    case "$ARGS" in *--steps*) S="";; *) S="--steps 20 --warmup 5";; esac
    env $ENVS python bench.py --gpus 1 $S --no-cpu-baseline --no-roofline $ARGS >> "$OUT/$NAME.jsonl" 2>> "$OUT/$NAME.err"
  done
done
for arm in "$@"; do
  NAME=$(echo "$arm" | cut -d'|' -f1)
  python - "$OUT/$NAME.jsonl" "$NAME" <<'PY'
import json, sys
ms = [json.loads(l)["ms_per_step"] for l in open(sys.argv[1]) if l.startswith("{")]
print(f"{sys.argv[2]:28s} " + " ".join(f"{m:.3f}" for m in ms) + (f"   median {sorted(ms)[len(ms)//2]:.3f}" if ms else "  (no result)"))
PY
done | tee "$OUT/summary.txt"
