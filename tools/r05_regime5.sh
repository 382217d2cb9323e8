set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_regime; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --per-step-events"
for k in none:0 mfma:60 hbm:60 alu:60; do
  n=$(echo $k | cut -d: -f1)
  $B --pre-burn $k > $O/r_$n.json 2> $O/r_$n.err
  echo == $k; grep -A1 "per-step device" $O/r_$n.err | tail -1; python -c "
import json; d=json.load(open('$O/r_$n.json')); print(d['value'], d['ms_per_step'])"
done
