set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_regime; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --per-step-events"
$B --steps 40 --warmup 5 > $O/q_base.json 2> $O/q_base.err
$B --steps 40 --warmup 5 --lr0 > $O/q_lr0.json 2> $O/q_lr0.err
$B --steps 20 --warmup 40 --idle-ms 1000 > $O/q_idle.json 2> $O/q_idle.err
$B --steps 40 --warmup 5 --no-lookahead > $O/q_nola.json 2> $O/q_nola.err
for f in base lr0 idle nola; do echo == $f; grep -A1 "per-step" $O/q_$f.err | grep -v "^--"; python -c "
import json; d=json.load(open('$O/q_$f.json')); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"; done
