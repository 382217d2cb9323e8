# Round 5, verdict item 1: what separates the driver's regime (bench.py --steps 20 --warmup 5 in a fresh process) from a long run.
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_regime}
O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
for i in 1 2 3; do $B --steps 20 --warmup 5 > $O/d20_$i.json 2> $O/d20_$i.err; done
$B --steps 20 --warmup 5 --repeat 8 > $O/blocks.json 2> $O/blocks.err
$B --steps 20 --warmup 5 --repeat 3 --per-step-events > $O/events.json 2> $O/events.err
$B --steps 20 --warmup 5 --clock-monitor-early > $O/clock.json 2> $O/clock.err
$B --steps 20 --warmup 50 > $O/w50.json 2> $O/w50.err
$B --steps 200 --warmup 20 > $O/long.json 2> $O/long.err
$B --steps 200 --warmup 5 > $O/long_w5.json 2> $O/long_w5.err
for f in $O/*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step'), d.get('blocks_ms_per_step'))"; done
grep -h "per-step\|^  \|shader clock\|blocks of" $O/*.err
