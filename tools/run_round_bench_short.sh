# The headline subset of tools/run_round_bench.sh (no PMC passes): bench JSON lines, step timeline, kernel trace.
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r04b}
mkdir -p gpurun_out/$TAG
B="python bench.py --steps 200 --warmup 20"
$B > gpurun_out/$TAG/bench_train_anet.json 2> gpurun_out/$TAG/bench_train_anet.err
for w in yc2_100m yc2_2d3d anet_ragged hbm_stress; do $B --workload $w --no-cpu-baseline > gpurun_out/$TAG/bench_$w.json 2> /dev/null; done
$B --eval --no-cpu-baseline > gpurun_out/$TAG/bench_eval_anet.json 2> /dev/null
$B --force-dp --no-cpu-baseline > gpurun_out/$TAG/bench_train_anet_dp1.json 2> /dev/null
$B --no-lookahead --no-cpu-baseline > gpurun_out/$TAG/bench_train_anet_no_lookahead.json 2> /dev/null
$B --no-cpu-baseline --no-roofline --step-stamps --clock-monitor > /dev/null 2> gpurun_out/$TAG/step_timeline.txt
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2> $REPO/gpurun_out/$TAG/kt.err
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py "$DB" $REPO/gpurun_out/$TAG/kernel_stats_bench_train_anet.csv > /dev/null
python $REPO/tools/rocpd_timeline.py "$DB" > $REPO/gpurun_out/$TAG/kernel_timeline.txt
cd $REPO
for f in gpurun_out/$TAG/bench_*.json; do python -c "
import json
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))"; done
