set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
B="python bench.py --steps 200 --warmup 20"
$B > gpurun_out/r04/bench_train_anet.json 2> gpurun_out/r04/bench_train_anet.err
$B --workload yc2_100m > gpurun_out/r04/bench_yc2_100m.json 2> gpurun_out/r04/bench_yc2_100m.err
for w in yc2_2d3d yc2_2d3d_2816 hbm_stress anet_ragged; do $B --workload $w --no-cpu-baseline > gpurun_out/r04/bench_$w.json 2> gpurun_out/r04/bench_$w.err; done
$B --workload anet_ragged --padded --no-cpu-baseline > gpurun_out/r04/bench_anet_ragged_padded.json 2> /dev/null
$B --eval --no-cpu-baseline > gpurun_out/r04/bench_eval_anet.json 2> /dev/null
$B --force-dp --no-cpu-baseline > gpurun_out/r04/bench_train_anet_dp1.json 2> gpurun_out/r04/bench_dp1.err
$B --no-lookahead --no-cpu-baseline > gpurun_out/r04/bench_train_anet_no_lookahead.json 2> /dev/null
python tools/dp_loss_probe.py > gpurun_out/r04/dp_loss_probe.txt 2>&1
$B --no-cpu-baseline --no-roofline --step-stamps --clock-monitor > /dev/null 2> gpurun_out/r04/step_timeline.txt
bash tools/profile_round.sh r04 > gpurun_out/r04/profile.log 2>&1
for f in gpurun_out/r04/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value_1thread'))"; done
