# Round-5 profile: the headline bench lines (driver regime AND long run), kernel trace, PMC passes -> gpurun_out/r05/
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_train_anet_20_5.json 2> $O/bench_train_anet_20_5.err
for i in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_train_anet_20_5_run$i.json 2> /dev/null; done
bash tools/run_round_bench_short.sh r05 > $O/short.log 2>&1
bash tools/profile_round.sh r05 > $O/profile.log 2>&1
tail -12 $O/short.log
python -c "
import json
for f in ['bench_train_anet_20_5','bench_train_anet_20_5_run2','bench_train_anet_20_5_run3']:
    d=json.load(open('$O/'+f+'.json')); r=d.get('roofline') or {}; print(f, d['value'], d['ms_per_step'], r.get('frac'), (r.get('trace') or {}).get('frac'))
"
ls $O | head -50
