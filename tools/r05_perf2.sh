set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_perf2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
B="python bench.py --no-cpu-baseline --steps 100 --warmup 20"
for w in 0 512 1024 4096; do
  $B --workload hbm_stress --opt ln_stream_wgs=$w > $O/hbm_$w.json 2> /dev/null
done
for i in 1 2; do
  COOT_DP_EARLY=1 $B --no-roofline --force-dp > $O/dp_early_$i.json 2> $O/dp_early_$i.err
  COOT_DP_EARLY=0 $B --no-roofline --force-dp > $O/dp_noearly_$i.json 2> /dev/null
  $B --no-roofline > $O/single_$i.json 2> /dev/null
done
for f in $O/*.json; do python -c "
import json; d=json.load(open('$f')); r=d.get('roofline') or {}; print('$f'.split('/')[-1], d['value'], d['ms_per_step'], r.get('frac'), (r.get('plain_launch') or {}).get('frac'), r.get('avg_launch_us'))"; done
