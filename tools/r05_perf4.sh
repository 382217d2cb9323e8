set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_perf4; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20"
for i in 1 2; do
  $B > $O/base_$i.json 2> /dev/null
  $B --opt half_tiles_2=8191 > $O/pre_full_$i.json 2> /dev/null
  $B --opt half_tiles_3=8191 > $O/qkvb_full_$i.json 2> /dev/null
  $B --opt half_tiles_2=8191 --opt half_tiles_3=8191 > $O/bwd_full_$i.json 2> /dev/null
  $B --opt half_tiles_1=32768 > $O/post_half_$i.json 2> /dev/null
done
for f in $O/*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
