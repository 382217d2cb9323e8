set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_perf3; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20"
for i in 1 2; do
  $B > $O/base_$i.json 2> /dev/null
  $B --opt half_tiles_0=32768 > $O/h0_$i.json 2> /dev/null
  $B --opt half_tiles_1=32768 > $O/h1_$i.json 2> /dev/null
  $B --opt half_tiles_0=32768 --opt half_tiles_1=32768 > $O/h01_$i.json 2> /dev/null
done
$B --opt half_tiles_0=32768 --opt half_tiles_1=32768 --step-stamps > /dev/null 2> $O/stamps_h01.txt
for f in $O/*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
grep -A25 "^step timeline (HIP" $O/stamps_h01.txt | head -30
