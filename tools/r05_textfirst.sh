set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_textfirst; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_train_trajectory.py tests/test_gpu_path.py tests/test_gpu_determinism.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
B="python bench.py --gpus 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  $B --steps 20 --warmup 5 --opt text_first=1 > $O/tf1_$i.json 2>/dev/null
  $B --steps 20 --warmup 5 --opt text_first=0 > $O/tf0_$i.json 2>/dev/null
done
$B --steps 200 --warmup 20 --opt text_first=1 > $O/tf1_long.json 2>/dev/null
$B --steps 200 --warmup 20 --opt text_first=0 > $O/tf0_long.json 2>/dev/null
$B --steps 20 --warmup 5 --opt text_first=1 --step-stamps > /dev/null 2> $O/stamps_tf1.txt
$B --steps 20 --warmup 5 --opt text_first=0 --step-stamps > /dev/null 2> $O/stamps_tf0.txt
for f in $O/tf*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], d['steps'], d['value'], d['ms_per_step'])"; done
for v in tf1 tf0; do echo "== $v first step"; grep -A22 "FIRST step" $O/stamps_$v.txt | grep "weights packed\|local forward done\|global forward done\|joined\|backward starts\|step done"; done
