set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_boxes; mkdir -p $O
T=$(date +%s)
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/box_${T}_$i.json 2>/dev/null; done
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > $O/box_${T}_long.json 2>/dev/null
for f in $O/box_${T}_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], d['steps'], d['value'], d['ms_per_step'])"; done
