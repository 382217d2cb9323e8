set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final2; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b$i.json')); print(d['value'], d['ms_per_step'])"; done
