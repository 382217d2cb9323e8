set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_20_5.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['trace'], d['cpu_baseline']['value'])"
