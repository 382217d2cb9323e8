set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_perf1; mkdir -p $O
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_train_trajectory.py -s -q > $O/traj.log 2>&1; echo "traj rc=$?"
timeout 900 python -m pytest tests/test_gpu_bench_dp8.py -s -q > $O/dp8.log 2>&1; echo "dp8 rc=$?"
grep -h "parameter deltas\|delta cosine\|passed\|failed\|8 ranks" $O/traj.log $O/dp8.log | tail -40
NS=$REPO/coot-videotext_amd/lib/libcoot_hip_nostore.so
B="python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20"
for i in 1 2; do
  $B > $O/b_def_$i.json 2> /dev/null
  COOT_HIP_LIB=$NS $B > $O/b_nostore_$i.json 2> /dev/null
done
python tools/chain_probe.py > $O/chain_def.txt 2>&1
COOT_HIP_LIB=$NS python tools/chain_probe.py > $O/chain_nostore.txt 2>&1
export TMPDIR=/tmp; cd /tmp
for v in def nostore; do
  L=$REPO/coot-videotext_amd/lib/libcoot_hip.so; [ $v = nostore ] && L=$NS
  rm -rf /tmp/prof_$v && COOT_HIP_LIB=$L rocprofv3 --kernel-trace -d /tmp/prof_$v -o kt -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2> $O/kt_$v.err
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py "$DB" $O/kernel_stats_$v.csv > /dev/null
  python $REPO/tools/rocpd_early_late.py "$DB" $O/early_late_$v.txt > /dev/null
done
cd $REPO
for f in $O/b_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
cat $O/chain_def.txt $O/chain_nostore.txt | grep "fused chain"
cat $O/early_late_def.txt | head -40
