set -u
cd $GRAFT_REPO_ROOT
REPO=$PWD
O=$REPO/gpurun_out/r05_perf1; mkdir -p $O
NS=$REPO/coot-videotext_amd/lib/libcoot_hip_nostore.so
export TMPDIR=/tmp; cd /tmp
for v in def; do
  L=$REPO/coot-videotext_amd/lib/libcoot_hip.so; [ $v = nostore ] && L=$NS
  rm -rf /tmp/prof_$v && COOT_HIP_LIB=$L rocprofv3 --kernel-trace -d /tmp/prof_$v -o kt -- python $REPO/bench.py --steps 35 --warmup 5 --no-cpu-baseline --no-roofline > $O/kt_$v.json 2> $O/kt_$v.err
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py "$DB" $O/kernel_stats_$v.csv > /dev/null
  python $REPO/tools/rocpd_early_late.py "$DB" $O/early_late_$v.txt > /dev/null
done
cd $REPO
cat $O/early_late_def.txt
