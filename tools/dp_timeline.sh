set -u
REPO=$PWD; OUT=$REPO/gpurun_out/r06dp; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for M in single dp; do
  F=""; [ $M = dp ] && F="--force-dp"
  rm -rf /tmp/p_$M && rocprofv3 --kernel-trace -d /tmp/p_$M -o kt -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $F > $OUT/kt_$M.json 2> $OUT/kt_$M.err
  DB=$(find /tmp/p_$M -name "*.db" | head -1)
  python $REPO/tools/rocpd_timeline.py "$DB" > $OUT/timeline_$M.txt 2>> $OUT/kt_$M.err
done
cd $REPO
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b_single_$i.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp > $OUT/b_dp_$i.json 2>/dev/null
done
grep -h -o '"ms_per_step": [0-9.]*' $OUT/b_*.json
