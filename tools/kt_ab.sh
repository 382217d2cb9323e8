#!/bin/bash
# kernel-trace stats of bench.py (20 / 5) per arm:  bash tools/kt_ab.sh <tag> "<name>|<env>|<bench args>" ...
set -u
cd $GRAFT_REPO_ROOT; REPO=$PWD; TAG=$1; shift; O=$REPO/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for arm in "$@"; do
  NAME=$(echo "$arm" | cut -d'|' -f1); ENVS=$(echo "$arm" | cut -d'|' -f2); ARGS=$(echo "$arm" | cut -d'|' -f3)
  rm -rf /tmp/kt_$NAME && env $ENVS rocprofv3 --kernel-trace -d /tmp/kt_$NAME -o kt -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $ARGS > /dev/null 2> $O/$NAME.err
  DB=$(find /tmp/kt_$NAME -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py "$DB" $O/kernel_stats_$NAME.csv > /dev/null 2>> $O/$NAME.err
  python $REPO/tools/rocpd_timeline.py "$DB" > $O/kernel_timeline_$NAME.txt 2>> $O/$NAME.err
  echo "== $NAME"; cut -d, -f1-5 $O/kernel_stats_$NAME.csv | sed 's/_ZN4coot12_GLOBAL__N_1//; s/_ZN4coot//' | head -24
done
