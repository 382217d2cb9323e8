#!/bin/bash
# A/B of one coot_set_option switch on the GPU box: bench line, kernel-trace stats and FETCH_SIZE of the named kernel per setting.
#   bash tools/ab_option.sh <tag> <kernel substring> "<COOT_OPTIONS a>" "<COOT_OPTIONS b>" [bench args]
set -u
TAG=$1; KERN=$2; A=$3; B=$4; shift 4
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for OPT in "$A" "$B"; do
  i=$((i+1))
  export COOT_OPTIONS="$OPT"
  for rep in 1 2; do python $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline "$@" >> "$OUT/bench_$i.json" 2>> "$OUT/bench_$i.err"; done
  rm -rf /tmp/prof_kt$i && rocprofv3 --kernel-trace -d /tmp/prof_kt$i -o kt -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > /dev/null 2> "$OUT/kt_$i.err"
  DB=$(find /tmp/prof_kt$i -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats_$i.csv" > /dev/null 2>> "$OUT/kt_$i.err"
  python $REPO/tools/rocpd_timeline.py "$DB" > "$OUT/kernel_timeline_$i.txt" 2>> "$OUT/kt_$i.err"
  rm -rf /tmp/prof_f$i && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f$i -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > /dev/null 2> "$OUT/pmc_$i.err"
  DB=$(find /tmp/prof_f$i -name "*.db" | head -1)
  python $REPO/tools/rocpd_pmc.py "$DB" "$OUT/pmc_fetch_$i.csv" > /dev/null 2>> "$OUT/pmc_$i.err"
  echo "== $OPT"; cat "$OUT/bench_$i.json" | python -c "import sys,json; [print(' ', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  grep "$KERN" "$OUT/kernel_stats_$i.csv" | cut -c1-200
  grep "$KERN" "$OUT/pmc_fetch_$i.csv" | cut -c1-200
done
