#!/bin/bash
# Round profile on the GPU box: kernel-trace stats, the HBM traffic counters (separate passes, MI355X_MICROARCH.md) and the MFMA
# counters of `python bench.py`, reduced to the summaries that are committed under profiles/.
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/
set -u
TAG=${1:-r02}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
cd /tmp
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- $BENCH > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats_bench_train_anet.csv" > /dev/null 2>> "$OUT/kt.err"
python $REPO/tools/rocpd_timeline.py "$DB" > "$OUT/kernel_timeline.txt" 2>> "$OUT/kt.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C && rocprofv3 --pmc $C --kernel-trace -d /tmp/prof_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> "$OUT/pmc_$C.err"
  DB=$(find /tmp/prof_$C -name "*.db" | head -1)
  python $REPO/tools/rocpd_pmc.py "$DB" "$OUT/pmc_$(echo $C | tr A-Z a-z).csv" > /dev/null 2>> "$OUT/pmc_$C.err"
done
python $REPO/tools/pmc_traffic.py "$OUT/pmc_fetch_size.csv" "$OUT/pmc_write_size.csv" "$OUT/traffic.json" > /dev/null 2> "$OUT/traffic.err"
# MFMA utilisation: matrix-pipe busy cycles and MFMA op counts against the busy cycles of the shader engines (own pass)
rocprofv3 -L 2>/dev/null | grep -iE "mfma|SQ_BUSY_CY|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU\b" | head -40 > "$OUT/counters_available.txt"
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 SQ_WAVE_CYCLES"; do
  N=$(echo $SET | cut -d" " -f1)
  rm -rf /tmp/prof_$N && rocprofv3 --pmc $SET --kernel-trace -d /tmp/prof_$N -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> "$OUT/pmc_$N.err"
  DB=$(find /tmp/prof_$N -name "*.db" | head -1)
  [ -n "$DB" ] && python $REPO/tools/rocpd_pmc.py "$DB" "$OUT/pmc_$(echo $N | tr A-Z a-z).csv" > /dev/null 2>> "$OUT/pmc_$N.err"
done
ls -la "$OUT"
