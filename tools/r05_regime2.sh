set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_regime; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline"
$B --steps 20 --warmup 5 --clock-monitor-early > $O/clock.json 2> $O/clock.err
$B --steps 40 --warmup 0 --clock-monitor-early > $O/clock_w0.json 2> $O/clock_w0.err
$B --steps 20 --warmup 5 --step-stamps > $O/stamps.json 2> $O/stamps.err
tail -3 $O/clock.err; tail -3 $O/clock_w0.err; cat $O/stamps.err
