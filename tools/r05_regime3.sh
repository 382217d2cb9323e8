set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_regime; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --per-step-events"
ls /sys/class/drm/ > $O/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/ | head -100 >> $O/sysfs_ls.txt 2>&1
$B --sysfs-clocks > $O/p_base.json 2> $O/p_base.err
$B --pre-burn hbm:40 --sysfs-clocks > $O/p_hbm.json 2> $O/p_hbm.err
$B --pre-burn alu:40 > $O/p_alu.json 2> $O/p_alu.err
$B --pre-burn hbm:200 > $O/p_hbm200.json 2> $O/p_hbm200.err
for f in base hbm alu hbm200; do echo == $f; grep -A1 "per-step\|sysfs clocks" $O/p_$f.err | grep -v "^--"; python -c "
import json; d=json.load(open('$O/p_$f.json')); print(d['value'], d['ms_per_step'])"; done
