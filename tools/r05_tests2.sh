set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tests2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f32_mode.py -s -q > $O/f32.log 2>&1; echo "f32 rc=$?"
timeout 600 python -m pytest tests/test_gpu_train_trajectory.py -q > $O/traj.log 2>&1; echo "traj rc=$?"
grep -h "relative error\|f32 mode\|bf16 path\|passed\|failed\|Error" $O/f32.log | tail -40
tail -3 $O/traj.log
