set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tests1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_trajectory.py -s -q > $O/traj.log 2>&1; echo "traj rc=$?"
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_determinism.py -s -q > $O/parity.log 2>&1; echo "parity rc=$?"
grep -h "parameter deltas\|passed\|failed\|Error" $O/traj.log | tail -30
grep -h "parameter gradients checked\|passed\|failed" $O/parity.log | tail -40
