set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tests3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_trajectory.py -s -q -k "ragged_packed" > $O/traj_packed.log 2>&1; echo "traj packed rc=$?"
grep -h "step [0-9]:\|parameter deltas\|passed\|failed" $O/traj_packed.log | tail -16
python tools/chain_probe.py > $O/chain_clock.txt 2>&1; grep "fused chain" $O/chain_clock.txt
COOT_REFERENCE_ROOT=$PWD/_refship timeout 900 python -m pytest tests/test_gpu_reference_on_device.py tests/test_reference_binding.py -s -q > $O/reference_on_gpu.log 2>&1; echo "ref rc=$?"; tail -3 $O/reference_on_gpu.log
timeout 600 python -m pytest tests/test_gpu_determinism.py -q > $O/det.log 2>&1; echo "det rc=$?"; tail -2 $O/det.log
