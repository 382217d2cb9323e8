set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_xcd; mkdir -p $O
for v in "" _xcd5 _xcd3; do
  L=$PWD/coot-videotext_amd/lib/libcoot_hip$v.so
  echo "== lib $v"; COOT_HIP_LIB=$L python tools/chain_probe.py 2>&1 | grep "fused chain" | tee -a $O/chain_xcd$v.txt
done
timeout 300 python -m pytest tests/test_gpu_determinism.py -q -k captured 2>&1 | tail -2
