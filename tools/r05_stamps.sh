set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_stamps; mkdir -p $O
for v in "" _nostore; do
  echo "== lib $v"
  COOT_HIP_LIB=$PWD/coot-videotext_amd/lib/libcoot_hip$v.so python tools/fused_stamps.py 2>&1 | grep "train=True" -A1 | grep -v "^--" | tee $O/stamps$v.txt
done
