#!/bin/bash
# rocprofv3 --pmc passes (one per counter set, own runs) of the forward chain launches at 200 and 62 resident tiles:
#   bash tools/pmc_chain_sets.sh <tag> "<counter set 1>" "<counter set 2>" ...   -> gpurun_out/<tag>/summary.txt
set -u
cd $GRAFT_REPO_ROOT
REPO=$PWD; TAG=$1; shift; O=$REPO/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  for N in 320 100; do
    rm -rf /tmp/pc_${i}_$N && timeout 200 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pc_${i}_$N -o pmc -- python $REPO/tools/chain_pmc_probe.py $N > /dev/null 2> $O/err_${i}_$N.txt
    DB=$(find /tmp/pc_${i}_$N -name "*.db" | head -1)
    [ -n "$DB" ] && python $REPO/tools/rocpd_pmc.py "$DB" $O/pmc_${i}_N$N.csv > /dev/null 2>> $O/err_${i}_$N.txt
  done
done
cd $REPO
python - $O <<'PY' | tee $O/summary.txt
import csv,glob,collections,sys
O=sys.argv[1]
tab=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/pmc_*_N*.csv')):
    N=f.split('_N')[-1].split('.')[0]
    for r in csv.DictReader(open(f)):
        k=r['kernel']
        if 'post_attn_fwd' in k or 'infc_qkv_fwd' in k:
            short='post_attn_fwd' if 'post_attn' in k else 'infc_qkv_fwd'
            tab[(short,r['counter'])][N]=float(r['total'])/max(int(r['dispatches']),1)
for (k,c),v in sorted(tab.items()):
    a,b=v.get('320'),v.get('100')
    print(f"{k:14s} {c:44s} N=320 {a if a is None else round(a):>16} N=100 {b if b is None else round(b):>16}  ratio {a/b if a and b else 0:.2f}")
PY
