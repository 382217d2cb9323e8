set -u
cd $GRAFT_REPO_ROOT
REPO=$PWD; O=$REPO/gpurun_out/r05_pmc_chain; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > $O/counters.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  for N in 320 100; do
    rm -rf /tmp/pc_${i}_$N && timeout 200 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pc_${i}_$N -o pmc -- python $REPO/tools/chain_pmc_probe.py $N > /dev/null 2> $O/err_${i}_$N.txt
    DB=$(find /tmp/pc_${i}_$N -name "*.db" | head -1)
    [ -n "$DB" ] && python $REPO/tools/rocpd_pmc.py "$DB" $O/pmc_${i}_N$N.csv > /dev/null 2>> $O/err_${i}_$N.txt
  done
done
cd $REPO
python - <<'PY'
import csv,glob,collections,os
O='gpurun_out/r05_pmc_chain'
tab=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/pmc_*_N*.csv')):
    N=f.split('_N')[-1].split('.')[0]
    for r in csv.DictReader(open(f)):
        k=r['kernel']
        if 'post_attn_fwd' in k or 'infc_qkv_fwd' in k:
            short='post_attn_fwd' if 'post_attn' in k else 'infc_qkv_fwd'
            tab[(short,r['counter'])][N]=float(r['avg_per_dispatch'])
for (k,c),v in sorted(tab.items()):
    a,b=v.get('320'),v.get('100')
    print(f"{k:14s} {c:28s} N=320 {a if a is None else round(a):>14} N=100 {b if b is None else round(b):>14}  ratio {a/b if a and b else 0:.2f}")
PY
