#!/bin/bash
# Run on the GPU box: stall / LDS / occupancy counters of the three large GEMM shapes, one counter
# group per pass (--pmc with --kernel-trace only).  -> gpurun_out/gemm_pmc.txt
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/gemm_pmc
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT -o p$i -- python $REPO/tools/gemm_pmc_probe.py > $OUT/p$i.log 2>&1
done
python - <<'PY' > $REPO/gpurun_out/gemm_pmc.txt
import glob, sqlite3, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for db in sorted(glob.glob('/root/repo/gpurun_out/gemm_pmc/*.db')):
    con = sqlite3.connect(db)
    for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
        if 'gemm_mfma_kernel' not in name: continue
        key = name.split('gemm_mfma_kernel')[1][:60]
        a = acc[key][counter]; a[0] += 1; a[1] += value
for key, cs in acc.items():
    print(key)
    for c, (n, v) in sorted(cs.items()):
        print('   %-28s %14.0f per launch (%d launches)' % (c, v / n, n))
PY
rm -f $OUT/*.db
cat $REPO/gpurun_out/gemm_pmc.txt
