#!/bin/bash
# Run on the GPU box: instruction mix of the implicit-GEMM conv kernels inside the ViZDoom bench.
export TMPDIR=/tmp
export D2P_GRAPH=0
REPO=$PWD
OUT=$REPO/gpurun_out/conv_pmc
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT -o p$i -- python $REPO/bench.py --preset vizdoom --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/p$i.log 2>&1
done
python - <<'PY' > $REPO/gpurun_out/conv_pmc.txt
import glob, sqlite3, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for db in sorted(glob.glob('/root/repo/gpurun_out/conv_pmc/*.db')):
    con = sqlite3.connect(db)
    for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
        if 'Im2col' not in name and 'Dgrad' not in name and 'conv_' not in name: continue
        key = name.split('(')[0][-90:]
        a = acc[key][counter]; a[0] += 1; a[1] += value
for key, cs in sorted(acc.items()):
    v = cs.get('SQ_INSTS_VALU', [1, 0])[1]; m = cs.get('SQ_INSTS_MFMA', [1, 1])[1]
    b = cs.get('SQ_BUSY_CYCLES', [1, 0]); mb = cs.get('SQ_VALU_MFMA_BUSY_CYCLES', [1, 0])
    util = mb[1] / (b[1] / 32 * 1024) if b[1] else 0
    print('%-92s launches %3d  VALU/MFMA %.1f  MFMA util %.0f%%  us %.0f' % (key, cs['SQ_INSTS_VALU'][0], (v - m) / max(m, 1), 100 * util, b[1] / max(b[0], 1) / 32 / 2400))
PY
rm -f $OUT/*.db
cat $REPO/gpurun_out/conv_pmc.txt
