#!/bin/bash
# Run on the GPU box: instruction-mix / stall counters of the recurrent step kernels inside the
# default bench (eager, one stream).  -> gpurun_out/lstm_pmc.txt
export TMPDIR=/tmp
export D2P_GRAPH=0
REPO=$PWD
OUT=$REPO/gpurun_out/lstm_pmc
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT -o p$i -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/p$i.log 2>&1
done
python - <<'PY' > $REPO/gpurun_out/lstm_pmc.txt
import glob, sqlite3, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for db in sorted(glob.glob('/root/repo/gpurun_out/lstm_pmc/*.db')):
    con = sqlite3.connect(db)
    q = "select k.kernel_name, k.counter_name, k.value, k.grid_size from counters_collection k"
    try:
        rows = con.execute(q).fetchall()
    except Exception:
        rows = [(a, b, c, 0) for a, b, c in con.execute("select kernel_name, counter_name, value from counters_collection")]
    for name, counter, value, grid in rows:
        if 'lstm_step' not in name: continue
        key = name.split('(')[0][-40:] + ' grid=%s' % grid
        a = acc[key][counter]; a[0] += 1; a[1] += value
for key, cs in sorted(acc.items()):
    print(key)
    for c, (n, v) in sorted(cs.items()):
        print('   %-28s %14.0f per launch (%d launches)' % (c, v / n, n))
PY
rm -f $OUT/*.db
cat $REPO/gpurun_out/lstm_pmc.txt
