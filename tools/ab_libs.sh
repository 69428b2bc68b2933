#!/bin/bash
# Same-box A/B of several builds of the library on the headline step: tools/ab_libs.sh "tagA tagB ..." [rounds] [steps]
# (tag "head" = csrc/libd2p_hip.so, tag X = csrc/libd2p_hip_X.so: python demo2program_amd/build.py --variant X DEFINE...)
# prints ms per step, instances/s and the instrumented pass's recurrent forward / backward ms per step (HIP events
# around the launches, one stream)
TAGS=$1; ROUNDS=${2:-3}; STEPS=${3:-300}
P=$PWD/demo2program_amd/csrc
for r in $(seq 1 $ROUNDS); do
  for t in $TAGS; do
    L=$P/libd2p_hip_$t.so; [ "$t" = "head" ] && L=$P/libd2p_hip.so
    out=$(env D2P_LIB_PATH=$L python bench.py --steps $STEPS --warmup 40 --no-cpu-baseline --no-h2d --no-config4 2>/dev/null | tail -1)
    echo "$t $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); p=(d.get("roofline") or {}).get("parts") or [{},{}]; print(d["ms_per_step"], d["value"], "fwd", p[0].get("ms_per_step"), "bwd", p[1].get("ms_per_step"))')"
  done
done
