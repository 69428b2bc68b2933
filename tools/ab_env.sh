#!/bin/bash
# Same-box A/B of one environment switch on the headline step: tools/ab_env.sh VAR A B [rounds] [steps]
# alternates `VAR=A` / `VAR=B` bench runs (no CPU leg, no extras) and prints ms_per_step of each.
VAR=$1; A=$2; B=$3; ROUNDS=${4:-3}; STEPS=${5:-300}
for r in $(seq 1 $ROUNDS); do
  for v in "$A" "$B"; do
    out=$(env $VAR=$v python bench.py --steps $STEPS --warmup 40 --no-cpu-baseline --no-roofline --no-h2d --no-config4 2>/dev/null | tail -1)
    echo "$VAR=$v $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
