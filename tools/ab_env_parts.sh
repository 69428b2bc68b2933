#!/bin/bash
# As tools/ab_env.sh, printing the instrumented pass's recurrent forward / backward ms per step as well:
# tools/ab_env_parts.sh VAR A B [rounds] [steps]
VAR=$1; A=$2; B=$3; ROUNDS=${4:-2}; STEPS=${5:-300}
for r in $(seq 1 $ROUNDS); do
  for v in "$A" "$B"; do
    out=$(env $VAR=$v python bench.py --steps $STEPS --warmup 40 --no-cpu-baseline --no-h2d --no-config4 2>/dev/null | tail -1)
    echo "$VAR=$v $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); p=d["roofline"]["parts"]; print(d["ms_per_step"], "fwd", p[0]["ms_per_step"], "bwd", p[1]["ms_per_step"])')"
  done
done
