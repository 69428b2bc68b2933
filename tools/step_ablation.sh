#!/bin/bash
# What each piece of the training step is worth in the real two-queue schedule: ms per step with the piece left out
# (bench.py --ablate -> Model.set_ablation, timing only -- the results of such a step are invalid), alternating with full steps on ONE box.
# usage: bash tools/step_ablation.sh [steps]
STEPS=${1:-200}
run() { python bench.py ${1:+--ablate $1} --steps $STEPS --warmup 40 --no-cpu-baseline --no-roofline --no-h2d --no-config4 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for piece in conv_fwd rn_fwd logits rn_bwd conv_bwd wgrad wgrad:demo_lstm wgrad:second_lstm adam scatter zq conv_fwd,conv_bwd rn_fwd,rn_bwd conv_fwd,conv_bwd,rn_fwd,rn_bwd,logits; do
  a=$(run none); b=$(run $piece); c=$(run none); d=$(run $piece)
  echo "$piece: full $a $c  without $b $d"
done
