#!/bin/bash
# sweep of the step-choice weights (bench, 2 steps each). usage: tools/sweep_scores.sh "<shade list>" "<fringe list>" [bench args]
cd "$GRAFT_REPO_ROOT"
SH=${1:-"40 55 70 90"}; FR=${2:-"200 280 360"}; shift 2 || true
for s in $SH; do for f in $FR; do
  SRT_SCORE_SHADE=$s SRT_SCORE_FRINGE=$f python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('shade $s fringe $f:', round(d['value'],1), round(d['kernel_ms_per_step'],2))"
done; done
