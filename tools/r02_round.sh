#!/bin/bash
# one measurement round on the GPU box: tests, default bench, diag, PMC passes of the headline scene
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-vX}; O=gpurun_out/r02_$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
python tools/diag.py --spp 64 > $O/diag.json 2>&1
bash tools/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
cat $O/pmc/summary.txt | head -40
