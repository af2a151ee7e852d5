#!/bin/bash
# round 2, first GPU call: issue-rate calibration, occupancy sweep of the HEAD kernel, PMC passes, phase shares
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_first; mkdir -p $O
python tools/calibrate.py --out $O/calib.json > $O/calib.log 2>&1 || { tail -5 $O/calib.log; exit 1; }
cat $O/calib.log
for w in 8 12 16; do
  SRT_WAVES_PER_CU=$w python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_w$w.json 2> $O/bench_w$w.err || { tail -5 $O/bench_w$w.err; exit 1; }
  python -c "import json;d=json.load(open('$O/bench_w$w.json'));print('waves/CU $w:',d['value'],d['kernel_ms_per_step'])"
done
python tools/diag.py --spp 64 > $O/diag.json 2>&1 || { tail -5 $O/diag.json; exit 1; }
cat $O/diag.json
bash tools/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
tail -40 $O/pmc.log
