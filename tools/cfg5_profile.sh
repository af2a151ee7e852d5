#!/bin/bash
# cfg 5's scene (100k-triangle mesh, 3840x2160) on one GPU: counter list of the box, instrumented phase split, bench number at 512 spp.
# usage (through gpurun): tools/cfg5_profile.sh <outdir> [list]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/cfg5}; mkdir -p $O
if [ "${2:-}" = "list" ]; then rocprofv3 --list-avail > $O/counters_avail.txt 2>&1 || rocprofv3 -L > $O/counters_avail.txt 2>&1; grep -c "" $O/counters_avail.txt; fi
python tools/diag.py --scene 101 --width 3840 --height 2160 --spp 16 > $O/diag_cfg5.json 2>&1
grep -h "cycles_shade\|cyc_per\|lane_util\|lanes_shaded\|mray_s\|\"V\"\|\"T\"\|\"ms\"" -A0 $O/diag_cfg5.json
timeout -k 10 600 python bench.py --scene 101 --width 3840 --height 2160 --spp ${CFG5_SPP:-512} --steps 1 --warmup 1 --no-cpu-baseline --no-calibration 2>/dev/null > $O/bench_cfg5.json
python -c "import json;d=json.load(open('$O/bench_cfg5.json'));print('cfg5 scene', d['config']['workload'], round(d['value'],1), 'Mray/s', round(d['kernel_ms_per_step'],1), 'ms')"
