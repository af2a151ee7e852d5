#!/bin/bash
# BASELINE.json configs 2, 4, 5 through bench.py on one GPU (not bench lines: recorded under profiles/ for reference),
# plus the instrumented phase split and the SQ/TCC counter passes for cfg 4 and cfg 5's scene.
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r02_other}; mkdir -p $O; OUT=$O/other_configs.jsonl; : > $OUT
timeout -k 10 300 python bench.py --scene 100 --width 1280 --height 720 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline --no-calibration 2>/dev/null >> $OUT
timeout -k 10 300 python bench.py --scene 1 --bvh 0 --width 1920 --height 1080 --spp 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-calibration 2>/dev/null >> $OUT
timeout -k 10 600 python bench.py --scene 101 --width 3840 --height 2160 --spp ${CFG5_SPP:-512} --steps 1 --warmup 0 --no-cpu-baseline --no-calibration 2>/dev/null >> $OUT
python - $OUT <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['config']['workload'], '|', round(d['value'], 1), 'Mray/s', round(d['kernel_ms_per_step'], 1), 'ms V', round(d['node_records_per_ray_V'], 2), 'T', round(d['tri_tests_per_ray_T'], 2), 'rays/path', round(d['rays_per_path'], 2))
PY
python tools/diag.py --scene 1 --bvh 0 --spp 64 > $O/diag_cfg4.json 2>&1
python tools/diag.py --scene 101 --width 3840 --height 2160 --spp 16 > $O/diag_cfg5.json 2>&1
grep -h "cycles_shade\|cyc_per\|lane_util\|lanes_shaded\|mray_s\|\"V\"\|\"T\"" -A0 $O/diag_cfg4.json $O/diag_cfg5.json
if [ "${PMC:-1}" = "1" ]; then
  bash tools/pmc_passes.sh $O/pmc_cfg4 --scene 1 --bvh 0 --spp 128 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration > $O/pmc_cfg4.log 2>&1
  bash tools/pmc_passes.sh $O/pmc_cfg5 --scene 101 --width 3840 --height 2160 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration > $O/pmc_cfg5.log 2>&1
  echo "--- cfg4 PMC"; cat $O/pmc_cfg4/summary.txt; echo "--- cfg5 PMC"; cat $O/pmc_cfg5/summary.txt
fi
