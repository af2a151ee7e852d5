#!/bin/bash
# kernel experiments: the default library, then every gpurun_exp_*.so in the repo root (bench only, 2 steps each)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['value'],1), round(d['kernel_ms_per_step'],2), d['fb_checksum'])"; }
run default
for f in gpurun_exp_*.so; do [ -f "$f" ] && SRT_LIB_PATH=$PWD/$f run $f; done
