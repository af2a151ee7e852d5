cd "$GRAFT_REPO_ROOT"
run() { python bench.py --scene 101 --width 3840 --height 2160 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-calibration 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['value'],1), round(d['kernel_ms_per_step'],2), d['fb_checksum'])"; }
for i in 1 2; do run default; for f in gpurun_exp_*.so; do SRT_LIB_PATH=$PWD/$f run $f; done; done
