set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cfg5_c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image_bit_exact or mesh100k_blocks or fuzz" > $O/tests_split.log 2>&1; tail -2 $O/tests_split.log
B="--scene 101 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline --no-calibration"
run() { python bench.py $B 2>/dev/null > $O/$1.json; python -c "import json;d=json.load(open('$O/$1.json'));print('$1', round(d['value'],1), 'Mray/s', round(d['kernel_ms_per_step'],1), 'ms', d['fb_checksum'])"; }
SRT_LIB_PATH=$PWD/gpurun_exp_nosplit.so run nosplit
run split_default
for D in 96 128 192 384; do SRT_SCORE_DEEP=$D run split_deep$D; done
for F in 500 1200; do SRT_SCORE_FRINGE=$F run split_fringe$F; done
for S in 200 480; do SRT_SCORE_SHADE=$S run split_shade$S; done
python tools/diag.py --scene 101 --width 3840 --height 2160 --spp 16 > $O/diag_split.json 2>&1
grep -h "cycles_shade\|cyc_per\|lane_util\|lanes_shaded\|mray_s\|\"ms\"" -A0 $O/diag_split.json
