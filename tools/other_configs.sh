#!/bin/bash
# BASELINE.json configs 2, 4, 5 through bench.py on one GPU (not bench lines: recorded under profiles/ for reference)
OUT=gpurun_out/other_configs.jsonl; : > $OUT
timeout -k 10 300 python bench.py --scene 100 --width 1280 --height 720 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null >> $OUT
timeout -k 10 300 python bench.py --scene 1 --bvh 0 --width 1920 --height 1080 --spp 2048 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null >> $OUT
timeout -k 10 600 python bench.py --scene 101 --width 3840 --height 2160 --spp 4096 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null >> $OUT
cut -c1-420 $OUT
